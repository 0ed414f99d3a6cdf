"""CTC loss and gradient of the HIP kernels against a float64 evaluation (torch CPU), next to torch's own float32 result:
python tools/ctc_accuracy.py"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
warnings.filterwarnings("ignore")
torch.manual_seed(0)
for (T, B, V, L, lens) in ((375, 3, 5001, 40, [375, 303, 195]), (1000, 2, 5001, 120, [1000, 777]), (100, 3, 60, 12, [100, 80, 50])):
    x = torch.randn(T, B, V)
    tg = torch.randint(0, V - 1, (B, L)); tl = torch.tensor([L, L - 3, L - 7][:B])
    il = torch.tensor(lens)

    def ref(dt):
        xr = x.to(dt).clone().requires_grad_(True)
        l = torch.nn.functional.ctc_loss(torch.log_softmax(xr, -1), tg, il, tl, blank=V - 1, reduction="sum", zero_infinity=True)
        l.backward()
        return float(l), xr.grad.double()
    l64, g64 = ref(torch.float64)
    l32, g32 = ref(torch.float32)
    xd = K.alloc_rows((T, B), V, torch.float32, "cuda"); xd.copy_(x.cuda())
    loss, grad, nll = K.ctc_loss(xd, tg.cuda(), tl.cuda(), il.to(torch.int32).cuda(), V - 1)
    g = grad.cpu().double()
    e = lambda a: float((a - g64).norm() / g64.norm())
    print("T=%4d V=%4d L=%3d  loss rel: hip %.1e torch-f32 %.1e   gradient |dg|/|g| vs float64: hip %.2e  torch-f32 %.2e" %
          (T, V, L, abs(float(loss) - l64) / l64, abs(l32 - l64) / l64, e(g), e(g32)))
