import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
torch.manual_seed(0)
def check(T, B, V, Lm, tl, il):
    blank = V - 1
    logits = torch.randn(T, B, V) * 2
    tgt = torch.randint(0, blank, (B, max(Lm, 1)))
    tl = torch.tensor(tl); il = torch.tensor(il)
    lp = torch.log_softmax(logits.double(), -1).requires_grad_(True)
    ref = torch.nn.functional.ctc_loss(lp, tgt, il, tl, blank=blank, reduction="sum", zero_infinity=True)
    ref.backward()
    gl = lp.grad - lp.detach().exp() * lp.grad.sum(-1, keepdim=True)
    loss, grad, nll = K.ctc_loss(logits.cuda(), tgt.cuda(), tl.cuda(), il.to(torch.int32).cuda(), blank)
    dl = abs(float(loss) - float(ref)) / max(abs(float(ref)), 1e-9)
    dg = float((grad.double().cpu() - gl).norm() / max(float(gl.norm()), 1e-12))
    print("T=%d B=%d V=%d L=%d tl=%s il=%s: loss %.6f ref %.6f rel %.1e grad rel %.1e nll %s" % (T, B, V, Lm, tl.tolist(), il.tolist(), float(loss), float(ref), dl, dg, [round(float(x), 3) for x in nll]))
    assert dl < 1e-5 and dg < 1e-4 and torch.isfinite(grad).all()
check(1, 3, 20, 2, [1, 0, 2], [1, 1, 1])
check(5, 4, 20, 3, [3, 0, 1, 2], [5, 0, 1, 3])          # an utterance with no frames at all
check(17, 2, 6, 8, [8, 8], [17, 16])                     # 8 repeated-prone units in 17 / 16 frames (tight)
check(40, 3, 5001, 31, [31, 16, 1], [40, 40, 33])        # SPL boundary: 63 positions
check(70, 3, 50, 32, [32, 31, 1], [70, 70, 65])          # 65 positions -> 2 per lane
check(300, 2, 50, 127, [127, 64], [300, 260])            # 255 positions -> 4 per lane (boundary)
check(300, 2, 50, 128, [128, 64], [300, 260])            # 257 -> 8 per lane
print("ok")
