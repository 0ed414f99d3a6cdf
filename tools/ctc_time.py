"""The CTC loss passes at the bench workload's shape (375 frames x 64 utterances x 5,001 units, transcripts of 20-60 units), timed one
by one, and the loss / gradient against float64 torch:  python tools/ctc_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K

T, B, V = int(os.environ.get("T", 375)), int(os.environ.get("B", 64)), 5001
Lmax = int(os.environ.get("L", 60))
ld = (V + 7) // 8 * 8
g = torch.Generator().manual_seed(0)
logits = torch.zeros(T, B, ld, dtype=torch.bfloat16)
logits[..., :V] = (torch.randn(T, B, V, generator=g) * 2.0).to(torch.bfloat16)
tgt = torch.randint(4, V - 1, (B, Lmax), generator=g)
tl = torch.randint(max(Lmax // 3, 1), Lmax + 1, (B,), generator=g)
il = torch.randint(T * 3 // 4, T + 1, (B,), generator=g); il[0] = T
blank = V - 1
x = logits.cuda()[..., :V]
args = (x, tgt.cuda(), tl.cuda(), il.to(torch.int32).cuda(), blank)


def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


loss, grad, nll = K.ctc_loss(*args)
if not os.environ.get("NOREF"):          # (skip under rocprofv3: the float64 CPU reference is what takes the time)
    lp = torch.log_softmax(logits[..., :V].double(), -1).requires_grad_(True)
    ref = torch.nn.functional.ctc_loss(lp, tgt, il, tl, blank=blank, reduction="sum", zero_infinity=True)
    ref.backward()
    # gradient w.r.t. logits from the gradient w.r.t. log-probs: g - softmax * sum(g)
    gl = lp.grad - lp.detach().exp() * lp.grad.sum(-1, keepdim=True)
    gg = grad.double().cpu()[..., :V]
    print("loss %.6f ref %.6f rel %.2e | grad rel %.2e" % (float(loss), float(ref), abs(float(loss) - float(ref)) / abs(float(ref)),
                                                            float((gg - gl).norm() / gl.norm())))
_, ws, _ = K.ctc_loss(*args, defer_grad=True)
t_fwd = timeit(lambda: K.ctc_loss(*args, defer_grad=True))
one = torch.ones(1, device="cuda")
t_bwd = timeit(lambda: K.ctc_loss_grad(ws, one))
_, _, lse = K.ctc_argmax(x, want_lse=True)
t_fwd_lse = timeit(lambda: K.ctc_loss(*args, defer_grad=True, lse=lse))
print("forward (row lse + alpha/beta) %.1f us | with lse given (alpha/beta alone + allocations) %.1f us | gradient %.1f us" % (t_fwd, t_fwd_lse, t_bwd))
