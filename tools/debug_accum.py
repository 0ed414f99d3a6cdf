import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_trainer_gpu import _setup
a, task, model, crit, tr = _setup(torch.float32, dropout=0.0)
s1 = tr.prepare(task.dummy_batch(seed=1)); s2 = tr.prepare(task.dummy_batch(seed=2))
model.train(); crit.train()
def grads(samples):
    tr.optimizer.zero_grad()
    for s in samples:
        l, _, _ = crit(model, s); l.backward()
    return model.arena.grad.clone()
g1 = grads([s1]); g1b = grads([s1]); g2 = grads([s2]); both = grads([s1, s2])
print("repeat diff", float((g1 - g1b).abs().max()))
for n, (off, cnt, shp) in model.arena.slices.items():
    d = (both[off:off+cnt] - g1[off:off+cnt] - g2[off:off+cnt]).abs().max().item()
    m = both[off:off+cnt].abs().max().item()
    if d > 1e-3 * max(m, 1e-3): print("%-55s diff %.3e max %.3e" % (n, d, m))
