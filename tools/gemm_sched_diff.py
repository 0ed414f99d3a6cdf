"""Where do the two gemm256 schedules differ?  python tools/gemm_sched_diff.py [epilogue 0..6] [M N K]
Decodes every differing element into the 256-wide kernel's coordinates: tile, wave (wr, wc), accumulator step (hm, ii, pp), lane (r16, q)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
dev, dt = "cuda", torch.bfloat16
ep = int(sys.argv[1]) if len(sys.argv) > 1 else 3
m, N, Kd = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (24000, 512, 512)
g = torch.Generator(device=dev).manual_seed(m + N + Kd)
a = torch.randn(m, Kd, device=dev, generator=g).to(dt); w = (torch.randn(N, Kd, device=dev, generator=g) * Kd ** -0.5).to(dt)
b = torch.randn(N, device=dev, generator=g); r = torch.randn(m, N, device=dev, generator=g).to(dt)
dy = torch.randn(m, N, device=dev, generator=g).to(dt); aux = torch.randn(m, Kd, device=dev, generator=g).to(dt)
calls = [lambda: K.gemm(a, w, bias=b), lambda: K.gemm(a, w, bias=b, p_drop=0.25, seed=9),
         lambda: K.gemm(a, w, bias=b, residual=r, p_drop=0.1, seed=3), lambda: K.gemm(a, w, bias=b, act=K.ACT_RELU, p_drop=0.1, seed=5),
         lambda: K.gemm(a, w, bias=b, act=K.ACT_RELU, residual=r), lambda: K.gemm(dy, w, trans_b=True),
         lambda: K.gemm(dy, w, trans_b=True, act=K.ACT_RELU_BWD, aux=aux, alpha=1.25)]
fn = calls[ep]
K.set_option("gemm256_sched", 0); want = fn().clone()
K.set_option("gemm256_sched", 1)
Mo, No = want.shape
t192 = ((Mo + 191) // 192) * ((No + 255) // 256); t256 = ((Mo + 255) // 256) * ((No + 255) // 256)
use192 = ((t192 + 255) // 256) * 192 < ((t256 + 255) // 256) * 256
BM = 192 if use192 else 256; HR = BM // 4
print("output %d x %d, tile rows %d" % (Mo, No, BM))
for rep in range(3):
    got = fn()
    d = (got != want).nonzero().cpu()
    print("launch %d: %d differing values" % (rep, d.shape[0]))
    if d.shape[0] == 0:
        continue
    gv, wv = got.cpu().float(), want.cpu().float()
    cnt = collections.Counter(); lanes = collections.Counter(); steps = collections.Counter(); waves = collections.Counter(); tiles = collections.Counter()
    for (row, col) in d.tolist():
        tr, tc = row // BM, col // 256
        lr, lc = row % BM, col % 256
        wr = lr // (2 * HR); hm = (lr % (2 * HR)) // HR; ii = (lr % HR) // 16; r16 = lr % 16
        pp = lc // 128; wc = (lc % 128) // 32; j = (lc % 32) // 16; q = (lc % 16) // 4; e = lc % 4
        lanes[(r16, q)] += 1; steps[(hm, ii, pp, j)] += 1; waves[(wr, wc)] += 1; tiles[(tr, tc)] += 1; cnt[e] += 1
    print("  by lane (r16, q):", sorted(lanes.items())[:24])
    print("  by step (hm, ii, pp, j):", sorted(steps.items()))
    print("  by wave (wr, wc):", sorted(waves.items()))
    print("  tiles hit: %d, element in quad: %s" % (len(tiles), dict(cnt)))
    for (row, col) in d[:12].tolist():
        print("   (%d, %d): got %g want %g" % (row, col, gv[row, col], wv[row, col]))
K.set_option("gemm256_sched", 0)
