"""Phase timeline of gemm256 from in-kernel s_memtime stamps (diagnostic build: make -C fbk_fairseq_st_amd/csrc dbg;
S2T_HIP_LIB=fbk_fairseq_st_amd/libs2t_hip_dbg.so python tools/gemm_timeline.py N K)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
N, Kd = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(24000, Kd, device="cuda", generator=g).to(torch.bfloat16)
w = (torch.randn(N, Kd, device="cuda", generator=g) * Kd ** -0.5).to(torch.bfloat16)
dbg = torch.zeros(24000, N, dtype=torch.bfloat16, device="cuda")          # aux_out-shaped; the first 8 KiB receive the stamps
bias = torch.randn(N, device="cuda", generator=g) if len(sys.argv) > 3 else None      # third argument: with a bias vector
for _ in range(3):
    K.gemm(a, w, bias=bias)
dbg.zero_()
K.gemm(a, w, bias=bias, aux_out=dbg)
torch.cuda.synchronize()
st = dbg.view(-1)[:16384].view(torch.int64).cpu()
for grp in (0, 1):
    t = st[grp * 512: grp * 512 + 240].view(-1, 2)
    t = t[t[:, 0] > 0]
    if len(t) < 3:
        continue
    mma = (t[:, 1] - t[:, 0]).tolist()
    per = (t[1:, 0] - t[:-1, 0]).tolist()
    print("group %d: %d phases" % (grp, len(t)))
    print("  MFMA cluster (stamp to stamp, cycles):", mma[:24])
    print("  period between MFMA-cluster starts:   ", per[:24])
t0 = st[0:240].view(-1, 2); t1 = st[512:752].view(-1, 2)
n = min(int((t0[:, 0] > 0).sum()), int((t1[:, 0] > 0).sum()))
print("group 1 cluster start minus group 0 cluster start:", (t1[:n, 0] - t0[:n, 0]).tolist()[:24])
print("group 1 start minus group 0 END:", (t1[:n, 0] - t0[:n, 1]).tolist()[:24])

for grp in (0, 1):
    m = st[1024 + grp * 512: 1024 + grp * 512 + 360].view(-1, 3)
    c = st[grp * 512: grp * 512 + 240].view(-1, 2)
    n = min(int((m[:, 0] > 0).sum()), int((c[:, 0] > 0).sum())) - 1
    print("group %d MEM segment: reads issue (t1-t0), DMA issue (t2-t1), MEM start -> own MFMA start, prev MFMA end -> MEM start" % grp)
    print("  reads:", (m[:n, 1] - m[:n, 0]).tolist()[:16])
    print("  dma:  ", (m[:n, 2] - m[:n, 1]).tolist()[:16])
    print("  mem start -> mfma start:", (c[:n, 0] - m[:n, 0]).tolist()[:16])
    print("  prev mfma end -> mem start:", (m[1:n + 1, 0] - c[:n, 1]).tolist()[:16])

# epilogue stamps (round 5): e0 K-loop end, e1 early staging issued, e2 addressing / bias done, (e3 unused),
# e4 all steps issued, e5 drain stores issued, e6 next tile's offsets computed (loop top)
names = ["early-stage", "setup", "all steps", "drain", "next offsets"]
for grp in (0, 1):
    e = st[2048 + grp * 256: 2048 + grp * 256 + 64].view(-1, 8)[:, :7]
    e = e[(e[:, 0] > 0) & (e[:, 5] > 0)]
    for row in e.tolist()[:3]:
        row = row[:3] + row[4:]
        d = [row[i + 1] - row[i] for i in range(5)]
        print("group %d epilogue: total %d cycles | " % (grp, row[5] - row[0]) + ", ".join("%s %d" % (n, v) for n, v in zip(names, d)))
