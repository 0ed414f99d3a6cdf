"""Phase timeline of the second-generation attention kernels from in-kernel s_memtime stamps (diagnostic build:
make -C fbk_fairseq_st_amd/csrc dbg;  S2T_HIP_LIB=fbk_fairseq_st_amd/libs2t_hip_dbg.so python tools/attn_timeline.py [p_drop]).
Stamps per tile: 0 loop top, 1 next tile's DMA issued, 2 scores / probabilities done (fwd, dQ), 3 last MFMA issued, 4 vmcnt(0) passed,
5 barrier passed.  Printed per wave: cycles between consecutive stamps, for the first workgroup and one in the middle of the grid."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K, lib as L
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
dev = "cuda"; B, H, T, d = 64, 8, 375, 64; D = H * d
qkv = torch.randn(T, B, 3 * D, device=dev).to(torch.bfloat16)
q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
o, lse = K.attn_fwd(q, k, v, H, p_drop=p, seed=1)
do = torch.randn_like(o); dqkv = torch.empty_like(qkv)
def run():
    K.attn_fwd(q, k, v, H, p_drop=p, seed=1)
    K.attn_bwd(q, k, v, o, do, lse, H, dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:], p_drop=p, seed=1)
for _ in range(20): run()
st = torch.zeros(3 * 4 * 4 * 64, dtype=torch.int64, device=dev)
h = L.load()
h.s2t_dbg_attn_stamps.argtypes = [ctypes.c_void_p]; h.s2t_dbg_attn_stamps.restype = ctypes.c_int
assert h.s2t_dbg_attn_stamps(st.data_ptr()) == 0
run(); torch.cuda.synchronize()
assert h.s2t_dbg_attn_stamps(None) == 0
s = st.cpu().view(3, 4, 4, 8, 8)
for kid, name in enumerate(("attn_fwd2", "attn_bwd_dq2", "attn_bwd_dkv2")):
    for slot in range(4):
        print("== %s, workgroup %s" % (name, ("0", "700", "1100", "1500")[slot]))
        base = int(s[kid, slot, :, 0, 0][s[kid, slot, :, 0, 0] > 0].min()) if (s[kid, slot, :, 0, 0] > 0).any() else 0
        for w in range(4):
            rows = []
            e = s[kid, slot, w, 7]
            rows.append("entry -> loop %6d | loop %6d | loop end -> exit %6d | total %6d" % (int(e[1] - e[0]), int(e[2] - e[1]), int(e[3] - e[2]), int(e[3] - e[0])))
            for t in range(7):
                x = s[kid, slot, w, t]
                if x[0] == 0: break
                seg = ["%5d" % int(x[i + 1] - x[i]) if x[i + 1] > 0 and x[i] > 0 else "    -" for i in range(5)]
                # stamp 2 is absent in the dK/dV kernel: show 1 -> 3 there
                if x[2] == 0 and x[3] > 0: seg[1] = "%5d" % int(x[3] - x[1]); seg[2] = "    ="
                rows.append("t%d @%6d: dma %s | S/P %s | PV/dQ %s | vmcnt %s | barrier %s" % ((t, int(x[0]) - base) + tuple(seg)))
            print(" wave %d\n   " % w + "\n   ".join(rows))
