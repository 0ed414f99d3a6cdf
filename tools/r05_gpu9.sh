#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_gpu9.txt
{
echo "== ln_bwd variants"; python tools/ln_time.py 2>&1 | grep "M=24000\|M=32000"
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from fbk_fairseq_st_amd import kernels as K
DEV='cuda'
def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)
heads, d, T, B = 2, 64, 203, 2
D = heads * d; bf = torch.bfloat16
q, k = rnd(T, B, D, dtype=bf, seed=1).to(DEV), rnd(T, B, D, dtype=bf, seed=2).to(DEV)
v1, v2 = rnd(T, B, D, dtype=bf, seed=3).to(DEV), rnd(T, B, D, dtype=bf, seed=4).to(DEV)
klen = torch.tensor([T, T - 37], dtype=torch.int32, device=DEV)
for seed in (77, 78, 79, 80, 81, 82):
    o1, lse = K.attn_fwd(q, k, v1, heads, klen=klen, p_drop=0.3, seed=seed)
    o2, _ = K.attn_fwd(q, k, v2, heads, klen=klen, p_drop=0.3, seed=seed)
    do = rnd(T, B, D, dtype=bf, seed=5).to(DEV)
    dq, dk, dv = [torch.empty_like(q) for _ in range(3)]
    K.attn_bwd(q, k, v1, o1, do, lse, heads, dq, dk, dv, klen=klen, p_drop=0.3, seed=seed)
    lhs = float((do.double() * o2.double()).sum()); rhs = float((dv.double() * v2.double()).sum())
    print("seed %d: <dO,O(V2)> %.4f  <dV,V2> %.4f  diff %.4f  (test bound %.4f)" % (seed, lhs, rhs, abs(lhs - rhs), 2e-2 * max(1.0, abs(lhs))))
PY
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-300 | tail -40
