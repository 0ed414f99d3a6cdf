"""Run only the encoder-shaped attention kernels (for rocprofv3 --pmc passes): python tools/attn_only.py [p_drop]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
dev = "cuda"; B, H, T, d = 64, 8, 375, 64; D = H * d
qkv = torch.randn(T, B, 3 * D, device=dev).to(torch.bfloat16)
q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
o, lse = K.attn_fwd(q, k, v, H, p_drop=p, seed=1)
do = torch.randn_like(o); dqkv = torch.empty_like(qkv)
for _ in range(3):
    K.attn_fwd(q, k, v, H, p_drop=p, seed=1)
    K.attn_bwd(q, k, v, o, do, lse, H, dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:], p_drop=p, seed=1)
torch.cuda.synchronize()
