"""cfg2_s_fp32 update time under different tile-shape thresholds of the f32 products (s2t_set_option gemm_f32_small_nt / _kt / _narrow)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from fbk_fairseq_st_amd import kernels as K  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
combos = [(192, 40, 0), (512, 40, 0), (512, 256, 0), (1024, 512, 0), (192, 40, 512), (512, 256, 1024), (2048, 2048, 0), (64, 16, 0), (64, 16, 512)]
if len(sys.argv) > 1:
    combos = [tuple(int(v) for v in c.split(",")) for c in sys.argv[1:]]
for nt, kt, nar in combos:
    K.set_option("gemm_f32_small_nt", nt); K.set_option("gemm_f32_small_kt", kt); K.set_option("gemm_f32_narrow", nar)
    e = bench.extra_config("cfg2_s_fp32", "s2t_transformer_s", torch.float32, dev, 10, 3, batch=32, frames=1000, tgt_len=30, roofline=True)
    r = e["roofline"]
    print("small_nt %4d small_kt %4d narrow %4d : %.3f ms/update | %s" % (nt, kt, nar, e["ms_per_step"], json.dumps(r["ms_per_step"])), flush=True)
