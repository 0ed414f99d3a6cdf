"""Where a decode step's kernels spend their cycles: runs the Cfg5 beam-5 search on the stamps twin of the library
(make -C fbk_fairseq_st_amd/csrc dec_stamps; S2T_HIP_LIB=.../libs2t_hip_decstamps.so) and prints, for workgroup (0, 0) of every kernel of the
chosen step (third argument; default the last), the shader cycles between phase boundaries and the clock (shader cycles per 100 MHz real-time tick)."""
import ctypes
import os
import sys

import torch

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, REPO)
os.environ.setdefault("S2T_HIP_LIB", os.path.join(REPO, "fbk_fairseq_st_amd", "libs2t_hip_decstamps.so"))
import bench  # noqa: E402
from fbk_fairseq_st_amd import lib as L  # noqa: E402
from fbk_fairseq_st_amd.sequence_generator import SequenceGenerator  # noqa: E402

dtype = torch.float32 if len(sys.argv) > 1 and sys.argv[1] == "fp32" else torch.bfloat16
maxlen = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
a, task, model, crit, trainer, _ = bench.build_all("s2t_transformer_m", 16, 1000, 40, 0, 1e-9, dtype, dev, criterion="label_smoothed_cross_entropy",
                                                   max_target_positions=1024)
model.eval()
gen = SequenceGenerator([model], task.target_dictionary, beam_size=5, max_len_a=0.0, max_len_b=maxlen, min_len=1)
gen.device_graph = False
sample = trainer.prepare(task.dummy_batch(seed=100))
net = {"net_input": {k: v for k, v in sample["net_input"].items() if k in ("src_tokens", "src_lengths")}}
# stamps of step number `at` (third argument; default: the last step, which is the forced-EOS one): the step function is wrapped, the
# search synchronised and the stamps read right after that call
at = int(sys.argv[3]) if len(sys.argv) > 3 else -1
buf = (ctypes.c_ulonglong * 128)()
lib = L.load()
inner = lib.s2t_decode_step
calls = [0]


def step(addr, st):
    rc = inner(addr, st)
    if calls[0] == at:
        torch.cuda.synchronize()
        assert L.load_ctypes().s2t_decode_read_stamps(buf) == 0
    calls[0] += 1
    return rc


lib.s2t_decode_step = step
gen.generate([model], net)
torch.cuda.synchronize()
if at < 0 or at >= calls[0]:
    rc = L.load_ctypes().s2t_decode_read_stamps(buf)
    assert rc == 0, rc
print("stamps of step %d of %d" % (at if 0 <= at < calls[0] else calls[0] - 1, calls[0]))
names = ["self", "cross", "ffn", "row", "sent", "logits", "final"]
for k, n in enumerate(names):
    v = list(buf[k * 16:(k + 1) * 16])
    st = [x for x in v[:14] if x]
    if len(st) < 2:
        continue
    d = [st[i + 1] - st[i] for i in range(len(st) - 1)]
    rt = v[15] - v[14]
    print("%-7s total %6d cycles = %.2f us at %.0f MHz | phases %s" % (n, st[-1] - st[0], rt / 100.0, (st[-1] - st[0]) / max(rt, 1) * 100.0, d))
