"""VGPR / AGPR / scratch / SGPR use of every kernel of one HIP source, from hipcc's -Rpass-analysis=kernel-resource-usage remarks:
    python tools/kernel_resources.py fbk_fairseq_st_amd/csrc/decode.hip [substring ...]
A kernel that runs one wave per SIMD may use 512 registers (VGPR + AGPR); anything in `scratch` is a spill to memory."""
import os
import re
import subprocess
import sys

src = sys.argv[1]
pats = sys.argv[2:]
inc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include")
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + inc, "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur, d = None, {}
for l in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = m.group(1); d[cur] = {}
    for k, s in (("VGPRs", "V"), ("AGPRs", "A"), ("ScratchSize [bytes/lane]", "scratch"), ("TotalSGPRs", "S")):
        m = re.search(r"remark:\s+" + re.escape(k) + r": (\d+)", l)
        if m and cur:
            d[cur][s] = int(m.group(1))
for k, v in d.items():
    if not pats or any(p in k for p in pats):
        print("%-90s %s" % (k[:90], v))
