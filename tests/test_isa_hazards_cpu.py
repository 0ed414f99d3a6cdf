"""Static check of the compiled gemm256 epilogue (no GPU needed: hipcc cross-compiles to gfx950 assembly).

The data registers of a wide LDS write (ds_write_b96 / b128) or a 16-byte vector store are read out over several cycles after the
instruction has issued; hipcc pads two wait states after a wide buffer store and none after a wide LDS write, and in the gemm256
epilogue -- 16 steps of ds_write_b128 / ds_read_b128 / buffer_store_dwordx4 per tile, with the SIMD partner's memory instructions in
the same queues -- a VALU write two or three instructions behind such an instruction ended up in the stored tile (round 4:
tests/test_kernels_gpu.py::test_gemm256_store_data_hazard_twins, tools/gemm_sched_diff.py).  The epilogue therefore closes every
step with an asm that READS those registers and waits; this test holds the compiler's output to it: in every instantiation of the
kernel no VALU instruction writes the data registers of a wide store within MIN_STATES issue slots (tools/isa_store_hazards.py)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
HIPCC = "/opt/rocm/bin/hipcc"
MIN_STATES = 6


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_gemm256_epilogue_keeps_wide_store_data_untouched(tmp_path):
    import isa_store_hazards
    out = str(tmp_path / "gemm256.s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-I" + os.path.join(REPO, "include"),
           "-S", "--cuda-device-only", os.path.join(REPO, "fbk_fairseq_st_amd", "csrc", "gemm256.hip"), "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    res = isa_store_hazards.scan(open(out).read())
    kernels = {k: v for k, v in res.items() if "gemm256_kernel" in k}
    assert len(kernels) == 20, sorted(kernels)                      # 10 epilogue variants x 2 tile heights
    close = {k: v for k, v in kernels.items() if v[0] is not None and v[0] < MIN_STATES}
    assert not close, "VALU writes to wide-store data within %d states: %s" % (MIN_STATES, close)
    # the transposed LDS reads behind inline asm (gemm_tile.hpp tr_read_asm): hipcc does not track them, the hand-placed
    # `s_waitcnt lgkmcnt(0)` does -- no instruction may name their destination registers before that wait
    tr = isa_store_hazards.scan_tr_reads(open(out).read())
    nn = {k: v for k, v in tr.items() if "gemm256_kernel" in k and v[0] > 0}
    assert nn, "no transposed reads found: the check is looking at the wrong instruction"
    early = {k: v[1][:2] for k, v in nn.items() if v[1]}
    assert not early, "a transposed LDS read's destination is used before s_waitcnt lgkmcnt(0): %s" % early
    # the checker itself: planted hazards of every kind it must see (VALU, both operands of a lane swap, LDS / vector-memory load
    # returns, a writer behind the loop's back edge) are found, an LDS-DMA load and a compare are not mistaken for writers
    assert isa_store_hazards.self_test()
