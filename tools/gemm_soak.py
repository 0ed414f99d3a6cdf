"""Soak of the PRODUCT gemm256 epilogues (masked, operand-reading, accumulating, 1-bit record) against the 128-wide route, bit for bit,
launch after launch, with and without a store-heavy kernel on a second stream (as the CTC side stream runs during training):
    python tools/gemm_soak.py [launches per variant, default 2000]
Every variant's reference comes from the 128-wide kernels (s2t_set_option "gemm256" 0: same MFMA instruction over K in the same
order, same epilogue arithmetic); the record variants are held to the plain ReLU forms of that route.  Prints one line per variant
and shape and a final "SOAK differing launches: N of T"; exits non-zero when N > 0.  tests/test_kernels_gpu.py runs soak() with 200."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fbk_fairseq_st_amd import kernels as K

DEV = "cuda"
SHAPES = [(24000, 2048, 512), (24000, 512, 2048), (24000, 512, 512), (23000, 1536, 512)]


def variants(M, N, Kd, g):
    dt = torch.bfloat16
    a = torch.randn(M, Kd, device=DEV, generator=g).to(dt); w = (torch.randn(N, Kd, device=DEV, generator=g) * Kd ** -0.5).to(dt)
    b = torch.randn(N, device=DEV, generator=g); r = torch.randn(M, N, device=DEV, generator=g).to(dt)
    dy = torch.randn(M, N, device=DEV, generator=g).to(dt); act = torch.randn(M, Kd, device=DEV, generator=g).to(dt)
    old_c = torch.randn(M, Kd, device=DEV, generator=g).to(dt)
    out = {
        "bias+res+drop (EXT_RES)": (lambda: K.gemm(a, w, bias=b, residual=r, p_drop=0.1, seed=3), None),
        "bias+relu+drop": (lambda: K.gemm(a, w, bias=b, act=K.ACT_RELU, p_drop=0.1, seed=5), None),
        "NN relu-bwd + aux (EXT_AUX)": (lambda: K.gemm(dy, w, trans_b=True, act=K.ACT_RELU_BWD, aux=act, alpha=1.25), None),
        "NN accumulate (EXT_OLD)": (lambda: K.gemm(dy, w, trans_b=True, out=old_c.clone(), accumulate=True), None),
    }
    nb = K.relu_mask_bytes(M, N, Kd)
    if nb:
        rec = torch.zeros(nb, dtype=torch.uint8, device=DEV)
        out["relu record + drop (RELU_MASK)"] = (lambda: K.gemm(a, w, bias=b, act=K.ACT_RELU_MASK, aux_out=rec, p_drop=0.1, seed=5),
                                                  lambda: K.gemm(a, w, bias=b, act=K.ACT_RELU, p_drop=0.1, seed=5))
        w2 = (torch.randn(Kd, N, device=DEV, generator=g) * 0.05).to(dt); dy2 = torch.randn(M, Kd, device=DEV, generator=g).to(dt)
        K.gemm(a, w, bias=b, act=K.ACT_RELU_MASK, aux_out=rec, p_drop=0.1, seed=5)
        rec_in = rec.clone()
        h = K.gemm(a, w, bias=b, act=K.ACT_RELU, p_drop=0.1, seed=5)
        out["NN through the record (RELU_BWD_MASK)"] = (lambda: K.gemm(dy2, w2, trans_b=True, act=K.ACT_RELU_BWD_MASK, aux=rec_in, alpha=1 / 0.9),
                                                         lambda: K.gemm(dy2, w2, trans_b=True, act=K.ACT_RELU_BWD, aux=h, alpha=1 / 0.9))
    return out


def soak(launches, shapes=SHAPES, side_stream=True, verbose=True):
    """-> (differing launches, total launches)"""
    bad = total = 0
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    for (M, N, Kd) in shapes:
        g = torch.Generator(device=DEV).manual_seed(M + N + Kd)
        for name, (fn, ref_fn) in variants(M, N, Kd, g).items():
            old = K.set_option("gemm256", 0)
            try:
                want = (ref_fn or fn)().clone()
            finally:
                K.set_option("gemm256", old)
            nbad = 0
            for modes in ((False, True) if side_stream else (False,)):
                for i in range(launches // (2 if side_stream else 1)):
                    if modes and i % 4 == 0:
                        with torch.cuda.stream(side):
                            junk.fill_(i & 255)                       # a store-only kernel beside the product
                    got = fn()
                    if not torch.equal(got, want):
                        nbad += 1
                    total += 1
            torch.cuda.synchronize()
            bad += nbad
            if verbose:
                print("%-44s %6d x %4d x %4d   %d differing of %d launches" % (name, M, N, Kd, nbad, launches))
    return bad, total


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    print(torch.cuda.get_device_name(0), "| launches per variant and shape:", n)
    bad, total = soak(n)
    print("SOAK differing launches: %d of %d" % (bad, total))
    sys.exit(1 if bad else 0)
