"""Offline quality check (CPU, numpy) of the dropout hash (csrc/common.hpp drop_hash4_lo) next to the one it replaced (three 32-bit
multiplies, quarter rate on CDNA): avalanche of all 64 output bits for every input bit, chi-square of the
16-bit fields, keep-bit correlations along keys and rows of a real index space (rows of 94 quads), keep counts per row / column
against the binomial variance.  python tools/drop_hash_check.py"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def u32(x): return x & M32
def mul32(a, c): return u32(a * np.uint64(c))
def mad24(a, c, add): return u32((a & np.uint64(0xFFFFFF)) * np.uint64(c) + add)
def xs(a, k): return a ^ (a >> np.uint64(k))


def bitrev(a):
    r = np.zeros_like(a)
    for i in range(32):
        r |= ((a >> np.uint64(i)) & np.uint64(1)) << np.uint64(31 - i)
    return r


def old_hash4(q, ks):
    x = xs(q ^ ks, 16); x = mul32(x, 0x7feb352d); x = xs(x, 15); x = mul32(x, 0x846ca68b); x = xs(x, 16)
    y = xs(mul32(x ^ np.uint64(0x68E31DA4), 0xB5297A4D), 15)
    return x, y


def drop_hash4_lo(q, ks):
    a = q ^ ks
    a = mad24(a, 0x3C6EF2, a); a = bitrev(a)
    a = mad24(a, 0x9E3778, a); a = bitrev(a)
    a = mad24(a, 0x85EBCA, a); x = xs(a, 16)
    y = bitrev(mad24(x ^ np.uint64(0x68E31DA4), 0xC2B2AE, x))
    return x, y


def avalanche(fn, n=200000):
    rng = np.random.default_rng(0)
    q = rng.integers(0, 1 << 25, n, dtype=np.uint64); ks = np.uint64(int(rng.integers(0, 1 << 32)))
    x0, y0 = fn(q, ks)
    worst = 0.0
    for b in range(26):
        x1, y1 = fn(q ^ np.uint64(1 << b), ks)
        for w in (x0 ^ x1, y0 ^ y1):
            for o in range(32):
                worst = max(worst, abs(float(((w >> np.uint64(o)) & np.uint64(1)).mean()) - 0.5))
    return worst


def stats(fn, p=0.1, rows=40000, quads=94):
    ks = np.uint64(0x5bd1e995)
    x, y = fn(np.arange(rows * quads, dtype=np.uint64), ks)
    f = np.stack([y & np.uint64(0xFFFF), y >> np.uint64(16), x & np.uint64(0xFFFF), x >> np.uint64(16)], 1).astype(np.int64)
    keep = f >= int(p * 65536)
    chi = [(((np.bincount((f[:, c] >> sh) & 255, minlength=256) - len(f) / 256) ** 2) / (len(f) / 256)).sum() / 255 for c in range(4) for sh in (8, 0)]
    k = keep.reshape(rows, quads * 4).astype(np.float64) - (1 - p); var = p * (1 - p)
    lag = max(abs(float((k[:, :-d] * k[:, d:]).mean() / var)) for d in (1, 2, 3, 4, 5, 8, 16))
    rowc = max(abs(float((k[:-d] * k[d:]).mean() / var)) for d in (1, 2, 3, 64))
    cnt = keep.reshape(rows, quads * 4)
    return dict(keep_rate=float(keep.mean()), chi_max=max(chi), chi_min=min(chi), key_corr=lag, row_corr=rowc,
                row_count_var=float(cnt.sum(1).var() / (quads * 4 * var)), col_count_var=float(cnt.sum(0).var() / (rows * var)))


for name, fn in (("round-1 hash (3 x v_mul_lo_u32)          ", old_hash4), ("drop_hash4_lo now (4 x v_mad_u32_u24 + 3 x v_bfrev)", drop_hash4_lo)):
    print("%-46s avalanche worst |P - 1/2| = %.4f" % (name, avalanche(fn)), " ".join("%s=%.4f" % kv for kv in stats(fn).items()))
