"""CPU checks of the C-ABI boundary: the library builds/loads and exports every symbol the header declares."""
import os
import re

from fbk_fairseq_st_amd import lib as L

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(REPO, "include", "s2t_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(s2t_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(L.LIB_PATH):
        L.build()
    lib = L.load()
    syms = _header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "symbol %s declared in s2t_hip.h but not exported" % s
    # every declared function is also bound with argtypes in the ctypes layer
    bound = set(L.SIGNATURES) | {"s2t_build_info"}
    assert set(syms) <= bound, set(syms) - bound
    assert lib.s2t_abi_version() == 9
    assert b"gfx950" in lib.s2t_build_info()


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any launch (no GPU needed): EINVAL / ENOTSUP codes."""
    lib = L.load()
    assert lib.s2t_gemm(0, 0, 0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, None, 0, None, None, 0, 0, 0, 1, 1.0, None) == -22
    assert lib.s2t_gemm(0, 0, 0, 0, 0, 4, 4, None, 4, None, 4, None, 4, None, None, 0, None, None, 0, 0, 0, 1, 1.0, None) == 0
    assert lib.s2t_layernorm_fwd(0, None, None, None, None, None, None, 4, 2048, 1e-5, None) == -95
    assert lib.s2t_ctc_rle(None, None, None, None, None, None, None, None, 4, 2, 0, None) == -22


def test_host_ctc_uer_matches_golden():
    import numpy as np
    import torch
    from fbk_fairseq_st_amd import kernels as K
    g = dict(np.load(os.path.join(REPO, "tests", "golden", "ctc_uer.npz")))
    pred = torch.from_numpy(g["lp"]).argmax(-1).to(torch.int32)
    e, n = K.host_ctc_uer(pred, torch.from_numpy(g["in_len"]), torch.from_numpy(g["tgt"]), torch.from_numpy(g["tgt_len"]), int(g["blank"]))
    assert (e, n) == (float(g["errors"]), float(g["total"]))


def test_generated_cpython_binding_matches_the_ctypes_binding():
    """lib.build_fastcall() generates one METH_FASTCALL wrapper per entry of lib.SIGNATURES (compiled against include/s2t_hip.h); it
    must expose the same names and answer the argument-validation calls exactly like ctypes (no GPU needed)"""
    import ctypes
    L.build_fastcall()
    fast = L._load_fastcall(None)
    assert fast is not None, "the generated binding did not load"
    raw = ctypes.CDLL(L.LIB_PATH)
    for name, argtypes in L.SIGNATURES.items():
        assert hasattr(fast, name), name
        getattr(raw, name).argtypes = argtypes
    for name in L._SIZE_T_RESULT:
        getattr(raw, name).restype = ctypes.c_size_t
    calls = [("s2t_gemm", (0, 0, 0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, None, 0, None, None, 0, 0, 0, 1, 1.0, None)),
             ("s2t_gemm", (0, 0, 0, 0, 0, 4, 4, None, 4, None, 4, None, 4, None, None, 0, None, None, 0, 0, 0, 1, 1.0, None)),
             ("s2t_layernorm_fwd", (0, None, None, None, None, None, None, 4, 2048, 1e-5, None)),
             ("s2t_ctc_rle", (None, None, None, None, None, None, None, None, 4, 2, 0, None)),
             ("s2t_gemm_relu_mask_bytes", (24000, 2048, 512)), ("s2t_gemm_relu_mask_bytes", (2560, 2048, 512)),
             ("s2t_dropout", (1, None, None, 0, 0.1, 2 ** 63 + 5, None)), ("s2t_abi_version", ())]
    for name, args in calls:
        assert getattr(fast, name)(*args) == getattr(raw, name)(*args), name
    assert fast.s2t_set_option(b"no_such_option", 1) == -22 and fast.s2t_build_info() == L.load().s2t_build_info()
    import pytest
    with pytest.raises(TypeError):
        fast.s2t_abi_version(1)
