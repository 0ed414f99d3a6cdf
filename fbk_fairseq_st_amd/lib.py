"""ctypes binding of libs2t_hip.so (include/s2t_hip.h).

The product path has NO fallback: if the library is missing, cannot be loaded or lacks a symbol,
importing an op raises.  Tensors cross the boundary as raw device pointers (``tensor.data_ptr()``)
plus sizes; the HIP stream is torch's current stream.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("S2T_HIP_LIB") or os.path.join(_HERE, "libs2t_hip.so")     # S2T_HIP_LIB: diagnostic twin (make dbg)
CSRC = os.path.join(_HERE, "csrc")

F32, BF16 = 0, 1
ABI_VERSION = 9              # s2t_abi_version() of the library this binding was written against
ACT_NONE, ACT_RELU, ACT_GELU, ACT_RELU_BWD, ACT_GELU_BWD, ACT_RELU_MASK, ACT_RELU_BWD_MASK = 0, 1, 2, 3, 4, 5, 6

c_int, c_long, c_float, c_double, c_void_p, c_size_t = (ctypes.c_int, ctypes.c_long, ctypes.c_float,
                                                        ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t)
c_ull = ctypes.c_ulonglong
c_longlong = ctypes.c_longlong
P = c_void_p

# name -> argtypes (restype is int unless noted).  Mirrors include/s2t_hip.h one to one.
SIGNATURES = {
    "s2t_abi_version": [],
    "s2t_gemm": [c_int] * 7 + [P, c_int, P, c_int, P, c_int, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_float, P],
    "s2t_gemm_gather": [c_int] * 7 + [P, c_int, P, c_int, P, c_int, P, P, c_int, P, P, c_int, c_int, c_int, c_int,
                                       c_float, P, c_int, P, P, c_float, c_ull, P],
    "s2t_gemm_relu_mask_bytes": [c_int, c_int, c_int],          # returns size_t
    "s2t_linear_wgrad": [c_int, c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, P, c_int, P],
    "s2t_wgrad_group": [c_int, P, P],
    "s2t_wgrad_group_f32": [c_int, P, P],
    "s2t_colsum": [c_int, P, c_int, c_int, c_int, P, P],
    "s2t_attn_fwd": [c_int] * 6 + [P, c_long, c_long] * 4 + [P, P, c_int, c_int, c_float, c_float, c_ull, P],
    "s2t_attn_bwd": [c_int] * 6 + [P, c_long, c_long] * 5 + [P, P] + [P, c_long, c_long] * 3 +
                    [P, c_int, c_int, c_float, c_float, c_ull, P],
    "s2t_attn_probs_avg": [c_int] * 6 + [P, c_long, c_long] * 2 + [P, c_int, c_float, P, P],
    "s2t_layernorm_fwd": [c_int, P, P, P, P, P, P, c_int, c_int, c_float, P],
    "s2t_layernorm_bwd": [c_int, P, P, P, P, P, P, P, P, P, c_int, c_int, P, c_float, c_ull, P],
    "s2t_conv1_fwd": [c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P],
    "s2t_conv1_bwd": [c_int, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "s2t_conv1_bwd_bn": [c_int] + [P] * 12 + [c_int, c_int, c_int, c_int, c_double, c_int, P],
    "s2t_chan_sums": [c_int, P, P, P, P, P, c_long, c_int, c_int, P],
    "s2t_bn_finalize": [P] * 10 + [c_double, c_int, c_int, c_float, c_float, P],
    "s2t_bn_apply": [c_int, P, P, P, P, c_long, c_int, c_float, c_ull, P],
    "s2t_bn_bwd_apply": [c_int, P, P, P, P, P, P, P, P, P, P, c_long, c_int, c_double, c_int, P],
    "s2t_permute_cf": [c_int, P, P, c_int, c_int, c_int, c_int, P],
    "s2t_permute_conv_w": [c_int, P, P, c_int, c_int, c_int, P],
    "s2t_add_pos": [c_int, P, P, P, P, c_int, c_int, c_int, c_float, c_ull, P],
    "s2t_ctc_argmax": [c_int, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "s2t_ctc_rle": [P] * 8 + [c_int, c_int, c_int, P],
    "s2t_ctc_compress_fwd": [c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "s2t_ctc_compress_bwd": [c_int, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "s2t_ctc_loss": [c_int, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P, P],
    "s2t_lsce": [c_int, P, P, P, P, c_long, c_int, c_int, c_float, c_int, c_float, P],
    "s2t_kd_loss": [c_int, P, P, P, P, P, P, c_long, c_int, c_int, c_int, c_float, c_float, c_int, c_float, P],
    "s2t_embed_fwd": [c_int, P, P, P, P, c_int, c_int, c_int, c_float, c_int, c_int, P],
    "s2t_log_softmax": [c_int, P, P, c_long, c_int, c_int, c_float, P],
    "s2t_softmax_probs": [c_int, P, P, c_long, c_int, c_int, c_float, P],
    "s2t_softmax_bwd": [c_int, P, P, P, c_long, c_int, c_int, c_float, c_int, P],
    "s2t_ensemble_lse": [c_int, P, P, c_size_t, P],
    "s2t_embed_bwd": [c_int, P, P, P, c_int, c_int, c_int, c_float, c_int, P],
    "s2t_act_bwd": [c_int, P, P, P, c_size_t, c_int, c_float, c_ull, P],
    "s2t_add_inplace": [c_int, P, P, c_size_t, P],
    "s2t_dropout": [c_int, P, P, c_size_t, c_float, c_ull, P],
    "s2t_grad_norm_clip": [P, c_size_t, P, c_float, c_float, P, P],
    "s2t_grad_norm_clip_div": [P, c_size_t, P, c_float, P, c_float, P, P],
    "s2t_adam_step": [P, P, P, P, P, c_size_t, P, c_float, c_float, c_float, c_float, c_float, c_int, P],
    "s2t_cast": [c_int, c_int, P, P, c_size_t, P],
    "s2t_scale_by_device_scalar": [c_int, P, c_size_t, P, P],
    "s2t_set_option": [ctypes.c_char_p, c_int],
    "s2t_comm_standin": [P, P, c_size_t, c_int, c_int, c_float, P],
    "s2t_prof_enable": [c_int],
    "s2t_prof_reset": [],
    "s2t_prof_read": [ctypes.c_char_p, P, P, P, P],
    "s2t_conv2_wgrad": [c_int, P, P, P, c_int, c_int, c_int, c_int, P],
    "s2t_conv2_fwd": [c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P],
    "s2t_conv2_dgrad": [c_int, P, P, P, c_int, c_int, c_int, c_int, c_float, c_ull, P],
    "s2t_topk": [c_int, P, P, P, c_long, c_int, c_int, c_int, P],
    "s2t_augment": [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "s2t_a2d_chan_stats": [c_int, P, P, P, P, P, P, P, P, c_long, c_int, c_int, c_int, c_int, c_int, P],
    "s2t_a2d_bn_act": [c_int, P, P, P, P, P, P, c_long, c_int, c_int, c_int, P],
    "s2t_a2d_bn_bwd": [c_int, P, P, P, P, P, P, P, P, P, c_long, c_int, c_int, c_int, c_int, c_double, c_int, P],
    "s2t_a2d_param_grads": [P, P, P, c_int, P],
    "s2t_a2d_pack_w": [c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, P],
    "s2t_a2d_conv_wgrad": [c_int, P, c_int, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, P],
    "s2t_a2d_planes": [c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "s2t_a2d_time_fwd": [c_int, P, P, P, c_int, c_int, c_int, c_float, c_ull, P],
    "s2t_a2d_time_bwd": [c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_float, c_ull, P],
    "s2t_a2d_freq_fwd": [c_int, P, P, P, c_int, c_int, c_int, c_float, c_ull, P],
    "s2t_a2d_freq_bwd": [c_int, P, P, P, P, c_int, c_int, c_int, c_float, c_ull, P],
    "s2t_layer_ws_bytes": [P, c_int],                           # returns size_t
    "s2t_layer_bwd_tmp_bytes": [P],                             # returns size_t
    "s2t_layer_fwd": [P, P, P],
    "s2t_layer_bwd": [P, P, P],
    "s2t_decode_prepare_enc": [c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, P],
    "s2t_decode_pack_weight": [c_int, P, c_int, c_int, c_int, P, P],
    "s2t_decode_begin": [P, c_int, P],
    "s2t_decode_step": [P, P],
    "s2t_decode_lds_bytes": [P],                                # returns size_t
    "s2t_decode_graph_create": [P, c_int, P],
    "s2t_decode_graph_launch": [P, P],
    "s2t_decode_graph_destroy": [P],
    "s2t_host_batch_by_size": [P, c_longlong, P, c_longlong, c_longlong, c_int, P, P, P],
    "s2t_host_ctc_uer": [P, P, c_int, c_int, P, P, c_int, c_int, P, P],
}

class WgradProblem(ctypes.Structure):
    """S2TWgradProblem of include/s2t_hip.h"""
    _fields_ = [("dY", c_void_p), ("X", c_void_p), ("dW", c_void_p), ("db", c_void_p),
                ("n_out", c_int), ("n_in", c_int), ("tokens", c_int), ("ldy", c_int), ("ldx", c_int), ("ldw", c_int)]


_LAYER_W = ("qkv", "o", "xq", "xkv", "xo", "fc1", "fc2")
_LAYER_LN = ("ln1_g", "ln1_b", "lnx_g", "lnx_b", "ln2_g", "ln2_b")


class LayerDesc(ctypes.Structure):
    """S2TLayerDesc of include/s2t_hip.h"""
    _fields_ = ([(n, c_int) for n in ("dtype", "decoder", "T", "B", "D", "heads", "ffn", "Ts", "gelu", "causal", "dist_penalty")] +
                [(n, c_float) for n in ("ln_eps", "p_drop", "p_attn", "p_act")] +
                [("w_" + n, c_void_p) for n in _LAYER_W] + [("b_" + n, c_void_p) for n in _LAYER_W] + [(n, c_void_p) for n in _LAYER_LN] +
                [("g_w_" + n, c_void_p) for n in _LAYER_W] + [("g_b_" + n, c_void_p) for n in _LAYER_W] + [("g_" + n, c_void_p) for n in _LAYER_LN])


class LayerCall(ctypes.Structure):
    """S2TLayerCall of include/s2t_hip.h"""
    _fields_ = [("training", c_int), ("self_klen", c_void_p), ("enc_klen", c_void_p),
                ("seed_sa_attn", c_ull), ("seed_sa_out", c_ull), ("seed_xa_attn", c_ull), ("seed_xa_out", c_ull), ("seed_ffn_act", c_ull),
                ("seed_ffn_out", c_ull), ("x", c_void_p), ("enc", c_void_p), ("y", c_void_p), ("ws", c_void_p),
                ("dy", c_void_p), ("dy_drop", c_void_p), ("dx", c_void_p), ("dx_drop", c_void_p), ("nxt_p", c_float), ("nxt_seed", c_ull),
                ("denc", c_void_p), ("denc_accumulate", c_int), ("tmp", c_void_p), ("items", c_void_p), ("max_items", c_int), ("n_items", c_int)]


_DEC_LAYER = ("ln1_g", "ln1_b", "w_qkv", "b_qkv", "w_o", "b_o", "lnx_g", "lnx_b", "w_xq", "b_xq", "w_xo", "b_xo", "ln2_g", "ln2_b",
              "w_fc1", "b_fc1", "w_fc2", "b_fc2", "kv_enc", "vt_enc", "kv_cache")


class DecodeLayer(ctypes.Structure):
    """S2TDecodeLayer of include/s2t_hip.h"""
    _fields_ = [(n, c_void_p) for n in _DEC_LAYER]


class DecodeDesc(ctypes.Structure):
    """S2TDecodeDesc of include/s2t_hip.h"""
    _fields_ = ([(n, c_int) for n in ("dtype", "B", "beam", "D", "heads", "ffn", "layers", "V", "ldv", "Ts", "Tsp", "max_len", "min_len",
                                      "ffn_slices", "gelu", "pad", "unk", "eos", "step0_all_slots")] +
                [(n, c_float) for n in ("ln_eps", "embed_scale", "unk_penalty", "inv_temperature")] +
                [(n, c_void_p) for n in ("layer", "lnf_g", "lnf_b", "w_out", "embed", "pos_table", "enc_klen", "init_scores",
                                         "x0", "x1", "part0", "part1", "xn", "logits", "steps", "anc", "cand_val", "cand_idx",
                                         "tok_hist", "par_hist", "cum_hist", "blacklist", "nfin", "finished", "fin_step", "fin_row",
                                         "fin_score")])


_lib = None


def build(verbose=False, twins=False):
    """Compile csrc/*.hip for gfx950 into libs2t_hip.so (hipcc cross-compiles without a GPU), then the CPython binding of its C ABI.
    twins=True also builds the two diagnostic libraries of the store-data hazard reproducer (tests/test_kernels_gpu.py, marked slow:
    S2T_SLOW=1); the product build does not depend on them."""
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 4)), "all"] + (["twins"] if twins else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libs2t_hip.so failed:\n" + res.stdout[-4000:])
    if verbose:
        print(res.stdout[-2000:])
    build_fastcall()
    return LIB_PATH


# ---- the binding itself.  ctypes converts every argument of a call through Python-level machinery: 4.0 us for the 30 arguments of
# s2t_gemm_gather against 0.3 us for a METH_FASTCALL wrapper doing the same conversions in C (measured in the build container).  At
# ~1,100 launches per update that is ~4 ms of host time -- more than the kernels of an 8-utterance batch take -- so the binding is a
# small CPython extension GENERATED from the signature table below: one wrapper per entry point of include/s2t_hip.h, which it
# includes (the C compiler checks every generated call against the header's prototype).  Same names, same arguments, same return
# values as the ctypes binding, which stays as the way in when the extension is absent (S2T_HIP_LIB diagnostic twins, a tree that was
# never built here): both call the same libs2t_hip.so, neither computes anything.
FAST_PATH = os.path.join(_HERE, "_s2t_fastcall.so")
_HOST_SIDE = ("s2t_host_batch_by_size", "s2t_host_ctc_uer")          # CPU work: release the GIL around the call, as ctypes does
_SIZE_T_RESULT = ("s2t_gemm_relu_mask_bytes", "s2t_layer_ws_bytes", "s2t_layer_bwd_tmp_bytes", "s2t_decode_lds_bytes")


def signature_hash():
    """hash of the table the binding is generated from (names, argument types, result kinds): baked into the generated module, so a
    module built from an older table is never used (its wrappers erase types through a function-pointer cast: a changed argument
    type with the same count would be mis-marshalled without any error)"""
    import hashlib
    text = ";".join("%s(%s)%s" % (n, ",".join(t.__name__ for t in SIGNATURES[n]), "z" if n in _SIZE_T_RESULT else "i") for n in sorted(SIGNATURES))
    return int(hashlib.sha256(text.encode()).hexdigest()[:15], 16)


def _fastcall_source():
    out = ['#define PY_SSIZE_T_CLEAN', '#include <Python.h>', '#include "s2t_hip.h"', '',
           'static inline unsigned long long as_ptr(PyObject* o) { return o == Py_None ? 0ull : PyLong_AsUnsignedLongLongMask(o); }', '']
    names = sorted(SIGNATURES)
    for name in names:
        at = SIGNATURES[name]
        n = len(at)
        decl, call = [], []
        for i, t in enumerate(at):
            if t in (c_float, c_double):
                decl.append("    const double a%d = PyFloat_AsDouble(args[%d]);" % (i, i))
                call.append("(%s)a%d" % ("float" if t is c_float else "double", i))
            elif t is P:
                decl.append("    const unsigned long long a%d = as_ptr(args[%d]);" % (i, i))
                call.append("(void*)a%d" % i)
            elif t is ctypes.c_char_p:
                decl.append("    const char* a%d = PyBytes_AsString(args[%d]);" % (i, i))
                call.append("a%d" % i)
            elif t is c_ull:
                decl.append("    const unsigned long long a%d = PyLong_AsUnsignedLongLongMask(args[%d]);" % (i, i))
                call.append("a%d" % i)
            else:                                  # c_int, c_long, c_longlong, c_size_t
                decl.append("    const long long a%d = PyLong_AsLongLong(args[%d]);" % (i, i))
                call.append({c_int: "(int)a%d", c_long: "(long)a%d", c_longlong: "(long long)a%d", c_size_t: "(size_t)a%d"}[t] % i)
        out.append("static PyObject* w_%s(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {" % name)
        out.append('    if (nargs != %d) { PyErr_SetString(PyExc_TypeError, "%s takes %d arguments"); return NULL; }' % (n, name, n))
        out += decl
        out.append("    if (PyErr_Occurred()) return NULL;")
        # the header's prototypes carry const / typed pointers: go through a function pointer of the table's erased types
        proto = ", ".join({c_float: "float", c_double: "double", P: "void*", ctypes.c_char_p: "const char*", c_ull: "unsigned long long",
                           c_int: "int", c_long: "long", c_longlong: "long long", c_size_t: "size_t"}[t] for t in at) or "void"
        ret = "size_t" if name in _SIZE_T_RESULT else "int"
        out.append("    %s (*fn)(%s) = (%s (*)(%s))%s;" % (ret, proto, ret, proto, name))
        if name in _HOST_SIDE:
            out.append("    %s r;" % ret)
            out.append("    Py_BEGIN_ALLOW_THREADS")
            out.append("    r = fn(%s);" % ", ".join(call))
            out.append("    Py_END_ALLOW_THREADS")
        else:
            out.append("    const %s r = fn(%s);" % (ret, ", ".join(call)))
        out.append("    return %s;" % ("PyLong_FromSize_t(r)" if ret == "size_t" else "PyLong_FromLong((long)r)"))
        out.append("}")
    out.append("static PyObject* w_s2t_build_info(PyObject* self, PyObject* const* args, Py_ssize_t nargs) { return PyBytes_FromString(s2t_build_info()); }")
    out.append("static PyObject* w_s2t_signature_hash(PyObject* self, PyObject* const* args, Py_ssize_t nargs) { return PyLong_FromLongLong(%dLL); }" % signature_hash())
    out.append("static PyMethodDef methods[] = {")
    for name in names + ["s2t_build_info", "s2t_signature_hash"]:
        out.append('    {"%s", (PyCFunction)(void (*)(void))w_%s, METH_FASTCALL, ""},' % (name, name))
    out.append("    {NULL, NULL, 0, NULL}};")
    out.append('static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_s2t_fastcall", "generated binding of include/s2t_hip.h", -1, methods};')
    out.append("PyMODINIT_FUNC PyInit__s2t_fastcall(void) { return PyModule_Create(&moddef); }")
    return "\n".join(out) + "\n"


def build_fastcall():
    """generate + compile the CPython binding next to libs2t_hip.so (gcc; links the library by its path relative to $ORIGIN)"""
    import sysconfig
    obj = os.path.join(CSRC, "_obj")
    os.makedirs(obj, exist_ok=True)
    src = os.path.join(obj, "s2t_fastcall.c")
    text = _fastcall_source()
    if not (os.path.exists(src) and open(src).read() == text and os.path.exists(FAST_PATH)
            and os.path.getmtime(FAST_PATH) >= os.path.getmtime(os.path.join(_HERE, "libs2t_hip.so"))):
        with open(src, "w") as f:
            f.write(text)
        cmd = ["gcc", "-O2", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(os.path.dirname(_HERE), "include"),
               src, "-o", FAST_PATH, "-L" + _HERE, "-l:libs2t_hip.so", "-Wl,-rpath,$ORIGIN"]
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            raise RuntimeError("building the CPython binding failed:\n" + res.stdout[-4000:])
    return FAST_PATH


def load():
    """Load the library and bind every symbol of the header; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libs2t_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU or PyTorch fallback for the S2T hot path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)         # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.s2t_build_info.restype = ctypes.c_char_p
    for name in _SIZE_T_RESULT:
        getattr(lib, name).restype = c_size_t
    lib.s2t_build_info.argtypes = []
    if lib.s2t_abi_version() != ABI_VERSION:
        raise ImportError("libs2t_hip.so at %s has ABI version %d, this package binds version %d -- rebuild it (make -C %s)"
                          % (LIB_PATH, lib.s2t_abi_version(), ABI_VERSION, CSRC))
    global _ctypes_lib
    _ctypes_lib = lib
    _lib = _load_fastcall(lib) or lib
    return _lib


def _load_fastcall(ctypes_lib):
    """the generated CPython binding when it is there, matches this table and sits on the library just loaded; else None (ctypes)"""
    if os.environ.get("S2T_HIP_LIB") or os.environ.get("S2T_HIP_CTYPES") or not os.path.exists(FAST_PATH):
        return None
    import importlib.util
    try:
        spec = importlib.util.spec_from_file_location("_s2t_fastcall", FAST_PATH)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    except Exception:
        return None
    if any(not hasattr(mod, n) for n in SIGNATURES) or mod.s2t_abi_version() != ABI_VERSION:
        return None
    if getattr(mod, "s2t_signature_hash", lambda: None)() != signature_hash():
        import warnings                   # generated from another table (SIGNATURES edited, or `make` without build()): ctypes instead
        warnings.warn("fbk_fairseq_st_amd: the generated binding %s was built from another signature table; falling back to ctypes "
                      "(~4 us per launch instead of 0.3) -- run __graft_entry__.build()" % FAST_PATH)
        return None
    return mod


_ctypes_lib = None


def load_ctypes():
    """the ctypes handle of the library (diagnostic symbols outside the table, tools/); load() may hand out the generated module"""
    load()
    return _ctypes_lib


class S2THipError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        if rc <= -1000:
            raise S2THipError("%s: HIP error %d" % (what, -rc - 1000))
        raise S2THipError("%s: error %d (%s)" % (what, rc, {-22: "invalid argument", -95: "not supported"}.get(rc, "?")))


def dt(t):
    """dtype code of a tensor (float32 / bfloat16 only)."""
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError("S2T kernels take float32 or bfloat16 tensors, got %s" % t.dtype)


def ptr(t):
    """device pointer of a tensor argument of a kernel entry point; host tensors are refused here (a host pointer handed to a
    kernel is a GPU memory fault that kills the process) -- host-side entry points take `.data_ptr()` themselves"""
    if t is None:
        return 0
    if not t.is_cuda:
        raise S2THipError("S2T kernels need device tensors: the hot path has no CPU fallback")
    return t.data_ptr()


_GET_DEVICE = torch._C._cuda_getDevice          # current device index of this thread, ~0.1 us (no Python-level bookkeeping)


def device_index():
    """index of torch's CURRENT device (not cached: a process that calls kernels before torch.cuda.set_device, or drives a second
    GPU from a tool, must never launch on device 0's stream with device-N pointers)"""
    return _GET_DEVICE()


def stream():
    """raw hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() costs ~4 us of Python per
    call (stream object construction); with ~1100 launches per update that is host time the small decoder-side kernels cannot
    hide, so the raw handle is fetched through the two C entry points directly."""
    return torch._C._cuda_getCurrentRawStream(_GET_DEVICE())


def require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise S2THipError("S2T kernels need device tensors: the hot path has no CPU fallback")
