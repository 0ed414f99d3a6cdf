"""ORACLE (test infrastructure only): integer parts of the S2T hot path.

Two independent restatements that the tests compare with each other, with the
golden vectors and with the HIP kernels:

  * ``*_np``  -- numpy / pure-Python loops (small cases)
  * ``*_c``   -- ctypes calls into oracle/_build/liboracle_int.so (int_ref.c)

Reference lines are cited per function.
"""
import ctypes
import os
import subprocess
from itertools import groupby

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_int.so")
_lib = None


def build_c(force=False):
    """Compile int_ref.c with gcc (called from __graft_entry__.build())."""
    src = os.path.join(_HERE, "int_ref.c")
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _SO, src])
    return _SO


def _c():
    global _lib
    if _lib is None:
        build_c()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_ctc_rle.restype = ctypes.c_int64
        _lib.orc_argmax_first.restype = ctypes.c_int32
        _lib.orc_align_errors.restype = ctypes.c_int32
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ---------------------------------------------------------------- CTC compression
def ctc_rle_np(pred, lengths):
    """conv_transformer.py:283-287: groupby over the first lengths[b] frames.

    pred (B,T) int, lengths (B,) -> list over b of [(tok, run), ...]."""
    out = []
    for b in range(pred.shape[0]):
        p = [int(v) for v in pred[b, : int(lengths[b])]]
        out.append([(k, len(list(g))) for k, g in groupby(p)])
    return out


def ctc_rle_c(pred, lengths):
    pred = np.ascontiguousarray(pred, dtype=np.int32)
    lengths = np.ascontiguousarray(lengths, dtype=np.int64)
    B, T = pred.shape
    run_tok = np.empty((B, T), np.int32)
    run_len = np.empty((B, T), np.int32)
    run_start = np.empty((B, T), np.int32)
    seg_id = np.empty((B, T), np.int32)
    new_len = np.empty((B,), np.int64)
    mx = _c().orc_ctc_rle(_p(pred), _p(lengths), ctypes.c_int32(B), ctypes.c_int32(T),
                          _p(run_tok), _p(run_len), _p(run_start), _p(seg_id), _p(new_len))
    return dict(run_tok=run_tok, run_len=run_len, run_start=run_start, seg_id=seg_id,
                new_len=new_len, max_len=int(mx))


def argmax_first_np(rows):
    """First maximal index per row (torch.argmax CPU tie rule; conv_transformer.py:284)."""
    rows = np.asarray(rows)
    return np.argmax(rows, axis=-1).astype(np.int32)   # numpy: first occurrence


def argmax_first_c(rows):
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    flat = rows.reshape(-1, rows.shape[-1])
    out = np.empty((flat.shape[0],), np.int32)
    lib = _c()
    for i in range(flat.shape[0]):
        out[i] = lib.orc_argmax_first(_p(flat[i]), ctypes.c_int32(flat.shape[1]))
    return out.reshape(rows.shape[:-1])


def compress_weights_np(prob, runs, strategy="avg", dtype=np.float32):
    """CTCCompressStrategy.{avg,weighted,softmax}, conv_transformer.py:385-426.

    prob (B,T,V) softmax output, runs = ctc_rle_np(...).  Returns W (B,T,T''max)."""
    B, T = prob.shape[0], prob.shape[1]
    new_max = max(len(r) for r in runs)
    W = np.zeros((B, T, new_max), dtype)
    for b, r in enumerate(runs):
        s = 0
        for j, (tok, n) in enumerate(r):
            if strategy == "avg":
                W[b, s:s + n, j] = 1.0 / n          # python double, rounded on store (:394)
            else:
                w = prob[b, s:s + n, tok].astype(np.float32)
                if strategy == "softmax":
                    e = np.exp(w - w.max())
                    w = (e / e.sum()).astype(np.float32)
                W[b, s:s + n, j] = w / w.sum()
            s += n
    return W


# ---------------------------------------------------------------- CTC unit error rate
def ctc_greedy_np(pred, blank):
    """CTC_loss.py:50-58: collapse repeats, drop blanks."""
    ded = [k for k, _ in groupby([int(v) for v in pred])]
    return [p for p in ded if p != blank]


def align_errors_np(refs, hyps):
    """wer_utils.py:71-203 (time_mediated=False) + CTC_loss.py:61-72."""
    nr, nh = len(refs), len(hyps)
    if nr == 0 and nh == 0:
        return 0
    sc = np.zeros((nr + 1, nh + 1))
    bt = np.zeros((nr + 1, nh + 1), np.int64)
    cols = nh + 1
    for i in range(nr + 1):
        for j in range(nh + 1):
            if i == 0 and j == 0:
                continue
            if i == 0:
                sc[i, j] = sc[i, j - 1] + 3; bt[i, j] = i * cols + j - 1; continue
            if j == 0:
                sc[i, j] = sc[i - 1, j] + 3; bt[i, j] = (i - 1) * cols + j; continue
            best = sc[i - 1, j - 1] + (0 if refs[i - 1] == hyps[j - 1] else 4)
            prev = (i - 1) * cols + j - 1
            ins = sc[i, j - 1] + 3
            if ins < best:
                best, prev = ins, i * cols + j - 1
            dele = sc[i - 1, j] + 3
            if dele < best:
                best, prev = dele, (i - 1) * cols + j
            sc[i, j] = best; bt[i, j] = prev
    errors, cur = 0, (nr + 1) * cols - 1
    while cur != 0:
        prev = int(bt[cur // cols, cur % cols])
        cr, cc, pr, pc = cur // cols, cur % cols, prev // cols, prev % cols
        if cr - 1 == pr and cc - 1 == pc:
            errors += int(refs[cr - 1] != hyps[cc - 1])
        else:
            errors += 1
        cur = prev
    return errors


def ctc_uer_np(pred, input_len, targets, target_len, blank):
    """compute_ctc_uer, CTC_loss.py:31-74.  pred (B,T) argmax ids."""
    e = n = 0.0
    for b in range(pred.shape[0]):
        dec = ctc_greedy_np(pred[b, : int(input_len[b])], blank)
        tgt = [int(v) for v in targets[b, : int(target_len[b])]]
        e += align_errors_np(dec, tgt)
        n += len(tgt)
    return e, n


def ctc_uer_c(pred, input_len, targets, target_len, blank):
    pred = np.ascontiguousarray(pred, dtype=np.int32)
    input_len = np.ascontiguousarray(input_len, dtype=np.int64)
    targets = np.ascontiguousarray(targets, dtype=np.int64)
    target_len = np.ascontiguousarray(target_len, dtype=np.int64)
    e, n = ctypes.c_double(0), ctypes.c_double(0)
    _c().orc_ctc_uer(_p(pred), _p(input_len), ctypes.c_int32(pred.shape[0]), ctypes.c_int32(pred.shape[1]),
                     _p(targets), _p(target_len), ctypes.c_int32(targets.shape[1]), ctypes.c_int32(blank),
                     ctypes.byref(e), ctypes.byref(n))
    return e.value, n.value


# ---------------------------------------------------------------- collate
def sort_desc_c(lengths):
    lengths = np.ascontiguousarray(lengths, dtype=np.int64)
    order = np.empty_like(lengths)
    _c().orc_sort_desc(_p(lengths), ctypes.c_int32(len(lengths)), _p(order))
    return order


def batch_by_size(indices, lens, max_tokens=-1, max_sentences=-1, bsz_mult=1):
    """fairseq/data/data_utils_fast.pyx:16-68 batch_by_size_fast, restated with plain lists (lens[idx] = frames of item idx)."""
    batches, batch, blen, sample_len = [], [], [], 0
    for idx in [int(i) for i in indices]:
        nt = int(lens[idx])
        blen.append(nt)
        sample_len = max(sample_len, nt)
        assert max_tokens <= 0 or sample_len <= max_tokens, "sentence at index {} of size {} exceeds max_tokens limit of {}!".format(idx, sample_len, max_tokens)
        num_tokens = (len(batch) + 1) * sample_len
        full = len(batch) > 0 and ((max_sentences > 0 and len(batch) == max_sentences) or (max_tokens > 0 and num_tokens > max_tokens))
        if full:
            mod_len = max(bsz_mult * (len(batch) // bsz_mult), len(batch) % bsz_mult)
            batches.append(batch[:mod_len])
            batch, blen = batch[mod_len:], blen[mod_len:]
            sample_len = max(blen) if blen else 0
        batch.append(idx)
    if batch:
        batches.append(batch)
    return batches


def apply_augment(x, row_map=None, fmask=None, tmask=None):
    """numpy application of the augmentation tables (s2t_augment's contract): x [B,T,F] -> [B,To,F]; row gather (-1 = zero row),
    then zero the (t0, width) rows and (f0, width) columns of every utterance (time_stretch.py:42-57, specaugment.py:95-110)."""
    B, T, F = x.shape
    if row_map is None:
        out = x.copy()
    else:
        out = np.zeros((B, row_map.shape[1], F), x.dtype)
        for b in range(B):
            ok = row_map[b] >= 0
            out[b, ok] = x[b, row_map[b][ok]]
    for b in range(B):
        if fmask is not None:
            for f0, w in fmask[b]:
                out[b, :, f0:f0 + w] = 0
        if tmask is not None:
            for t0, w in tmask[b]:
                out[b, t0:t0 + w, :] = 0
    return out
