"""On-disk data of the S2T path: the TorchNet indexed format and frame-budget batching.

File format (fairseq/data/indexed_dataset.py:110-133 reader, :277-343 writer), little-endian:

    <prefix>.idx   "TNTIDX\\0\\0" | u64 version = 1 | u64 dtype code | u64 element size | u64 n_items | u64 n_sizes |
                   i64 dim_offsets[n_items + 1] | i64 data_offsets[n_items + 1] (in elements) | i64 sizes[n_sizes]
    <prefix>.bin   the items back to back, row-major

dtype codes 1..8 = uint8, int8, int16, int32, int64, float(=f64 in numpy, written with element size 4 by the reference:
:281-289), double, float32.  Filterbank datasets hold float32 items of two dimensions (frames, features)
(examples/speech_recognition/data/fbank_dataset.py:97-152, writer preprocess_audio.py:44-58); token datasets hold int32
items stored +1 ("Lua indexing", :305-307) that the reader shifts back when `fix_lua_indexing` is set.

MI355X-first choices: the .bin file is memory-mapped once (items are zero-copy views until they are pinned for the
H2D copy) instead of seek+read per item, and batches are formed by the native `s2t_host_batch_by_size`.
"""
import os
import struct

import numpy as np
import torch

from . import lib as L
from .registry import FairseqDataset

_MAGIC = b"TNTIDX\x00\x00"
_DTYPES = {1: np.uint8, 2: np.int8, 3: np.int16, 4: np.int32, 5: np.int64, 6: np.float64, 7: np.float64, 8: np.float32}
_CODES = {np.dtype(np.uint8): 1, np.dtype(np.int8): 2, np.dtype(np.int16): 3, np.dtype(np.int32): 4, np.dtype(np.int64): 5,
          np.dtype(np.float64): 7, np.dtype(np.float32): 8}


def index_file_path(prefix):
    return prefix + ".idx"


def data_file_path(prefix):
    return prefix + ".bin"


class IndexedDataset:
    """Reader (IndexedDataset / IndexedCachedDataset of the reference: same items, sizes and errors)."""

    def __init__(self, path, fix_lua_indexing=False):
        self.path = path
        self.fix_lua_indexing = fix_lua_indexing
        with open(index_file_path(path), "rb") as f:
            magic = f.read(8)
            assert magic == _MAGIC, ("Index file doesn't match expected format. "
                                     "Make sure that --dataset-impl is configured properly.")
            assert struct.unpack("<Q", f.read(8)) == (1,)
            code, self.element_size = struct.unpack("<QQ", f.read(16))
            self.dtype = np.dtype(_DTYPES[code])
            if self.dtype.itemsize != self.element_size:          # code 6 ("np.float") is written with 4-byte elements
                self.dtype = np.dtype(np.float32) if self.element_size == 4 else self.dtype
            self._len, self.s = struct.unpack("<QQ", f.read(16))
            self.dim_offsets = np.fromfile(f, dtype="<i8", count=self._len + 1)
            self.data_offsets = np.fromfile(f, dtype="<i8", count=self._len + 1)
            self.sizes = np.fromfile(f, dtype="<i8", count=self.s)
        self._data = None

    @staticmethod
    def exists(path):
        return os.path.exists(index_file_path(path)) and os.path.exists(data_file_path(path))

    def _map(self):
        if self._data is None:
            self._data = np.memmap(data_file_path(self.path), dtype=self.dtype, mode="r")
        return self._data

    def check_index(self, i):
        if i < 0 or i >= self._len:
            raise IndexError("index out of range")

    def item_view(self, i):
        """zero-copy numpy view of item i, shaped by its stored dimensions"""
        self.check_index(i)
        shape = tuple(int(v) for v in self.sizes[self.dim_offsets[i]:self.dim_offsets[i + 1]])
        return self._map()[self.data_offsets[i]:self.data_offsets[i + 1]].reshape(shape)

    def __getitem__(self, i):
        item = torch.from_numpy(np.array(self.item_view(i))).long()
        if self.fix_lua_indexing:
            item -= 1
        return item

    def __len__(self):
        return self._len

    def num_tokens(self, index):
        return self.sizes[index]

    def size(self, index):
        return self.sizes[index]

    @property
    def supports_prefetch(self):
        return False                                     # the mapping makes prefetching a no-op

    def prefetch(self, indices):
        pass


class FilterBanksDataset(IndexedDataset):
    """float32 items of two dimensions (frames, features); fbank_dataset.py:97-152."""

    def __init__(self, path, cached=True, legacy_audio_fix_lua_indexing=False):
        super().__init__(path)
        self.cached = cached
        self.legacy_audio_fix_lua_indexing = legacy_audio_fix_lua_indexing
        assert self.dtype == np.float32
        assert len(self.sizes) == len(self) * 2

    def __getitem__(self, i):
        item = torch.from_numpy(np.array(self.item_view(i)))
        return item - 1 if self.legacy_audio_fix_lua_indexing else item

    def num_tokens(self, index):
        return self.sizes[index * 2]                     # frames

    def size(self, index):
        return self.sizes[index * 2]

    @property
    def frame_lengths(self):
        """int64 [n_items]: frames per utterance (what frame-budget batching consumes)"""
        return np.ascontiguousarray(self.sizes[0::2])


class IndexedDatasetBuilder:
    """Writer; byte-identical files to the reference's builders (indexed_dataset.py:277-343, preprocess_audio.py:44-58)."""

    def __init__(self, out_file, dtype=np.int32, lua_offset=1):
        self.out_file = open(out_file, "wb")
        self.dtype = np.dtype(dtype)
        self.lua_offset = lua_offset                     # token datasets are stored +1
        self.data_offsets, self.dim_offsets, self.sizes = [0], [0], []
        self.element_size = self.dtype.itemsize

    def add_item(self, tensor):
        a = tensor.numpy() if torch.is_tensor(tensor) else np.asarray(tensor)
        n = self.out_file.write(np.ascontiguousarray(a + self.lua_offset if self.lua_offset else a, dtype=self.dtype).tobytes())
        self.data_offsets.append(self.data_offsets[-1] + n // self.element_size)
        self.sizes.extend(int(s) for s in a.shape)
        self.dim_offsets.append(self.dim_offsets[-1] + a.ndim)

    def finalize(self, index_file):
        self.out_file.close()
        with open(index_file, "wb") as f:
            f.write(_MAGIC)
            f.write(struct.pack("<Q", 1))
            f.write(struct.pack("<QQ", _CODES[self.dtype], self.element_size))
            f.write(struct.pack("<QQ", len(self.data_offsets) - 1, len(self.sizes)))
            for arr in (self.dim_offsets, self.data_offsets, self.sizes):
                f.write(np.array(arr, dtype="<i8").tobytes())


class AudioIndexedDatasetBuilder(IndexedDatasetBuilder):
    def __init__(self, out_file, fix_lua_indexing=False):
        super().__init__(out_file, dtype=np.float32, lua_offset=1 if fix_lua_indexing else 0)


def batch_by_size(indices, num_tokens_fn_or_lens, max_tokens=None, max_sentences=None, required_batch_size_multiple=1):
    """fairseq/data/data_utils.py:200-234 on the native batcher.  `num_tokens_fn_or_lens`: an int64 array of lengths indexed by
    dataset index (preferred: no Python call per utterance) or the reference's callable."""
    indices = np.ascontiguousarray(np.fromiter(indices, dtype=np.int64) if not isinstance(indices, np.ndarray) else indices, dtype=np.int64)
    if callable(num_tokens_fn_or_lens):
        n = int(indices.max()) + 1 if indices.size else 0
        lens = np.zeros(n, np.int64)
        for i in indices:
            lens[i] = num_tokens_fn_or_lens(int(i))
    else:
        lens = np.ascontiguousarray(num_tokens_fn_or_lens, dtype=np.int64)
    n = indices.size
    if max_tokens is not None and max_tokens > 0 and n:
        over = np.nonzero(lens[indices] > max_tokens)[0]
        assert over.size == 0, "sentence at index {} of size {} exceeds max_tokens limit of {}!".format(
            int(indices[over[0]]) if over.size else -1, int(lens[indices[over[0]]]) if over.size else -1, max_tokens)
    flat = np.empty(max(n, 1), np.int64)
    offs = np.empty(n + 1, np.int64)
    nb = L.ctypes.c_longlong(0)
    rc = L.load().s2t_host_batch_by_size(indices.ctypes.data, n, lens.ctypes.data, -1 if max_tokens is None else int(max_tokens),
                                         -1 if max_sentences is None else int(max_sentences), int(required_batch_size_multiple),
                                         flat.ctypes.data, offs.ctypes.data, L.ctypes.addressof(nb))
    L.check(rc, "s2t_host_batch_by_size")
    return [flat[offs[b]:offs[b + 1]].tolist() for b in range(nb.value)]


# ------------------------------------------------------------------ dataset wrappers of the S2T task
def apply_mv_norm(features):
    """per-utterance mean / variance normalisation (examples/speech_recognition/data/data_utils.py:9-25)"""
    mean, var = features.mean(0), features.var(0)
    eps = 1e-8
    inv = 1.0 / (torch.sqrt(var) + eps) if bool((var < eps).any()) else 1.0 / torch.sqrt(var)
    return (features - mean) * inv


class FilterBankToTextDataset(FairseqDataset):
    """fbank_dataset.py:17-95: filterbank item + target token item -> {"id", "data": [frames, tokens]}"""

    def __init__(self, src_dataset, tgt_dataset, tgt_dict, skip_normalization=False):
        from .data import Seq2SeqCollater
        assert len(src_dataset) == len(tgt_dataset)
        self.src_dataset, self.tgt_dataset, self.tgt_dict = src_dataset, tgt_dataset, tgt_dict
        self.skip_normalization = skip_normalization
        self.s2s_collater = Seq2SeqCollater(0, 1, pad_index=tgt_dict.pad(), eos_index=tgt_dict.eos(), move_eos_to_beginning=True)

    def __getitem__(self, index):
        tgt = self.tgt_dataset[index] if self.tgt_dataset is not None else None
        src = self.src_dataset[index]
        if not self.skip_normalization:
            src = apply_mv_norm(src)
        return {"id": index, "data": [src, tgt]}

    def __len__(self):
        return len(self.src_dataset)

    def collater(self, samples):
        return self.s2s_collater.collate(samples)

    def num_tokens(self, index):
        return self.src_dataset.size(index)

    def size(self, index):
        return (self.src_dataset.size(index), self.tgt_dataset.size(index) if self.tgt_dataset is not None else 0)

    @property
    def frame_lengths(self):
        return self.src_dataset.frame_lengths

    def ordered_indices(self):
        return np.arange(len(self))                      # the reference does not sort filterbanks by length (fbank_dataset.py:78-81)


class TranscriptionWrapperDataset(FairseqDataset):
    """transcription_dataset.py:7-86: adds the source-language transcript of every utterance"""

    def __init__(self, tgt_dataset, transcription_dataset, transcription_dict):
        self.tgt_dataset, self.transcription_dataset, self.transcription_dict = tgt_dataset, transcription_dataset, transcription_dict

    def __getitem__(self, index):
        item = self.tgt_dataset[index]
        item["transcript_target"] = self.transcription_dataset[index]
        return item

    def __len__(self):
        return len(self.tgt_dataset)

    def collater(self, samples):
        from .data import collate_with_transcripts
        return collate_with_transcripts(self.tgt_dataset.s2s_collater, samples, self.transcription_dict.pad(), self.transcription_dict.eos())

    def num_tokens(self, index):
        return self.tgt_dataset.num_tokens(index)

    def size(self, index):
        return self.tgt_dataset.size(index)

    @property
    def frame_lengths(self):
        return self.tgt_dataset.frame_lengths

    def ordered_indices(self):
        return self.tgt_dataset.ordered_indices()


def load_s2t_split(data_path, split, src_lang, tgt_lang, src_dict, tgt_dict, skip_normalization=False,
                   legacy_audio_fix_lua_indexing=False):
    """<split>.npz.{idx,bin} filterbanks + <split>.<tgt> targets + <split>.<src> transcripts
    (speech_recognition.py:73-83, speech_translation_ctc.py:48-72; token files carry the +1 Lua offset)."""
    prefix = os.path.join(data_path, split)
    fb = FilterBanksDataset(prefix + ".npz", True, legacy_audio_fix_lua_indexing)
    tgt = IndexedDataset(prefix + "." + tgt_lang, fix_lua_indexing=True)
    ds = FilterBankToTextDataset(fb, tgt, tgt_dict, skip_normalization)
    tr = IndexedDataset(prefix + "." + src_lang, fix_lua_indexing=True)
    assert len(ds) == len(tr)
    return TranscriptionWrapperDataset(ds, tr, src_dict)


# ------------------------------------------------------------------ word-level knowledge distillation: teacher outputs on disk
class TeacherOutputDataset(IndexedDataset):
    """fairseq/data/knowledge_distillation.py:26-55: items [L, K] of teacher top-k columns (int32 -> long) or logits (float32)."""

    def __init__(self, prefix, dtype):
        super().__init__(prefix, fix_lua_indexing=False)
        self.dtype = np.dtype(dtype)

    @staticmethod
    def save_bin(prefix, data_list, dtype=np.float32):
        b = IndexedDatasetBuilder(prefix + ".bin", dtype, lua_offset=0)
        for d in data_list:
            b.add_item(np.array(d, dtype=dtype))
        b.finalize(prefix + ".idx")

    def __getitem__(self, i):
        item = torch.from_numpy(np.array(self.item_view(i)))
        return item.long() if self.dtype.kind in "iu" else item.float()


class DatasetWithTeacherOutput(FairseqDataset):
    """knowledge_distillation.py:58-153: adds `teacher_output` = [columns, logits] to every item and batch"""

    def __init__(self, src, teacher_probs, teacher_idxs, tgt_dict, distill_k):
        self.src, self.teacher_probs, self.teacher_idxs, self.tgt_dict, self.distill_k = src, teacher_probs, teacher_idxs, tgt_dict, distill_k

    def __getitem__(self, index):
        item = self.src[index]
        item["teacher_output"] = [self.teacher_idxs[index], self.teacher_probs[index]]
        return item

    def __len__(self):
        return len(self.src)

    def num_tokens(self, index):
        return self.src.num_tokens(index)

    def size(self, index):
        return self.src.size(index)

    @property
    def frame_lengths(self):
        return self.src.frame_lengths

    def ordered_indices(self):
        return self.src.ordered_indices()

    def collater(self, samples):
        batch = self.src.collater(samples)
        if len(samples) > 0:
            L_t = batch["target"].shape[1]
            pad = self.tgt_dict.pad()
            by_id = {}
            for s in samples:
                ti, tp = s["teacher_output"]
                by_id[s["id"]] = (torch.nn.functional.pad(ti, (0, 0, 0, L_t - ti.shape[0]), value=pad),
                                  torch.nn.functional.pad(tp, (0, 0, 0, L_t - tp.shape[0])))
            ids = batch["id"].tolist()
            batch["teacher_output"] = [torch.stack([by_id[i][0] for i in ids]), torch.stack([by_id[i][1] for i in ids])]
        return batch


def dump_teacher_topk(task, model, dataset, k, max_tokens=12000, max_sentences=None, max_positions=None, required_batch_size_multiple=8):
    """scripts/generate_topk.py:20-74 on the HIP engine: the teacher's top-k logits at every non-pad target position of `dataset`
    (target-forced forward in eval mode).  Returns [[columns per position], [logits per position]] per utterance, in dataset order."""
    from . import kernels as K
    from .iterators import get_batch_iterator
    itr = get_batch_iterator(dataset, max_tokens=max_tokens, max_sentences=max_sentences, max_positions=max_positions,
                             ignore_invalid_inputs=True, required_batch_size_multiple=required_batch_size_multiple).next_epoch_itr(shuffle=False)
    outputs = [None] * len(dataset)
    was_training = model.training
    model.eval()
    dev = model.device if hasattr(model, "device") else torch.device("cuda")
    try:
        with torch.no_grad():
            for s in itr:
                if "net_input" not in s:
                    continue
                ni = s["net_input"]
                enc = model.encoder(ni["src_tokens"].to(dev), ni["src_lengths"])
                logits, _ = model.decoder(ni["prev_output_tokens"].to(dev), encoder_out=enc)          # (B, L, V) view of [L*B, V] rows
                B, Lt, V = logits.shape
                rows = logits.transpose(0, 1).reshape(Lt * B, V)                                        # time-major rows, padded stride kept
                vals, idx = K.topk(rows, k)                                                             # time-major rows
                vals = vals.view(Lt, B, k).transpose(0, 1).cpu().numpy(); idx = idx.view(Lt, B, k).transpose(0, 1).cpu().numpy()
                keep = s["target"].ne(task.target_dictionary.pad()).numpy().astype(bool)
                for i, id_s in enumerate(s["id"].tolist()):
                    outputs[id_s] = [idx[i, keep[i]].tolist(), vals[i, keep[i]].tolist()]
    finally:
        model.train(was_training)
    return outputs
