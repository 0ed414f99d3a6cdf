"""Host-side batch assembly for the S2T path: dictionary, collaters, synthetic batches.

  Dictionary                      fairseq/data/dictionary.py:18-60 (specials <s>=0 <pad>=1 </s>=2 <unk>=3)
  Seq2SeqCollater.collate         examples/speech_recognition/data/collaters.py:21-131
  TranscriptionWrapper collater   examples/speech_recognition/data/transcription_dataset.py:22-64
"""
import numpy as np
import torch


class Dictionary:
    def __init__(self, pad="<pad>", eos="</s>", unk="<unk>", bos="<s>"):
        self.symbols, self.indices = [], {}
        self.bos_index = self.add_symbol(bos)
        self.pad_index = self.add_symbol(pad)
        self.eos_index = self.add_symbol(eos)
        self.unk_index = self.add_symbol(unk)
        self.nspecial = len(self.symbols)

    def add_symbol(self, word):
        if word in self.indices:
            return self.indices[word]
        self.indices[word] = len(self.symbols)
        self.symbols.append(word)
        return self.indices[word]

    def index(self, sym):
        return self.indices.get(sym, self.unk_index)

    def __len__(self):
        return len(self.symbols)

    def __getitem__(self, i):
        return self.symbols[i] if i < len(self.symbols) else self.symbols[self.unk_index]

    def pad(self):
        return self.pad_index

    def eos(self):
        return self.eos_index

    def unk(self):
        return self.unk_index

    def bos(self):
        return self.bos_index

    @classmethod
    def load(cls, path):
        """`dict.<lang>.txt`: one `<symbol> <count>` per line (fairseq/data/dictionary.py:196-253)."""
        d = cls()
        with open(path, "r", encoding="utf-8") as f:
            for line in f:
                w = line.rstrip().rsplit(" ", 1)[0]
                if w:
                    d.add_symbol(w)
        return d

    @classmethod
    def synthetic(cls, n_words):
        d = cls()
        for i in range(n_words):
            d.add_symbol("w%d" % i)
        return d

    def unk_string(self, escape=False):
        return "<%s>" % self.symbols[self.unk_index] if escape else self.symbols[self.unk_index]

    def string(self, tensor, bpe_symbol=None, escape_unk=False, extra_symbols_to_ignore=None):
        """fairseq/data/dictionary.py:59-94: token ids -> text, EOS (and BOS) dropped, optional BPE continuation marker removed"""
        if torch.is_tensor(tensor) and tensor.dim() == 2:
            return "\n".join(self.string(t, bpe_symbol, escape_unk, extra_symbols_to_ignore) for t in tensor)
        skip = set(extra_symbols_to_ignore or ())
        skip.add(self.eos_index)
        words = [self.unk_string(escape_unk) if int(i) == self.unk_index else self[int(i)]
                 for i in tensor if int(i) not in skip and int(i) != self.bos_index]
        sent = " ".join(words)
        if bpe_symbol == "sentencepiece":
            sent = sent.replace(" ", "").replace("\u2581", " ").strip()
        elif bpe_symbol is not None:
            sent = (sent + " ").replace(bpe_symbol, "").rstrip()
        return sent


_StandaloneDictionary = Dictionary
try:
    from .registry import inside_fairseq as _inside_fairseq
    if _inside_fairseq():
        # Inside a fairseq process the dictionaries are fairseq's own class: `generate.main` post-processes hypotheses through
        # `tgt_dict.string / encode_line / unk_string` and the BLEU scorer (fairseq_cli/generate.py:160-230), checkpoints pickle
        # nothing of it, and nothing on the hot path reads more than pad / eos / len / index.
        from fairseq.data import Dictionary as _FairseqDictionary

        class Dictionary(_FairseqDictionary):                                    # noqa: F811
            @classmethod
            def synthetic(cls, n_words):
                d = cls()
                for i in range(n_words):
                    d.add_symbol("w%d" % i)
                return d
except ImportError:                                                              # pragma: no cover
    pass


def collate_tokens(values, pad_idx, eos_idx, move_eos_to_beginning=False):
    """fairseq/data/data_utils.py:collate_tokens with left_pad=False."""
    size = max(v.size(0) for v in values)
    res = values[0].new(len(values), size).fill_(pad_idx)
    for i, v in enumerate(values):
        if move_eos_to_beginning:
            assert v[-1] == eos_idx
            res[i, 0] = eos_idx
            res[i, 1:len(v)] = v[:-1]
        else:
            res[i, :len(v)] = v
    return res


class Seq2SeqCollater:
    """collaters.py:21-131: zero-pad frames, sort by frame count (descending), eos-shifted decoder input."""

    def __init__(self, feature_index=0, label_index=1, pad_index=1, eos_index=2, move_eos_to_beginning=True):
        self.feature_index, self.label_index = feature_index, label_index
        self.pad_index, self.eos_index, self.move_eos_to_beginning = pad_index, eos_index, move_eos_to_beginning

    def collate(self, samples):
        if len(samples) == 0:
            return {}
        parsed = []
        for s in samples:
            src = s["data"][self.feature_index]
            if src is None:
                continue
            if isinstance(src, np.ndarray):
                src = torch.from_numpy(src)
            tgt = s["data"][self.label_index]
            if isinstance(tgt, np.ndarray):
                tgt = torch.from_numpy(tgt).long()
            elif isinstance(tgt, list):
                tgt = torch.LongTensor(tgt)
            parsed.append({"id": s["id"], "source": src, "target": tgt})
        ids = torch.LongTensor([s["id"] for s in parsed])
        lens = torch.LongTensor([s["source"].size(0) for s in parsed])
        f_dim = parsed[0]["source"].size(1)
        frames = parsed[0]["source"].new_zeros(len(parsed), int(lens.max()), f_dim)
        for i, s in enumerate(parsed):
            frames[i, : s["source"].size(0)] = s["source"]
        lens, order = lens.sort(descending=True, stable=True)
        ids, frames = ids.index_select(0, order), frames.index_select(0, order)
        target = target_lengths = prev = None
        if parsed[0].get("target") is not None:
            ntokens = sum(len(s["target"]) for s in parsed)
            tl = [s["target"] for s in parsed]
            target = collate_tokens(tl, self.pad_index, self.eos_index).index_select(0, order)
            target_lengths = torch.LongTensor([t.size(0) for t in tl]).index_select(0, order)
            prev = collate_tokens(tl, self.pad_index, self.eos_index, self.move_eos_to_beginning).index_select(0, order)
        else:
            ntokens = sum(len(s["source"]) for s in parsed)
        batch = {"id": ids, "ntokens": ntokens, "net_input": {"src_tokens": frames, "src_lengths": lens},
                 "target": target, "target_lengths": target_lengths, "nsentences": len(parsed)}
        if prev is not None:
            batch["net_input"]["prev_output_tokens"] = prev
        return batch


def collate_with_transcripts(collater, samples, pad, eos):
    """TranscriptionWrapperDataset.collater (transcription_dataset.py:22-64)."""
    batch = collater.collate(samples)
    if len(batch) == 0:
        return {}
    pos = {s["id"]: i for i, s in enumerate(samples)}
    order = torch.tensor([pos[i] for i in batch["id"].tolist()])
    tr = [s["transcript_target"] for s in samples]
    batch["transcript_target"] = collate_tokens(tr, pad, eos).index_select(0, order)
    batch["transcript_target_lengths"] = torch.LongTensor([t.shape[0] for t in tr]).index_select(0, order)
    batch["net_input"]["transcript_prev_output_tokens"] = collate_tokens(tr, pad, eos, True).index_select(0, order)
    return batch


def synthetic_batch(B, T, L, Lt, V_tgt, V_src_blank, feat=80, seed=0, lengths=None):
    """MuST-C-shaped synthetic mini-batch: N(0,1) features (what per-utterance CMVN yields),
    zero in the padded tail; uniform target / transcript tokens in [4, V) ending with EOS.
    V_src_blank = index of <ctc_blank> (transcripts never contain it)."""
    g = torch.Generator().manual_seed(seed)
    lengths = [T] * B if lengths is None else sorted(lengths, reverse=True)
    Tm = max(lengths)
    x = torch.zeros(B, Tm, feat)
    for b, l in enumerate(lengths):
        x[b, :l] = torch.randn(l, feat, generator=g)

    def toks(n, hi):
        t = torch.randint(4, hi, (B, n), generator=g)
        t[:, -1] = 2
        prev = torch.cat([torch.full((B, 1), 2, dtype=torch.long), t[:, :-1]], 1)
        return t, prev
    tgt, prev = toks(L, V_tgt)
    tr, trp = toks(Lt, V_src_blank)
    return {"id": torch.arange(B), "ntokens": B * L, "nsentences": B, "nframes": int(sum(lengths)),
            "net_input": {"src_tokens": x, "src_lengths": torch.tensor(lengths), "prev_output_tokens": prev},
            "target": tgt, "target_lengths": torch.full((B,), L), "transcript_target": tr,
            "transcript_target_lengths": torch.full((B,), Lt)}


class SyntheticS2TDataset:
    """In-memory MuST-C-shaped utterances behind the dataset interface the batch iterator reads: N(0,1)
    filterbanks (what per-utterance CMVN yields), uniform target / transcript tokens ending in EOS.  `lengths`: frames per
    utterance (fixed, or e.g. the lognormal of Cfg4).  Items have the layout of FilterBankToTextDataset wrapped by
    TranscriptionWrapperDataset (indexed.py), so collation is the product collater (Seq2SeqCollater + transcripts)."""

    def __init__(self, lengths, tgt_len, transcript_len, V_tgt, V_src_blank, feat=80, seed=0, pad=1, eos=2):
        g = torch.Generator().manual_seed(seed)
        self.lengths = np.asarray(lengths, dtype=np.int64)
        self.pad, self.eos = pad, eos
        self.feats = [torch.randn(int(l), feat, generator=g) for l in self.lengths]

        def toks(n, hi):
            t = torch.randint(4, hi, (len(self.lengths), n), generator=g)
            t[:, -1] = eos
            return t
        # target length follows the utterance (L ~ T / 25) unless a fixed one is asked for
        self.tgt = toks(tgt_len, V_tgt) if tgt_len > 0 else None
        self.tr = toks(transcript_len, V_src_blank)
        if self.tgt is None:
            self.tgt_list = []
            for l in self.lengths:
                n = max(2, int(l) // 25)
                t = torch.randint(4, V_tgt, (n,), generator=g); t[-1] = eos
                self.tgt_list.append(t)
        self.collate = Seq2SeqCollater(0, 1, pad, eos, True)

    def __len__(self):
        return len(self.lengths)

    def __getitem__(self, i):
        tgt = self.tgt[i] if self.tgt is not None else self.tgt_list[i]
        return {"id": i, "data": [self.feats[i], tgt], "transcript_target": self.tr[i]}

    def collater(self, samples):
        b = collate_with_transcripts(self.collate, samples, self.pad, self.eos)
        if len(b):
            b["nframes"] = int(b["net_input"]["src_lengths"].sum())
        return b

    def num_tokens(self, i):
        return int(self.lengths[i])

    def size(self, i):
        return (int(self.lengths[i]), int((self.tgt[i] if self.tgt is not None else self.tgt_list[i]).shape[0]))

    @property
    def frame_lengths(self):
        return self.lengths

    def ordered_indices(self):
        return np.arange(len(self), dtype=np.int64)              # as the reference's fbank datasets (fbank_dataset.py:78-81)
