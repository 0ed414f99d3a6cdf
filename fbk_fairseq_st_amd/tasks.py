"""Tasks of the S2T path behind the reference's task interface.

  speech_translation_with_transcription   examples/speech_recognition/tasks/speech_translation_ctc.py:18-79
                                           (target dictionary + source dictionary with <ctc_blank> appended)
  dummy_s2t                                synthetic-filterbank benchmark task, after the pattern of
                                           fairseq/benchmark/dummy_lm.py:50-119 (one pre-built batch forever)
"""
import os

from .data import Dictionary, synthetic_batch
from .registry import FairseqTask, register_task


@register_task("speech_translation_with_transcription")
class SpeechTranslationCTCTask(FairseqTask):
    @staticmethod
    def add_args(parser):
        a = parser.add_argument
        a("data", help="path to data directory")
        a("-s", "--source-lang", default=None, metavar="SRC")
        a("-t", "--target-lang", default=None, metavar="TARGET")
        a("--max-source-positions", default=1024, type=int, metavar="N")
        a("--max-target-positions", default=1024, type=int, metavar="N")
        a("--skip-normalization", action="store_true")
        a("--bucket-by-length", action="store_true",
          help="build-defined: order utterances by frame count before batching (the reference batches in dataset order)")
        a("--legacy-audio-fix-lua-indexing", action="store_true")
        # speech_recognition.py:114-133
        a("--specaugment", action="store_true")
        a("--frequency-masking-pars", type=int, default=13)
        a("--time-masking-pars", type=int, default=13)
        a("--frequency-masking-num", type=int, default=2)
        a("--time-masking-num", type=int, default=2)
        a("--specaugment-rate", type=float, default=1.0)
        a("--time-stretch", action="store_true")
        a("--time-stretch-rate", type=float, default=1.0)
        a("--time-stretch-w", type=int, default=1)
        a("--time-stretch-low", type=float, default=0.8)
        a("--time-stretch-high", type=float, default=1.25)

    def __init__(self, args, tgt_dict, src_dict=None):
        super().__init__(args)
        self.tgt_dict = tgt_dict
        self.src_dict = src_dict
        self.is_source_speech = True                     # read by generate.py:61-67
        from .augment import SpecAugment, TimeStretch
        g = lambda k, d: getattr(args, k, d)
        self.specaugment = SpecAugment(g("frequency_masking_pars", 13), g("time_masking_pars", 13), g("frequency_masking_num", 2),
                                       g("time_masking_num", 2), g("specaugment_rate", 1.0)) if g("specaugment", False) else None
        self.time_stretch = TimeStretch(g("time_stretch_rate", 1.0), g("time_stretch_w", 1), g("time_stretch_low", 0.8),
                                        g("time_stretch_high", 1.25)) if g("time_stretch", False) else None

    def train_step(self, sample, model, criterion, optimizer, update_num, ignore_grad=False):
        """speech_recognition.py:254-263: TimeStretch, then SpecAugment, then the generic step"""
        if self.time_stretch is not None:
            sample = self.time_stretch(sample)
        if self.specaugment is not None:
            sample = self.specaugment(sample)
        return super().train_step(sample, model, criterion, optimizer, update_num, ignore_grad)

    @classmethod
    def setup_task(cls, args, **kwargs):
        """speech_translation_ctc.py:35-47: dict.<tgt>.txt, dict.<src>.txt (+ <ctc_blank> as last symbol)."""
        tgt = Dictionary.load(os.path.join(args.data, "dict.%s.txt" % args.target_lang))
        src = Dictionary.load(os.path.join(args.data, "dict.%s.txt" % args.source_lang))
        if getattr(args, "criterion", None) == "ctc_multi_loss":         # only then (speech_translation_ctc.py:44-45)
            src.add_symbol("<ctc_blank>")
        return cls(args, tgt, src)

    def load_dataset(self, split, combine=False, **kwargs):
        """speech_translation_ctc.py:48-72: <split>.npz filterbanks, <split>.<tgt> targets, <split>.<src> transcripts (TNTIDX files)"""
        from .indexed import load_s2t_split
        a = self.args
        self.datasets[split] = load_s2t_split(a.data.split(os.pathsep)[0], split, a.source_lang, a.target_lang, self.src_dict, self.tgt_dict,
                                              getattr(a, "skip_normalization", False), getattr(a, "legacy_audio_fix_lua_indexing", False))

    def get_batch_iterator(self, dataset, max_tokens=None, max_sentences=None, max_positions=None, ignore_invalid_inputs=False,
                           required_batch_size_multiple=1, seed=1, num_shards=1, shard_id=0, num_workers=0, epoch=1):
        """fairseq_task.py:107-199 on the native batcher + pinned-memory prefetch thread (iterators.py).  Inside a fairseq process
        the reference's own `train.main` drives the result (next_epoch_idx, GroupedIterator over a CountingIterator, state_dict /
        load_state_dict in checkpoints: fairseq/data/iterators.py:170-340), so there it IS fairseq's EpochBatchIterator, built by
        the inherited method from this package's datasets (same indices, same filter, same frame-budget batches: golden
        `iterator.npz`)."""
        from .registry import inside_fairseq
        if inside_fairseq():
            if getattr(self.args, "bucket_by_length", False):
                import warnings                 # say so instead of dropping the flag silently
                warnings.warn("--bucket-by-length (and the pinned-memory prefetch thread) belong to this package's own batch iterator; inside a "
                              "fairseq process the reference's EpochBatchIterator is used and the flag has no effect")
            return super().get_batch_iterator(dataset, max_tokens=max_tokens, max_sentences=max_sentences, max_positions=max_positions,
                                              ignore_invalid_inputs=ignore_invalid_inputs,
                                              required_batch_size_multiple=required_batch_size_multiple, seed=seed,
                                              num_shards=num_shards, shard_id=shard_id, num_workers=num_workers, epoch=epoch)
        from .iterators import get_batch_iterator
        return get_batch_iterator(dataset, max_tokens, max_sentences, max_positions, ignore_invalid_inputs, required_batch_size_multiple,
                                  seed, num_shards, shard_id, epoch, bucket_by_length=getattr(self.args, "bucket_by_length", False))

    @property
    def source_dictionary(self):
        return self.src_dict

    @property
    def target_dictionary(self):
        return self.tgt_dict

    def max_positions(self):
        return (getattr(self.args, "max_source_positions", 1024), getattr(self.args, "max_target_positions", 1024))


@register_task("speech_translation_dualdecoding")
class SpeechTranslationDualDecodingTask(SpeechTranslationCTCTask):
    """tasks/speech_translation_dualdecoding.py:16-38: the CTC task whose generator is the two-phase one (transcript first, then
    translation) for dual-decoder models."""

    def build_generator(self, models, args):
        from .sequence_generator import TwoPhaseSequenceGenerator
        return TwoPhaseSequenceGenerator(models, self.source_dictionary, self.target_dictionary, beam_size=getattr(args, "beam", 5),
                                         max_len_a=getattr(args, "max_len_a", 0), max_len_b=getattr(args, "max_len_b", 200),
                                         min_len=getattr(args, "min_len", 1), normalize_scores=(not getattr(args, "unnormalized", False)),
                                         len_penalty=getattr(args, "lenpen", 1), unk_penalty=getattr(args, "unkpen", 0),
                                         temperature=getattr(args, "temperature", 1.0),
                                         match_source_len=getattr(args, "match_source_len", False),
                                         no_repeat_ngram_size=getattr(args, "no_repeat_ngram_size", 0))


@register_task("dummy_s2t")
class DummyS2TTask(SpeechTranslationCTCTask):
    """Synthetic 80-mel filterbank batches of a fixed shape; dictionaries of the requested sizes."""

    @staticmethod
    def add_args(parser):
        a = parser.add_argument
        a("--dict-size", default=8000 - 4, type=int)
        a("--src-dict-size", default=5000 - 4, type=int)
        a("--batch-size", default=8, type=int)
        a("--frames", default=1500, type=int)
        a("--tgt-len", default=40, type=int)
        a("--transcript-len", default=40, type=int)

    @classmethod
    def setup_task(cls, args, **kwargs):
        tgt = Dictionary.synthetic(getattr(args, "dict_size", 7996))
        src = Dictionary.synthetic(getattr(args, "src_dict_size", 4996))
        src.add_symbol("<ctc_blank>")
        return cls(args, tgt, src)

    def load_dataset(self, split, combine=False, n_utterances=256, lengths=None, seed=0, **kwargs):
        """synthetic utterances behind the real dataset / iterator interface (bench.py --loader: collate, pinning and the H2D
        copy then sit inside the timed loop)"""
        from .data import SyntheticS2TDataset
        a = self.args
        if lengths is None:
            lengths = [a.frames] * n_utterances
        self.datasets[split] = SyntheticS2TDataset(lengths, a.tgt_len, a.transcript_len, len(self.tgt_dict),
                                                   self.src_dict.index("<ctc_blank>"), feat=getattr(a, "input_feat_per_channel", 80),
                                                   seed=seed)

    def dummy_batch(self, seed=0, lengths=None):
        a = self.args
        return synthetic_batch(a.batch_size, a.frames, a.tgt_len, a.transcript_len, len(self.tgt_dict),
                               self.src_dict.index("<ctc_blank>"), feat=getattr(a, "input_feat_per_channel", 80),
                               seed=seed, lengths=lengths)
