"""Beam-search generation on the HIP incremental decoder.

Mirrors the observable behaviour of the reference's `SequenceGenerator` with `BeamSearch`
(fairseq/sequence_generator.py:26-600, fairseq/search.py:50-85): same constructor arguments, same
`generate(models, sample)` result (per sentence a list of hypotheses sorted by score, each a dict with
`tokens`, `score`, `positional_scores`, `attention`, `alignment`), same candidate rules (2*beam
candidates per step, EOS only finalised from the top `beam`, length-normalised scores, forced EOS at
`max_len`, `min_len`, unk penalty, temperature).

Design differences (results identical): finished sentences are masked instead of compacted out of the
batch, so the decoder state never needs the `batch_idxs` re-indexing and the encoder-side K/V are never
re-gathered; every floating-point operation of a step (embedding, the six decoder layers with
in-place K/V rows, output projection, log-softmax) is a libs2t_hip.so kernel; torch is used for the
integer selection bookkeeping (top-k, gathers of token/score rows) exactly as the reference does.
Ensembles (EnsembleModel.forward_decoder :711-770: every member runs its own encoder and incremental decoder, the
log-probabilities meet in one logsumexp kernel), prefix tokens (:270-280,449-476) and n-gram blocking (:617-650) follow the
reference; with `retain_attention=True` every hypothesis carries its `attention` (src_len x tgt_len, the last decoder layer's
encoder-attention averaged over heads and ensemble members: sequence_generator.py:286-292,510-560,757-768) and, with `print_alignment`, the
hard `alignment` generate.py prints (utils.extract_hard_alignment, fairseq/utils.py); sampling is not part of this path.  `TwoPhaseSequenceGenerator` runs the same loop twice for dual-decoder
models: transcripts with the auxiliary decoder, then translations seeded by the transcript scores.
"""
import math

import torch

from . import kernels as K


class BeamSearch:
    """fairseq/search.py:50-85: candidates of one step = top 2*beam of (cumulative score + log-prob) over beam x vocab."""

    def __init__(self, tgt_dict):
        self.pad, self.unk, self.eos = tgt_dict.pad(), tgt_dict.unk(), tgt_dict.eos()
        self.vocab_size = len(tgt_dict)

    def step(self, step, lprobs, scores):
        """lprobs f32 [B, beam, V]; scores [B, beam, >=step] cumulative.  Returns (scores, token ids, beam ids), each [B, k]."""
        B, beam, V = lprobs.shape
        if step == 0:
            cand = lprobs[:, 0, :]                                        # all beams are identical copies of <eos>
        else:
            cand = (lprobs + scores[:, :, step - 1].unsqueeze(-1)).view(B, beam * V)
        k = min(2 * beam, cand.shape[1] - 1)                              # -1: pad is never selected (search.py:71-75)
        top_s, top_i = torch.topk(cand, k)
        return top_s, top_i % V, top_i // V


class HierarchicalBeamSearch(BeamSearch):
    """twophase_sequence_generator.py:17-49: at step 0 every beam slot competes, each starting from the score handed over by the
    previous search (`prev_scores` [B, beam, 1]); later steps are BeamSearch.step."""

    def step(self, step, lprobs, scores, prev_scores=None):
        if step > 0 or prev_scores is None:
            return super().step(step, lprobs, scores)
        B, beam, V = lprobs.shape
        cand = (lprobs + prev_scores).view(B, beam * V)
        k = min(2 * beam, cand.shape[1] - 1)
        top_s, top_i = torch.topk(cand, k)
        return top_s, top_i % V, top_i // V


class SequenceGenerator:
    def __init__(self, models, tgt_dict, beam_size=1, max_len_a=0, max_len_b=200, min_len=1, normalize_scores=True, len_penalty=1.0,
                 unk_penalty=0.0, retain_dropout=False, temperature=1.0, match_source_len=False, no_repeat_ngram_size=0,
                 search_strategy=None, eos=None, retain_attention=False, print_alignment=False):
        self.models = list(models) if isinstance(models, (list, tuple)) else [models]
        if not 1 <= len(self.models) <= 8:
            raise ValueError("an ensemble of 1..8 models (s2t_ensemble_lse), got %d" % len(self.models))
        self.pad, self.unk = tgt_dict.pad(), tgt_dict.unk()
        self.eos = tgt_dict.eos() if eos is None else eos
        self.vocab_size = len(tgt_dict)
        self.beam_size = min(beam_size, self.vocab_size - 1)              # sequence_generator.py:69
        self.max_len_a, self.max_len_b, self.min_len = max_len_a, max_len_b, min_len
        self.normalize_scores, self.len_penalty, self.unk_penalty = normalize_scores, len_penalty, unk_penalty
        self.temperature = temperature
        assert temperature > 0, "--temperature must be greater than 0"
        if match_source_len or retain_dropout:
            raise NotImplementedError("match_source_len / retain_dropout are outside the S2T hot path")
        self.no_repeat_ngram_size = int(no_repeat_ngram_size)
        # the reference records the averaged encoder attention of every step whenever the decoder returns one (:286-292); here it costs
        # a kernel per step (the fused attention never materialises P), so it is on request -- task.build_generator sets it for
        # generate.py --print-alignment
        self.print_alignment = bool(print_alignment)
        self.retain_attention = bool(retain_attention) or self.print_alignment
        self.search = BeamSearch(tgt_dict) if search_strategy is None else search_strategy
        self.device_graph = True                                          # replay one recorded step (hipGraph); False: launch every step's kernels
        self.record_stats = False                                         # bench.py: two extra host syncs per call -> last_stats
        self.last_stats = {}

    # ------------------------------------------------------------------ API of the reference
    @torch.no_grad()
    def generate(self, models, sample, prefix_tokens=None, bos_token=None, **unused):
        was_training = [m.training for m in self.models]                  # the reference also ignores `models` here (:149-161)
        for m in self.models:
            m.eval()                                                      # sequence_generator.py:86-87
        try:
            return self._generate(self.models[0], sample, bos_token, prefix_tokens=prefix_tokens)
        finally:
            for m, t in zip(self.models, was_training):
                m.train(t)

    def _generate(self, model, sample, bos_token, prefix_tokens=None):
        net_input = sample["net_input"]
        src_tokens = net_input["src_tokens"]
        B, src_len = src_tokens.shape[0], src_tokens.shape[1]
        max_len = min(int(self.max_len_a * src_len + self.max_len_b), min(m.max_decoder_positions() for m in self.models) - 1)
        assert self.min_len <= max_len, "min_len cannot be larger than max_len, please adjust these!"
        if self.record_stats:
            import time
            torch.cuda.synchronize(); t0 = time.perf_counter()
        encs = [m.encoder.forward_non_torchscript(net_input) for m in self.models]       # one column per SENTENCE (see _expand)
        if self.record_stats:
            torch.cuda.synchronize(); t1 = time.perf_counter()
        hyps = self._beam_search([m.decoder for m in self.models], encs, B, src_tokens.device, max_len, self.search, bos_token,
                                 self.pad, self.unk, self.eos, self.vocab_size, prefix_tokens=prefix_tokens)
        if self.record_stats:
            torch.cuda.synchronize(); t2 = time.perf_counter()
            self.last_stats.update(encoder_s=t1 - t0, search_s=t2 - t1, src_frames=int(encs[0].encoder_out.shape[0]))
        for b, hs in enumerate(hyps):
            for h in hs:
                h.pop("origin")
                if self.print_alignment and h["attention"] is not None:
                    h["alignment"] = self._hard_alignment(h["attention"], net_input["src_lengths"][b], h["tokens"])
        return hyps

    def _hard_alignment(self, attn, src_len, tgt_tokens):
        """SequenceGeneratorWithAlignment.generate for a model without full-context alignment (sequence_generator.py:830-841) +
        fairseq/utils.py:486-503 extract_hard_alignment: for every target position that is neither pad nor EOS, the source position
        with the largest attention weight, as (source, target) pairs.  The source side of this model is filterbank frames, so the
        reference's token-to-word mapping has nothing to map: the pair holds the 0-based index on the axis the attention is over (the
        encoder's output frames, after subsampling and CTC compression) and the 0-based target position.  Index work only."""
        valid = (tgt_tokens.ne(self.pad) & tgt_tokens.ne(self.eos)).nonzero(as_tuple=False).view(-1)
        if attn.numel() == 0 or valid.numel() == 0:
            return []
        _, src_idx = attn.t()[valid.to(attn.device)].max(dim=1)
        return list(zip(src_idx.tolist(), valid.tolist()))

    def _expand(self, decoder, enc, B):
        """encoder output with one copy per hypothesis slot (sequence_generator.py:176-196)"""
        order0 = torch.arange(B, device=enc.encoder_out.device).repeat_interleave(self.beam_size)
        return decoder.owner.encoder.reorder_encoder_out(enc, order0)

    def _device_search(self, decoder, enc, B, max_len, search, bos_token, pad, unk, eos, V, prev_scores):
        """The whole loop inside libs2t_hip.so (decode.py / csrc/decode.hip) when nothing but the plain or the hierarchical beam search
        is asked for; None when this search needs the step-by-step path below."""
        from . import decode as DEC
        if not DEC.device_search_enabled() or type(search) not in (BeamSearch, HierarchicalBeamSearch):
            return None
        eng = decoder.engine
        if eng.hp.layernorm_embedding:                                     # the step's first launch takes the embedding sum as it is
            return None
        eo = enc.encoder_out.contiguous()
        if not eo.is_cuda:       # host tensors only reach this class under the CPU tests' stand-in engine (tests/cpu_stubs.py: host logic)
            return None
        klen = enc.src_lengths.to(torch.int32) if enc.encoder_padding_mask is not None else None
        ses = DEC.BeamDecodeSession(eng, decoder.pfx, eo, klen, self.beam_size, max_len, self.min_len, pad, unk, eos, V, self.unk_penalty,
                                    self.temperature, init_scores=prev_scores, step0_all_slots=prev_scores is not None)
        if not ses.ok:
            return None
        self.last_stats["steps"] = ses.run(eos if bos_token is None else bos_token, graph=self.device_graph)
        self.last_stats["launches_per_step"] = ses.launches_per_step
        hyps = ses.hypotheses(self.normalize_scores, self.len_penalty)
        out = []
        for hs in hyps:                                                    # sequence_generator.py:486-496: best first
            idx = sorted(range(len(hs)), key=lambda i: hs[i]["_score"])
            out.append([hs[i] for i in reversed(idx)])
            for h in hs:
                h.pop("_score")
        return out

    def _beam_search(self, decoder, enc, B, dev, max_len, search, bos_token, pad, unk, eos, V, prev_scores=None, prefix_tokens=None):
        """The search loop of sequence_generator.py:198-500 over `decoder` (incremental HIP decoder; a LIST of decoders with a list
        of encoder outputs = an ensemble).  `enc`: the encoder output(s), one column per sentence.  prev_scores [B, beam, 1]: starting
        scores of the slots for a HierarchicalBeamSearch.  prefix_tokens int64 [B, P]: forced first tokens (pad = free).  Every
        hypothesis also records `origin`, the slot of step 0 it descends from.  Returns per sentence the finalized hypotheses, best first."""
        beam = self.beam_size
        decoders = list(decoder) if isinstance(decoder, (list, tuple)) else [decoder]
        encs = list(enc) if isinstance(decoder, (list, tuple)) else [enc]
        if len(decoders) == 1 and prefix_tokens is None and self.no_repeat_ngram_size == 0 and not self.retain_attention \
                and decoders[0].owner.training is False:
            out = self._device_search(decoders[0], encs[0], B, max_len, search, bos_token, pad, unk, eos, V, prev_scores)
            if out is not None:
                return out
        encs = [self._expand(d, e, B) for d, e in zip(decoders, encs)]
        states = [d.begin_incremental(e, max_len + 1) for d, e in zip(decoders, encs)]
        attn = None                                                        # [N, Ts, max_len + 2], column step + 1 = the attention of step `step` (:286-292)
        if self.retain_attention:
            for d, st in zip(decoders, states):
                st["attn_layer"] = d.owner.hp.dec_layers - 1                # alignment_layer's default: the last layer (transformer.py:700-703)
        if prefix_tokens is not None:
            prefix_tokens = prefix_tokens.to(dev)

        N = B * beam
        scores = torch.zeros((N, max_len + 1), dtype=torch.float32, device=dev)
        tokens = torch.full((N, max_len + 2), pad, dtype=torch.int64, device=dev)
        tokens[:, 0] = eos if bos_token is None else bos_token
        origin = torch.arange(beam, device=dev).repeat(B)
        blacklist = torch.zeros((B, beam), dtype=torch.bool, device=dev)
        done = torch.zeros((B,), dtype=torch.bool, device=dev)           # sentence already has `beam` hypotheses
        finalized = [[] for _ in range(B)]
        finished = [False] * B
        cand_size = 2 * beam
        row0 = (torch.arange(B, device=dev) * beam).unsqueeze(1)
        cand_rank = torch.arange(cand_size, device=dev)

        reorder = None
        for step in range(max_len + 1):                                   # one extra step for the EOS marker
            self.last_stats["steps"] = step + 1
            member = []
            for d, st in zip(decoders, states):
                if reorder is not None:
                    d.reorder_incremental(st, reorder)
                member.append(K.log_softmax(d.step_incremental(st, tokens[:, step]), self.temperature))      # f32 [N, V]
            lprobs = member[0] if len(member) == 1 else K.ensemble_lse(member)       # log of the members' mean probability
            if self.retain_attention and all(st["attn"] is not None and st["attn"].shape[-1] == states[0]["attn"].shape[-1] for st in states):
                avg = states[0]["attn"][:, 0, :]
                for st in states[1:]:
                    avg = avg + st["attn"][:, 0, :]                        # index-free sum of a handful of [N, Ts] rows (:757-768)
                if attn is None:
                    attn = torch.zeros((avg.shape[0], avg.shape[1], max_len + 2), dtype=torch.float32, device=dev)
                attn[:, :, step + 1] = avg / len(states) if len(states) > 1 else avg
            lprobs[lprobs != lprobs] = -math.inf
            lprobs[:, pad] = -math.inf
            lprobs[:, unk] -= self.unk_penalty
            if step >= max_len:
                lprobs[:, :eos] = -math.inf
                lprobs[:, eos + 1:] = -math.inf
            if prefix_tokens is not None and step < prefix_tokens.shape[1] and step < max_len:
                lprobs, tokens, scores = self._prefix_tokens(step, lprobs, scores, tokens, prefix_tokens, beam, pad, eos)
            elif step < self.min_len:                                      # (does not apply inside a prefix: :270-280)
                lprobs[:, eos] = -math.inf
            if self.no_repeat_ngram_size > 0:
                self._no_repeat_ngram(tokens, lprobs, step)

            if prev_scores is not None:
                cand_scores, cand_tok, cand_beam = search.step(step, lprobs.view(B, beam, V), scores.view(B, beam, -1), prev_scores)
            else:
                cand_scores, cand_tok, cand_beam = search.step(step, lprobs.view(B, beam, V), scores.view(B, beam, -1))
            k = cand_scores.shape[1]
            cand_row = cand_beam + row0                                    # row of the parent hypothesis in tokens/scores

            eos_mask = cand_tok.eq(eos) & cand_scores.ne(-math.inf)
            eos_mask[:, :beam] &= ~blacklist
            top_eos = eos_mask[:, :beam] & ~done.unsqueeze(1)
            if bool(top_eos.any()):                                        # the step's only host sync
                self._finalize(step, top_eos, cand_row, cand_scores, tokens, scores, finalized, finished, max_len, eos, origin, attn)
                done = torch.tensor(finished, device=dev)
                if all(finished):
                    break
            assert step < max_len

            # next beam = the first `beam` candidates (rank order) that are not EOS; slots that could only be filled by an
            # EOS candidate are black-listed for the next step (sequence_generator.py:417-446)
            eos_mask[:, :beam] |= blacklist
            key = eos_mask.to(torch.int64) * cand_size + cand_rank[:k]
            key_top, pick = torch.topk(key, beam, dim=1, largest=False)
            blacklist = key_top.ge(cand_size)
            parent = torch.gather(cand_row, 1, pick).view(-1)
            tokens[:, :step + 1] = tokens.index_select(0, parent)[:, :step + 1]
            tokens[:, step + 1] = torch.gather(cand_tok, 1, pick).view(-1)
            if step > 0:
                scores[:, :step] = scores.index_select(0, parent)[:, :step]
            scores[:, step] = torch.gather(cand_scores, 1, pick).view(-1)
            origin = origin.index_select(0, parent)
            if attn is not None:
                attn[:, :, :step + 2] = attn.index_select(0, parent)[:, :, :step + 2]      # :427-430
            reorder = parent

        out = []
        for hyps in finalized:                                             # sequence_generator.py:486-496: best first
            idx = sorted(range(len(hyps)), key=lambda i: hyps[i]["score"].item())
            out.append([hyps[i] for i in reversed(idx)])
        return out

    @staticmethod
    def _prefix_tokens(step, lprobs, scores, tokens, prefix_tokens, beam, pad, eos):
        """sequence_generator.py:449-481: rows whose sentence has a forced token at `step` may only continue with it (its own
        log-probability kept); when the forced token is EOS every slot of the sentence becomes a copy of the first one.
        Index bookkeeping only (gather / scatter / fill)."""
        ptok = prefix_tokens[:, step].unsqueeze(-1).repeat(1, beam).view(-1)
        plp = lprobs.gather(-1, ptok.unsqueeze(-1))
        mask = ptok.ne(pad)
        forced = torch.full_like(lprobs, -math.inf).scatter_(-1, ptok.unsqueeze(-1), plp)
        lprobs = torch.where(mask.unsqueeze(-1), forced, lprobs)
        eos_mask = ptok.eq(eos)
        if bool(eos_mask.any()):
            rows = eos_mask.view(-1, beam)[:, 0]
            first = tokens[eos_mask].view(-1, beam, tokens.shape[-1])[:, 0, 1:step + 1]
            assert bool((first == prefix_tokens[rows][:, :step]).all())

            def rep(t):
                t = t.view(-1, beam, t.shape[-1])
                t[rows] = t[rows][:, :1, :]
                return t.view(-1, t.shape[-1])
            tokens, scores, lprobs = rep(tokens), rep(scores), rep(lprobs)
        return lprobs, tokens, scores

    def _no_repeat_ngram(self, tokens, lprobs, step):
        """sequence_generator.py:596-650: a hypothesis may not produce a token that completes an n-gram it already contains.
        Host-side integer work on the token rows, as in the reference (one device-to-host copy of `tokens` per step)."""
        n = self.no_repeat_ngram_size
        if step + 2 - n < 0:
            return
        rows = tokens[:, :step + 1].tolist()
        ban_r, ban_c = [], []
        for r, g in enumerate(rows):
            head = g[len(g) - (n - 1):] if n > 1 else []
            for j in range(len(g) - n + 1):
                if g[j:j + n - 1] == head:
                    ban_r.append(r); ban_c.append(g[j + n - 1])
        if ban_r:
            lprobs[torch.tensor(ban_r, device=lprobs.device), torch.tensor(ban_c, device=lprobs.device)] = -math.inf

    def _finalize(self, step, top_eos, cand_row, cand_scores, tokens, scores, finalized, finished, max_len, eos, origin, attn=None):
        """sequence_generator.py:502-600 finalize_hypos: hypotheses ending in EOS among the top `beam` candidates."""
        beam = self.beam_size
        sent, rank = top_eos.nonzero(as_tuple=True)                       # row-major: sentence ascending, rank ascending
        rows = cand_row[sent, rank]
        eos_s = cand_scores[sent, rank]
        toks = tokens.index_select(0, rows)[:, 1:step + 2].clone()
        toks[:, step] = eos
        org = origin.index_select(0, rows).tolist()
        pos = scores.index_select(0, rows)[:, :step + 1].clone()
        pos[:, step] = eos_s
        pos[:, 1:] = pos[:, 1:] - pos[:, :-1]
        attn_clone = attn.index_select(0, rows)[:, :, 1:step + 2] if attn is not None else None     # :510-514: src_len x tgt_len
        if self.normalize_scores:
            eos_s = eos_s / (step + 1) ** self.len_penalty
        for i, s in enumerate(sent.tolist()):
            if len(finalized[s]) < beam:
                finalized[s].append({"tokens": toks[i], "score": eos_s[i], "attention": None if attn_clone is None else attn_clone[i],
                                     "alignment": None,
                                     "positional_scores": pos[i], "origin": org[i]})
        for s in set(sent.tolist()):
            if not finished[s] and (len(finalized[s]) == beam or step == max_len):
                finished[s] = True


class TwoPhaseSequenceGenerator(SequenceGenerator):
    """Generation with the dual-decoder model (examples/speech_recognition/twophase_sequence_generator.py:52-170, built by the
    `speech_translation_dualdecoding` task): (1) beam search with `model.auxiliary_decoder` over the transcript dictionary ->
    `beam` transcript hypotheses per sentence (:477-762); (2) HierarchicalBeamSearch with `model.decoder`: slot i starts from the
    normalised score of transcript hypothesis i (:171-475; the dual-decoder's target decoder does not read the transcript,
    conv_transformer_dualdecoder.py:83-84).  The target length limit uses the longest transcript hypothesis of the batch as
    `src_len` (:178,213-218).  Every returned hypothesis carries `aux_tokens`, the transcript it descends from (:966-975)."""

    def __init__(self, models, src_dict, tgt_dict, **kw):
        super().__init__(models, tgt_dict, **kw)
        if not hasattr(self.models[0], "auxiliary_decoder"):
            raise TypeError("TwoPhaseSequenceGenerator needs a model with an auxiliary decoder (conv_transformer_dualdecoder)")
        self.src_pad, self.src_unk = src_dict.pad(), src_dict.unk()
        self.src_eos = src_dict.eos() if kw.get("eos") is None else kw["eos"]
        self.src_vocab_size = len(src_dict)
        self.src_search = BeamSearch(src_dict)
        self.search = HierarchicalBeamSearch(tgt_dict)
        if self.beam_size > self.src_vocab_size - 1:
            raise ValueError("beam larger than the transcript vocabulary")

    def _generate(self, model, sample, bos_token, prefix_tokens=None):
        if prefix_tokens is not None or len(self.models) != 1:
            raise NotImplementedError("the two-phase generator takes one dual-decoder model and no prefix tokens "
                                      "(twophase_sequence_generator.py:127-170)")
        net_input = sample["net_input"]
        src_tokens = net_input["src_tokens"]
        dev = src_tokens.device
        B, src_len = src_tokens.shape[0], src_tokens.shape[1]
        beam = self.beam_size
        max_pos = model.max_decoder_positions() - 1
        max_len = min(int(self.max_len_a * src_len + self.max_len_b), max_pos)
        assert self.min_len <= max_len, "min_len cannot be larger than max_len, please adjust these!"
        enc = model.encoder.forward_non_torchscript(net_input)
        aux = self._beam_search(model.auxiliary_decoder, enc, B, dev, max_len, self.src_search, bos_token,
                                self.src_pad, self.src_unk, self.src_eos, self.src_vocab_size)
        assert all(len(hs) == beam for hs in aux)
        max_aux_len = max(h["tokens"].shape[0] for hs in aux for h in hs)
        max_len2 = min(int(self.max_len_a * max_aux_len + self.max_len_b), max_pos)
        assert self.min_len <= max_len2, "min_len cannot be larger than max_len, please adjust these!"
        prev_scores = torch.stack([h["score"] for hs in aux for h in hs]).view(B, beam, 1)
        hyps = self._beam_search(model.decoder, enc, B, dev, max_len2, self.search, bos_token, self.pad, self.unk, self.eos,
                                 self.vocab_size, prev_scores=prev_scores)
        for b, hs in enumerate(hyps):
            for h in hs:
                h["aux_tokens"] = aux[b][h.pop("origin")]["tokens"]
        return hyps
