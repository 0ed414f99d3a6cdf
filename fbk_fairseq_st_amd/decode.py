"""Device-resident beam search over one incremental decoder.

The search loop of the reference (fairseq/sequence_generator.py:243-447 with fairseq/search.py:55-83 and the incremental
TransformerDecoder, fairseq/models/transformer.py:674-782) as `3 * layers + 4` kernel launches per decoding step, all of them
inside libs2t_hip.so (csrc/decode.hip, include/s2t_hip.h `s2t_decode_*`): the host launches one recorded step after the other
and looks at the `finished` flags every few steps.  Nothing of the state is re-ordered when the beam changes: the kernels follow
an ancestor table, the encoder-side K/V exist once per sentence, and the hypotheses are read back at the end by walking the
recorded (token, parent, cumulative score) triples.

`SequenceGenerator` (sequence_generator.py) takes this path for one model with the plain or the hierarchical beam search and
keeps its step-by-step path (the same decoder kernels + torch index bookkeeping) for ensembles, prefix tokens, n-gram blocking
and attention output.
"""
import ctypes
import os

import numpy as np
import torch

from . import lib as L

POLL_STEPS = 8                       # the host reads the `finished` flags every POLL_STEPS steps (one small D2H copy + sync)
LDS_CAP = 152 * 1024                 # csrc/decode.hip LDS_CAP


def _pick_hidden_slice(ffn, B):
    """hidden units per F-launch workgroup (64, 128 or 256): enough (sentence, slice) workgroups for the 256 CUs"""
    forced = int(os.environ.get("S2T_DECODE_HS", "0"))              # diagnostic: tools/decode_stamps.py compares the slice widths
    if forced and ffn % forced == 0:
        return forced
    for hs in ((128, 64, 256) if B * (ffn // 128) >= 128 else (64, 128, 256)):
        if ffn % hs == 0:
            return hs
    return 0


class BeamDecodeSession:
    """State + launch sequence of one beam search.  enc_out [Ts, B, D] (one column per SENTENCE), enc_klen int32 [B] or None."""

    def __init__(self, engine, pfx, enc_out, enc_klen, beam, max_len, min_len, pad, unk, eos, V, unk_penalty=0.0, temperature=1.0,
                 init_scores=None, step0_all_slots=False):
        hp = engine.hp
        Ts, B, D = enc_out.shape
        self.engine, self.B, self.beam, self.N, self.max_len = engine, B, beam, B * beam, max_len
        self.pad, self.eos = pad, eos
        dev, dtype = engine.dev, engine.dtype
        N, H, Ff, Ld = self.N, hp.heads, hp.ffn, hp.dec_layers
        Tsp = (Ts + 127) // 128 * 128
        hs = _pick_hidden_slice(Ff, B)
        self.ok = hs > 0
        if not self.ok:
            return
        FS = Ff // hs
        lib = L.load()
        st = L.stream()
        keep = self._keep = []

        self.bufs = {}

        def dev_t(shape, dt, name=None):
            t = torch.empty(shape, dtype=dt, device=dev)
            keep.append(t)
            if name:
                self.bufs[name] = t                          # by name for tools/decode_debug.py and the tests
            return t
        d = self.desc = L.DecodeDesc()
        d.dtype, d.B, d.beam, d.D, d.heads, d.ffn, d.layers, d.V, d.ldv = L.F32 if dtype == torch.float32 else L.BF16, B, beam, D, H, Ff, Ld, V, V
        d.Ts, d.Tsp, d.max_len, d.min_len, d.ffn_slices, d.gelu = Ts, Tsp, max_len, min_len, FS, int(hp.act == "gelu")
        d.pad, d.unk, d.eos, d.step0_all_slots = pad, unk, eos, int(bool(step0_all_slots))
        d.ln_eps, d.embed_scale = hp.ln_eps, 1.0 if hp.no_scale_embedding else float(D) ** 0.5
        d.unk_penalty, d.inv_temperature = float(unk_penalty), 1.0 / float(temperature)
        self.layers = (L.DecodeLayer * Ld)()
        d.layer = ctypes.addressof(self.layers)
        self.addr = ctypes.addressof(d)
        self.ok = 0 < lib.s2t_decode_lds_bytes(self.addr) <= LDS_CAP         # shape limits of csrc/decode.hip: checked before anything is allocated
        if not self.ok:
            return
        # state: one int32 and one float32 block, so that the read-back at the end is two copies
        M2 = max_len + 2
        isz = dict(tok_hist=M2 * N, par_hist=M2 * N, fin_step=B * beam, fin_row=B * beam, nfin=B, finished=B, steps=B, blacklist=N,
                   anc=N * (max_len + 1), cand_idx=N * 2 * beam)
        fsz = dict(cum_hist=M2 * N, fin_score=B * beam, cand_val=N * 2 * beam)
        self.ibuf = dev_t((sum(isz.values()),), torch.int32)
        self.fbuf = dev_t((sum(fsz.values()),), torch.float32)
        self.ioff, self.foff = {}, {}
        o = 0
        for k, n in isz.items():
            self.ioff[k] = (o, n); setattr(d, k, self.ibuf.data_ptr() + 4 * o); o += n
        o = 0
        for k, n in fsz.items():
            self.foff[k] = (o, n); setattr(d, k, self.fbuf.data_ptr() + 4 * o); o += n
        self.read_i = self.ioff["steps"][0]               # [tok_hist | par_hist | fin_step | fin_row | nfin | finished] come first
        self.read_f = self.foff["cand_val"][0]
        d.x0, d.x1 = dev_t((N, D), torch.float32, "x0").data_ptr(), dev_t((N, D), torch.float32, "x1").data_ptr()
        np_max = max(H, FS)
        d.part0, d.part1 = dev_t((np_max, N, D), dtype, "part0").data_ptr(), dev_t((np_max, N, D), dtype, "part1").data_ptr()
        d.xn = dev_t((N, D), dtype, "xn").data_ptr()
        d.logits = dev_t((N, V), torch.float32, "logits").data_ptr()
        if enc_klen is not None:
            keep.append(enc_klen)
            d.enc_klen = enc_klen.data_ptr()
        if init_scores is not None:
            init_scores = init_scores.to(device=dev, dtype=torch.float32).contiguous().view(-1)
            assert init_scores.numel() == N
            keep.append(init_scores)
            d.init_scores = init_scores.data_ptr()
        W, P = engine.W, engine.P
        ptr = lambda t: (keep.append(t), t.data_ptr())[1]

        def packed(name):
            """the weight in fragment-major order (s2t_decode_pack_weight): re-made per search -- the weights may have moved since the last
            one, and 52 MB of copies are tens of microseconds against a search of tens of milliseconds"""
            w = W(name)
            n, k = w.shape
            wp = dev_t(((n + 15) // 16 * 16, k), dtype)
            L.check(lib.s2t_decode_pack_weight(d.dtype, w.data_ptr(), w.stride(0), n, k, wp.data_ptr(), st), "s2t_decode_pack_weight")
            return wp.data_ptr()
        d.lnf_g, d.lnf_b = ptr(P(pfx + "layer_norm.weight")), ptr(P(pfx + "layer_norm.bias"))
        d.w_out = packed(engine.out_proj(pfx) + ".weight")
        d.embed = ptr(W(pfx + "embed_tokens.weight"))
        d.pos_table = ptr(engine.table(pad + 3 + max_len, pad))
        enc2d = enc_out.reshape(Ts * B, D)
        for l in range(Ld):
            lp = pfx + "layers.%d." % l
            y = self.layers[l]
            for ln, stem in (("ln1", "self_attn_layer_norm"), ("lnx", "encoder_attn_layer_norm"), ("ln2", "final_layer_norm")):
                setattr(y, ln + "_g", ptr(P(lp + stem + ".weight"))); setattr(y, ln + "_b", ptr(P(lp + stem + ".bias")))
            for f, stem in (("qkv", "self_attn.qkv"), ("o", "self_attn.out_proj"), ("xq", "encoder_attn.q_proj"),
                            ("xo", "encoder_attn.out_proj"), ("fc1", "fc1"), ("fc2", "fc2")):
                setattr(y, "w_" + f, packed(lp + stem + ".weight")); setattr(y, "b_" + f, ptr(P(lp + stem + ".bias")))
            kv = engine.linear(enc2d, lp + "encoder_attn.kv")               # [Ts * B, 2D]: the reference's static_kv, once per sentence
            kp, vp = dev_t((B, H, Tsp, 64), dtype), dev_t((B, H, 64, Tsp), dtype)      # fragment-major inside (include/s2t_hip.h)
            L.check(lib.s2t_decode_prepare_enc(d.dtype, kv.data_ptr(), kp.data_ptr(), vp.data_ptr(), Ts, Tsp, B, D, H, st), "s2t_decode_prepare_enc")
            keep.append(kv)
            y.kv_enc, y.vt_enc = kp.data_ptr(), vp.data_ptr()
            y.kv_cache = dev_t((max_len + 1, N, 2 * D), dtype, "cache%d" % l).data_ptr()
        self.launches_per_step = 3 * Ld + 4
        self.steps_run = 0

    def view_i(self, name):
        o, n = self.ioff[name]
        return self.ibuf[o:o + n]

    def view_f(self, name):
        o, n = self.foff[name]
        return self.fbuf[o:o + n]

    def run(self, bos, graph=True):
        """the whole search; returns the number of steps launched"""
        lib = L.load()
        st = L.stream()
        L.check(lib.s2t_decode_begin(self.addr, int(bos), st), "s2t_decode_begin")
        exec_ = ctypes.c_void_p(0)
        per = POLL_STEPS if graph else 1                    # steps per launch: one recorded graph holds POLL_STEPS of them
        if graph:
            L.check(lib.s2t_decode_graph_create(self.addr, per, ctypes.addressof(exec_)), "s2t_decode_graph_create")
        fo, fn = self.ioff["finished"]
        finished = self.ibuf[fo:fo + fn]
        steps = 0
        try:
            while steps < self.max_len + 1:
                if graph:
                    L.check(lib.s2t_decode_graph_launch(exec_.value, st), "s2t_decode_graph_launch")
                else:
                    L.check(lib.s2t_decode_step(self.addr, st), "s2t_decode_step")
                steps += per                                # (the last graph may run past max_len: its kernels return at once there)
                if steps % POLL_STEPS == 0 and steps < self.max_len + 1 and bool(finished.all()):
                    break
        finally:
            if graph and exec_.value:
                torch.cuda.current_stream().synchronize()
                lib.s2t_decode_graph_destroy(exec_.value)
        steps = min(steps, self.max_len + 1)
        self.steps_run = steps
        return steps

    def hypotheses(self, normalize_scores, len_penalty):
        """Read the records back and rebuild what finalize_hypos (sequence_generator.py:502-600) collects: per sentence the
        finalised hypotheses in the order the reference appends them, each (tokens, score, positional_scores, origin)."""
        B, beam, N = self.B, self.beam, self.N
        ib = self.ibuf[:self.read_i].cpu().numpy()
        fb = self.fbuf[:self.read_f].cpu().numpy()
        iv = lambda k: ib[self.ioff[k][0]:self.ioff[k][0] + self.ioff[k][1]]
        fv = lambda k: fb[self.foff[k][0]:self.foff[k][0] + self.foff[k][1]]
        M2 = self.max_len + 2
        sent, tok, pos, score, origin, length = walk_records(
            iv("tok_hist").reshape(M2, N), iv("par_hist").reshape(M2, N), fv("cum_hist").reshape(M2, N), iv("nfin"),
            iv("fin_step").reshape(B, beam), iv("fin_row").reshape(B, beam), fv("fin_score").reshape(B, beam), beam, self.pad, self.eos,
            normalize_scores, len_penalty)
        dev = self.engine.dev
        tok_d, pos_d, score_d = torch.from_numpy(tok).to(dev), torch.from_numpy(pos).to(dev), torch.from_numpy(score).to(dev)
        out = [[] for _ in range(B)]
        for f in range(sent.shape[0]):
            n = int(length[f])
            out[int(sent[f])].append({"tokens": tok_d[f, :n], "score": score_d[f], "attention": None, "alignment": None,
                                      "positional_scores": pos_d[f, :n], "origin": int(origin[f]), "_score": float(score[f])})
        return out


def walk_records(tok_h, par_h, cum_h, nfin, fin_step, fin_row, fin_score, beam, pad, eos, normalize_scores, len_penalty):
    """Host side of the device search (numpy): from the per-step selection records -- arrangement i = the beam after i selections:
    tok_h[i][n] the token slot n received, par_h[i][n] the slot of arrangement i-1 it continues, cum_h[i][n] its cumulative score --
    and the finalisation records (step, parent slot, score of the EOS candidate; nfin[s] of them per sentence, in the order the
    reference appends them) to what finalize_hypos builds (sequence_generator.py:502-560): tokens (ending in EOS), positional scores
    (differences of the cumulative ones), the length-normalised score (:553-554) and the step-0 slot each hypothesis descends from.
    Returns (sentence, tokens [F, Lmax] pad-filled, positional scores, score, origin, length), one row per hypothesis."""
    B = nfin.shape[0]
    sel = np.arange(beam)[None, :] < nfin[:, None]
    sent = np.nonzero(sel)[0]
    fstep = fin_step[sel].astype(np.int64)
    rows = fin_row[sel].astype(np.int64)
    fsc = fin_score[sel].astype(np.float32)
    F = int(sent.shape[0])
    Tm = int(fstep.max()) if F else 0
    tok = np.full((F, Tm + 1), pad, dtype=np.int64)
    cum = np.zeros((F, Tm + 1), dtype=np.float32)
    idx = np.arange(F)
    tok[idx, fstep] = eos
    cum[idx, fstep] = fsc
    for i in range(Tm, 0, -1):                                         # arrangement i -> i-1 along the parent links
        act = fstep >= i
        r = rows[act]
        tok[act, i - 1] = tok_h[i][r]
        cum[act, i - 1] = cum_h[i][r]
        rows[act] = par_h[i][r]
    origin = rows % beam
    pos = cum.copy()
    pos[:, 1:] = cum[:, 1:] - cum[:, :-1]
    score = fsc / ((fstep + 1).astype(np.float64) ** len_penalty).astype(np.float32) if normalize_scores else fsc
    return sent, tok, pos, score.astype(np.float32), origin, fstep + 1


def device_search_enabled():
    return os.environ.get("S2T_DEVICE_SEARCH", "1") != "0"
