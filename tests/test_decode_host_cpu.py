"""Host logic of the device-resident beam search (fbk_fairseq_st_amd/decode.py): rebuilding the hypotheses from the per-step selection
records.  The records are produced here by a numpy re-enactment of the reference's bookkeeping (fairseq/sequence_generator.py:417-446:
the tokens / scores buffers re-ordered by the chosen parents every step), so the walk over parent links must give back exactly the
buffers' rows."""
import numpy as np

from fbk_fairseq_st_amd.decode import _pick_hidden_slice, walk_records


def test_walk_records_rebuilds_the_reordered_buffers():
    rs = np.random.RandomState(5)
    B, beam, steps, V, pad, eos = 3, 4, 9, 50, 1, 2
    N = B * beam
    tokens = np.full((N, steps + 2), pad, np.int64); tokens[:, 0] = eos
    scores = np.zeros((N, steps + 1), np.float32)
    origin = np.tile(np.arange(beam), B)
    tok_h = np.zeros((steps + 2, N), np.int32); par_h = np.zeros((steps + 2, N), np.int32); cum_h = np.zeros((steps + 2, N), np.float32)
    nfin = np.zeros(B, np.int32); fin_step = np.zeros((B, beam), np.int32); fin_row = np.zeros((B, beam), np.int32)
    fin_score = np.zeros((B, beam), np.float32)
    expected = [[] for _ in range(B)]
    for t in range(steps + 1):
        # some hypotheses end at this step (finalised from the arrangement BEFORE the re-ordering, as the reference does)
        for s in range(B):
            if rs.rand() < 0.35 and nfin[s] < beam:
                r = s * beam + rs.randint(beam)
                sc = np.float32(scores[r, t - 1] - rs.rand()) if t else np.float32(-rs.rand())
                k = nfin[s]
                fin_step[s, k], fin_row[s, k], fin_score[s, k] = t, r, sc
                nfin[s] += 1
                cum = np.concatenate([scores[r, :t], [sc]]).astype(np.float32)
                pos = cum.copy(); pos[1:] = cum[1:] - cum[:-1]
                expected[s].append((np.concatenate([tokens[r, 1:t + 1], [eos]]), pos, sc / np.float32((t + 1) ** 0.8), origin[r]))
        if t == steps:
            break
        parent = (np.arange(B)[:, None] * beam + rs.randint(beam, size=(B, beam))).reshape(-1)
        newtok = rs.randint(4, V, size=N)
        newsc = (scores[parent, t - 1] if t else 0) - rs.rand(N).astype(np.float32)
        tokens[:, :t + 1] = tokens[parent, :t + 1]; tokens[:, t + 1] = newtok
        scores[:, :t] = scores[parent, :t]; scores[:, t] = newsc
        origin = origin[parent]
        tok_h[t + 1], par_h[t + 1], cum_h[t + 1] = newtok, parent, newsc
    sent, tok, pos, score, org, length = walk_records(tok_h, par_h, cum_h, nfin, fin_step, fin_row, fin_score, beam, pad, eos, True, 0.8)
    assert sent.shape[0] == int(nfin.sum()) > 0
    seen = [0] * B
    for f in range(sent.shape[0]):
        s = int(sent[f]); et, ep, es, eo = expected[s][seen[s]]; seen[s] += 1
        n = int(length[f])
        assert tok[f, :n].tolist() == et.tolist()
        np.testing.assert_array_equal(pos[f, :n], ep)
        assert score[f] == es and int(org[f]) == int(eo)


def test_hidden_slice_choice():
    assert _pick_hidden_slice(2048, 16) == 128 and _pick_hidden_slice(2048, 1) == 64 and _pick_hidden_slice(768, 16) in (64, 128, 256)
    assert _pick_hidden_slice(100, 4) == 0


def test_implicit_attention_request_is_dropped_for_long_inputs():
    """eval-mode forwards return the encoder attention by default (transformer.py:700-703), but a source longer than the
    attention-probability kernel takes must not make validation / generation fail when nobody asked for the attention"""
    import types
    from fbk_fairseq_st_amd.conv_transformer import TransformerDecoder
    dec = TransformerDecoder.__new__(TransformerDecoder)
    dec.owner = types.SimpleNamespace(hp=types.SimpleNamespace(dec_layers=6))
    dec.training = False
    assert dec._alignment_request(None, None, None, src_frames=375) == (5, None)
    assert dec._alignment_request(None, None, None, src_frames=4000) == (None, None)          # implicit: dropped
    assert dec._alignment_request(None, None, True, src_frames=4000) == (5, None)             # asked for: kept (the kernel will refuse)
    assert dec._alignment_request(2, 4, None, src_frames=4000) == (2, 4)
    dec.training = True
    assert dec._alignment_request(None, None, None, src_frames=100) == (None, None)
