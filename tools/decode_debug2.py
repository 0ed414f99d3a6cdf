"""First launches of a device decode step against torch arithmetic on the same weights (diagnostic; s2t_set_option decode_stop_after)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_model_gpu import build_gen  # noqa: E402
from fbk_fairseq_st_amd import decode as DEC  # noqa: E402
from fbk_fairseq_st_amd import kernels as K  # noqa: E402
from fbk_fairseq_st_amd import lib as L  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "c"
task, model, src, lens, opts, exp, (cfg, W) = build_gen(tag, torch.float32)
beam, max_len = opts["beam_size"], int(opts["max_len_a"] * src.shape[1] + opts["max_len_b"])
dec = model.decoder
eng = dec.engine
P = lambda n: eng.P("decoder." + n).float()
with torch.no_grad():
    enc = model.encoder(src, lens)
    B, V = src.shape[0], len(task.target_dictionary)
    klen = enc.src_lengths.to(torch.int32) if enc.encoder_padding_mask is not None else None
    ses = DEC.BeamDecodeSession(eng, dec.pfx, enc.encoder_out.contiguous(), klen, beam, max_len, opts["min_len"], 1, 3, 2, V)
    lib = L.load()
    N, D, H = B * beam, eng.hp.D, eng.hp.heads
    d = lambda a, b: "%.3e (max %.3e)" % (float((a - b).abs().max()), float(b.abs().max()))

    def run(k):
        L.check(lib.s2t_decode_begin(ses.addr, 2, L.stream()), "begin")
        K.set_option("decode_stop_after", k)
        L.check(lib.s2t_decode_step(ses.addr, L.stream()), "step")
        K.set_option("decode_stop_after", 0)
        torch.cuda.synchronize()
    run(1)
    x0 = ses.bufs["x0"].clone()
    emb = (D ** 0.5) * P("embed_tokens.weight")[2] + eng.table(16, 1)[2]
    print("x0 vs embedding:", d(x0, emb.expand(N, D)))
    h = F.layer_norm(x0, (D,), P("layers.0.self_attn_layer_norm.weight"), P("layers.0.self_attn_layer_norm.bias"), 1e-5)
    qkv = h @ P("layers.0.self_attn.qkv.weight").t() + P("layers.0.self_attn.qkv.bias")
    c0 = ses.bufs["cache0"][0]
    print("k row:", d(c0[:, :D], qkv[:, D:2 * D]), " v row:", d(c0[:, D:], qkv[:, 2 * D:]))
    print("x1 (copy of x0 by the writer):", d(ses.bufs["x1"], x0))
    Wo = P("layers.0.self_attn.out_proj.weight")
    v = qkv[:, 2 * D:]
    for hh in range(H):
        share = v[:, hh * 64:(hh + 1) * 64] @ Wo[:, hh * 64:(hh + 1) * 64].t()
        print("  share head %d:" % hh, d(ses.bufs["part0"][hh], share))
    run(2)
    x1 = x0 + P("layers.0.self_attn.out_proj.bias") + v @ Wo.t()
    print("x after self block (X[0]):", d(ses.bufs["x0"], x1))
    run(1)
    with torch.no_grad():
        hh = 0
        got = ses.bufs["part0"][hh]                       # [N, D]
        Woh = Wo[:, hh * 64:(hh + 1) * 64]                # [D, 64]
        o_est = got @ torch.linalg.pinv(Woh.t())          # [N, 64]
        resid = (o_est @ Woh.t() - got).abs().max()
        print("o_est residual %.3e" % float(resid))
        vh, kh, qh = qkv[:, 2 * D + hh * 64: 2 * D + hh * 64 + 64], qkv[:, D + hh * 64: D + hh * 64 + 64], qkv[:, hh * 64: hh * 64 + 64]
        print("o_est vs v:", d(o_est, vh), " vs k:", d(o_est, kh), " vs q*scale:", d(o_est, qh * 0.125))
        print("row-wise |o_est - v| max:", [round(float(x), 3) for x in (o_est - vh).abs().max(dim=1)[0]])
        print("col-wise |o_est - v| max:", [round(float(x), 2) for x in (o_est - vh).abs().max(dim=0)[0]])
        exp0 = vh @ Woh.t()
        print("col-tile-wise share err:", [round(float((got[:, c:c + 16] - exp0[:, c:c + 16]).abs().max()), 3) for c in range(0, D, 16)])
        print("o_est[0,:12]", [round(float(x), 4) for x in o_est[0, :12]])
        print("v[0,:12]    ", [round(float(x), 4) for x in vh[0, :12]])
        # where does each o_est column come from?
        full_v = qkv[0, 2 * D:]
        full_k = qkv[0, D:2 * D]
        for c in range(8):
            dv = (full_v - o_est[0, c]).abs(); dk = (full_k - o_est[0, c]).abs()
            print("  col %d: nearest v index %d (err %.1e), nearest k index %d (err %.1e)" % (c, int(dv.argmin()), float(dv.min()), int(dk.argmin()), float(dk.min())))
