"""Step-by-step comparison of the device-resident search (csrc/decode.hip) with the per-kernel incremental decoder on a golden case:
logits of every step given the SAME tokens / beam re-ordering (the device's own choices), then the choices against torch.topk."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_model_gpu import build_gen  # noqa: E402
from fbk_fairseq_st_amd import decode as DEC  # noqa: E402
from fbk_fairseq_st_amd import kernels as K  # noqa: E402
from fbk_fairseq_st_amd import lib as L  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "c"
dtype = torch.bfloat16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else torch.float32
task, model, src, lens, opts, exp, (cfg, W) = build_gen(tag, dtype)
beam, max_len = opts["beam_size"], int(opts["max_len_a"] * src.shape[1] + opts["max_len_b"])
dec = model.decoder
with torch.no_grad():
    enc = model.encoder(src, lens)
    B = src.shape[0]
    V = len(task.target_dictionary)
    klen = enc.src_lengths.to(torch.int32) if enc.encoder_padding_mask is not None else None
    ses = DEC.BeamDecodeSession(dec.engine, dec.pfx, enc.encoder_out.contiguous(), klen, beam, max_len, opts["min_len"], 1, 3, 2, V,
                                opts["unk_penalty"], opts["temperature"])
    print("session ok", ses.ok, "Ts", enc.encoder_out.shape[0], "B", B, "beam", beam, "max_len", max_len)
    lib = L.load()
    L.check(lib.s2t_decode_begin(ses.addr, 2, L.stream()), "begin")
    order0 = torch.arange(B, device=src.device).repeat_interleave(beam)
    enc_x = model.encoder.reorder_encoder_out(enc, order0)
    st = dec.begin_incremental(enc_x, max_len + 1)
    N = B * beam
    toks = torch.full((N,), 2, dtype=torch.int64, device=src.device)
    M2 = max_len + 2
    for step in range(min(max_len + 1, 6)):
        x0 = ses.bufs["x0"].clone()
        L.check(lib.s2t_decode_step(ses.addr, L.stream()), "step")
        torch.cuda.synchronize()
        ref = dec.step_incremental(st, toks).float()[:, :V]
        got = ses.bufs["logits"]
        print("step %d: logits max|diff| %.3e (ref max %.3e)   x0 finite %s" % (step, float((ref - got).abs().max()), float(ref.abs().max()),
                                                                               bool(torch.isfinite(x0).all())))
        tok_h = ses.view_i("tok_hist").view(M2, N)[step + 1].long()
        par_h = ses.view_i("par_hist").view(M2, N)[step + 1].long()
        cum_h = ses.view_f("cum_hist").view(M2, N)[step + 1]
        lp = K.log_softmax(ref.contiguous(), opts["temperature"])
        print("   device tokens", tok_h.tolist()[:10], "parents", par_h.tolist()[:10], "cum", [round(v, 3) for v in cum_h.tolist()[:5]])
        print("   steps", ses.view_i("steps").tolist(), "nfin", ses.view_i("nfin").tolist(), "finished", ses.view_i("finished").tolist())
        dec.reorder_incremental(st, par_h)
        toks = tok_h
