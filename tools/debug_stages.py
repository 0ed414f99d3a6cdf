"""Per-stage comparison of the HIP engine with the golden trace of the reference (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_model_gpu import build, to_dev
from fbk_fairseq_st_amd import kernels as K

name = sys.argv[1] if len(sys.argv) > 1 else "model_a"
g, cfg, W, sample, meta, model, crit = build(name)
model.train()
eng = model.engine
s = to_dev(sample); ni = s["net_input"]
def err(a, b, what):
    a = a.detach().float().cpu().numpy(); 
    if a.shape != b.shape: print(what, "SHAPE", a.shape, b.shape); return
    print("%-14s err %.3e scale %.3e" % (what, np.abs(a - b).max(), np.abs(b).max()))
lens = ni["src_lengths"]
x, len4, len4_32, c = eng.subsample_fwd(ni["src_tokens"], lens.to(torch.int64), True, 0)
B = x.shape[1]
err(c["y1n"].permute(0, 3, 1, 2), g["train_conv0"], "conv0+bn")
T4, F4, C = c["T4"], c["F4"], 64
z2n = c["z2n"].view(T4, B, F4, C).permute(1, 3, 0, 2)
err(z2n, g["train_conv1"], "conv1+bn")
z2 = c["z2"].view(T4, B, F4, C).permute(1, 3, 0, 2)
import torch.nn.functional as F
from oracle import s2t_ref
# reference conv1 pre-BN from the oracle
tr = {}
xs, l2, st = s2t_ref.subsample(W, cfg, sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"], True, tr)
y0 = torch.from_numpy(g["train_conv0"])
pre = F.relu(F.conv2d(y0, W["encoder.convolutions.1.weight"], W["encoder.convolutions.1.bias"], stride=2, padding=1))
err(z2, pre.numpy(), "conv1 pre-bn")
err(x, g["train_enc_in0"], "enc_in0")
print("len4", len4.tolist())
out, ctx = eng.encoder_forward(ni["src_tokens"], lens, True, 0, return_all_hiddens=True)
for l in range(cfg["enc_layers"]):
    k = "train_enc_layer%d" % l
    if k in g: err(out["states"][l], g[k], "layer%d" % l)
# ---- layer 0 internals vs the oracle
ca, cf = ctx["layers"][0]
xin = torch.from_numpy(g["train_enc_in0"])
p = "encoder.layers.0."
h = s2t_ref.layer_norm(W, p + "self_attn_layer_norm.", xin)
err(ca["h"].view(h.shape), h.numpy(), "L0 ln1")
q = F.linear(h, W[p + "self_attn.q_proj.weight"], W[p + "self_attn.q_proj.bias"])
k = F.linear(h, W[p + "self_attn.k_proj.weight"], W[p + "self_attn.k_proj.bias"])
v = F.linear(h, W[p + "self_attn.v_proj.weight"], W[p + "self_attn.v_proj.bias"])
D = cfg["D"]
err(ca["qkv"][:, :, :D], q.numpy(), "L0 q")
err(ca["qkv"][:, :, D:2 * D], k.numpy(), "L0 k")
err(ca["qkv"][:, :, 2 * D:], v.numpy(), "L0 v")
mask = s2t_ref.length_mask(torch.tensor([16, 13, 10]), 16)
att = s2t_ref.mha(W, p + "self_attn.", cfg["heads"], h, h, mask)
x1 = xin + att
err(cf["x"].view(x1.shape), x1.numpy(), "L0 x1 (after attn)")
# context before out-proj
Tq, B_, _ = h.shape; hd = D // cfg["heads"]
qh = (q * hd ** -0.5).view(Tq, B_ * cfg["heads"], hd).transpose(0, 1); kh = k.view(Tq, B_ * cfg["heads"], hd).transpose(0, 1); vh = v.view(Tq, B_ * cfg["heads"], hd).transpose(0, 1)
sc = torch.bmm(qh, kh.transpose(1, 2)).view(B_, cfg["heads"], Tq, Tq).masked_fill(mask[:, None, None, :], float("-inf")).view(B_ * cfg["heads"], Tq, Tq)
cx = torch.bmm(torch.softmax(sc, -1), vh).transpose(0, 1).reshape(Tq, B_, D)
err(ca["ctx"], cx.numpy(), "L0 ctx")
x1t = torch.from_numpy(x1.numpy())
h2 = s2t_ref.layer_norm(W, p + "final_layer_norm.", x1t)
err(cf["h"].view(h2.shape), h2.numpy(), "L0 ln2")
a = F.relu(F.linear(h2, W[p + "fc1.weight"], W[p + "fc1.bias"]))
err(cf["a"].view(a.shape), a.numpy(), "L0 fc1+relu")
y = x1t + F.linear(a, W[p + "fc2.weight"], W[p + "fc2.bias"])
err(out["states"][0], y.numpy(), "L0 out")
err(eng.P(p + "fc1.weight"), W[p + "fc1.weight"].numpy(), "fc1.w")
err(eng.P(p + "fc2.weight"), W[p + "fc2.weight"].numpy(), "fc2.w")
err(eng.P(p + "fc2.bias"), W[p + "fc2.bias"].numpy(), "fc2.b")
a_d = cf["a"]; 
y2 = K.gemm(a_d, eng.W(p + "fc2.weight"), bias=eng.P(p + "fc2.bias"), residual=cf["x"])
err(y2.view(y.shape), y.numpy(), "fc2 direct")
y3 = K.gemm(a_d, eng.W(p + "fc2.weight"), bias=eng.P(p + "fc2.bias"))
err(y3.view(y.shape), (y - x1t).numpy(), "fc2 nores")
