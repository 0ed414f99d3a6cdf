"""Markdown table of per-update kernel time by kernel group from a rocprofv3 kernel_stats.csv (the number of traced updates is
taken from the adam_kernel launch count): python tools/prof_table.py profiles/r02_bench_kernel_stats.csv"""
import csv, sys

GROUPS = [("`gemm256_kernel` NT (Y = XWᵀ + epilogue)", lambda n: "gemm256" in n and "Lb0E" in n),
          ("`gemm256_kernel` NN (dX = dY·W)", lambda n: "gemm256" in n),
          ("`wgrad_group_kernel` (all dW + db of a pass)", lambda n: "wgrad_group" in n),
          ("`attn_fwd2`", lambda n: "attn_fwd2" in n), ("`attn_bwd_dq2`", lambda n: "attn_bwd_dq2" in n),
          ("`attn_bwd_dkv2`", lambda n: "attn_bwd_dkv2" in n),
          ("decoder self-attention (first-generation kernels, T = 40)", lambda n: "attn_" in n),
          ("`gemm_fast_kernel` 128x128 / 64x64, `gemm_tn2_kernel` (decoder-side, logits, fc3 dW)", lambda n: "gemm_fast" in n or "gemm_tn2" in n),
          ("`conv2_fwd_kernel` / `conv2_dgrad_kernel` (direct 3x3 stride-2 convolution)", lambda n: "conv2_fwd" in n or "conv2_dgrad" in n),
          ("`gemm_kernel` with row gather (implicit-GEMM convolutions)", lambda n: "gemm_kernel" in n),
          ("`ln_bwd`", lambda n: "ln_bwd" in n), ("`ln_fwd`", lambda n: "ln_fwd" in n),
          ("`conv1_fwd` / `conv1_bwd` / `conv2_wgrad`", lambda n: "conv1_" in n or "conv2_wgrad" in n),
          ("`chan_sums` / `bn_apply` / `bn_bwd_apply` / `bn_finalize`", lambda n: "chan_sums" in n or "bn_" in n),
          ("`ctc_alphabeta` (side stream) / `ctc_grad` / `ctc_argmax_row` / rle / compress", lambda n: "ctc_" in n),
          ("`adam_kernel` + `sumsq` + clip", lambda n: "adam" in n or "sumsq" in n or "clip_coef" in n),
          ("`lsce`, embeddings, dropout, act_bwd, add_pos, colsum, permutes", lambda n: any(k in n for k in ("lsce", "embed", "dropout", "act_bwd", "add_pos", "colsum", "permute", "cast", "scale_by"))),
          ("copies / fills (runtime)", lambda n: "copyBuffer" in n or "fillBuffer" in n or "at::native" in n)]

rows = list(csv.DictReader(open(sys.argv[1])))
steps = [int(r["Calls"]) for r in rows if "adam_kernel" in r["Name"]][0]
acc = [[0, 0.0] for _ in GROUPS]
other = [0, 0.0]
for r in rows:
    for i, (_, f) in enumerate(GROUPS):
        if f(r["Name"]):
            acc[i][0] += int(r["Calls"]); acc[i][1] += float(r["TotalDurationNs"])
            break
    else:
        other[0] += int(r["Calls"]); other[1] += float(r["TotalDurationNs"])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("traced updates: %d" % steps)
print("| kernel | launches per update | ms per update | us each |\n|---|---|---|---|")
for (name, _), (c, ns) in zip(GROUPS, acc):
    if c:
        print("| %s | %.0f | %.2f | %.1f |" % (name, c / steps, ns / steps / 1e6, ns / c / 1e3))
print("| everything else | %.0f | %.2f | |" % (other[0] / steps, other[1] / steps / 1e6))
print("| **sum of kernel time** | | **%.2f** | |" % (tot / steps / 1e6))
