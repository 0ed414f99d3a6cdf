// One Transformer layer per C call (include/s2t_hip.h, "one Transformer layer per call"): the launch schedule of
// engine.py's self_attn_block / cross_attn_block / ffn_block functions (fairseq/modules/transformer_layer.py:87-139, 243-377,
// pre-LN) restated in C++, launch for launch and seed for seed, over the entry points of this library.  No kernel lives here.
// Why: a launch issued from Python costs 6-12 us of host time (allocation, argument marshalling, the call), a launch issued from
// here ~2.5 us (hipLaunchKernel); with 8 utterances per GPU the kernels of an update take ~5 ms and the ~360 calls of the
// per-kernel path ~7.  tests/test_engine_gpu.py holds both paths to the same bits.
#include "common.hpp"
#include "../../include/s2t_hip.h"

namespace {
constexpr size_t AL = 256;
inline size_t up(size_t n) { return (n + AL - 1) / AL * AL; }

// saved activations of one layer, in bytes from `ws`
struct Ws {
    size_t h1, st1, qkv, ctx, lse, y1;                    // self-attention block (y1 = its output = input of the next block)
    size_t h2, st2, q, kv, ctx2, lse2, y2;                // encoder-attention block (decoder)
    size_t h3, st3, a, pre, amask;                        // FFN
    size_t total, nmask;
};
Ws layout(const S2TLayerDesc& L, int training) {
    Ws w{};
    const size_t M = (size_t)L.T * L.B, D = L.D, e = 2, Ms = (size_t)L.Ts * L.B;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += up(bytes); return at; };
    w.h1 = take(M * D * e); w.st1 = take(2 * M * 4); w.qkv = take(M * 3 * D * e); w.ctx = take(M * D * e);
    w.lse = take((size_t)L.B * L.heads * L.T * 4); w.y1 = take(M * D * e);
    if (L.decoder) {
        w.h2 = take(M * D * e); w.st2 = take(2 * M * 4); w.q = take(M * D * e); w.kv = take(Ms * 2 * D * e); w.ctx2 = take(M * D * e);
        w.lse2 = take((size_t)L.B * L.heads * L.T * 4); w.y2 = take(M * D * e);
    }
    w.h3 = take(M * D * e); w.st3 = take(2 * M * 4); w.a = take(M * (size_t)L.ffn * e);
    if (L.gelu) w.pre = take(M * (size_t)L.ffn * e);
    w.nmask = (training && !L.gelu) ? s2t_gemm_relu_mask_bytes((int)M, L.ffn, L.D) : 0;
    if (w.nmask) w.amask = take(w.nmask);
    w.total = o;
    return w;
}
// scratch of the backward: everything a weight-gradient product reads as dY stays here until the grouped launch
struct Tmp { size_t d3, da, dh3, dx2, dx2d, d2, dctx2, dq, dkv, dh2, dx1, dx1d, d1, dctx, dqkv, dh1, delta, total; };
Tmp tmp_layout(const S2TLayerDesc& L) {
    Tmp t{};
    const size_t M = (size_t)L.T * L.B, D = L.D, e = 2, Ms = (size_t)L.Ts * L.B;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += up(bytes); return at; };
    t.d3 = take(M * D * e); t.da = take(M * (size_t)L.ffn * e); t.dh3 = take(M * D * e); t.dx2 = take(M * D * e); t.dx2d = take(M * D * e);
    if (L.decoder) {
        t.d2 = take(M * D * e); t.dctx2 = take(M * D * e); t.dq = take(M * D * e); t.dkv = take(Ms * 2 * D * e); t.dh2 = take(M * D * e);
        t.dx1 = take(M * D * e); t.dx1d = take(M * D * e);
    }
    t.d1 = take(M * D * e); t.dctx = take(M * D * e); t.dqkv = take(M * 3 * D * e); t.dh1 = take(M * D * e);
    t.delta = take((size_t)L.B * L.heads * L.T * 4);
    t.total = o;
    return t;
}

#define TRY(call) do { const int rc_ = (call); if (rc_ != S2T_OK) return rc_; } while (0)

// y = act(x W^T + b) with the epilogue options of the per-kernel path (kernels.gemm)
int linear(const S2TLayerDesc& L, int M, int N, int K, const void* x, const void* w, const float* b, void* y, const void* residual,
           const void* aux, void* aux_out, int ldaux, int act, float p_drop, unsigned long long seed, void* st) {
    return s2t_gemm_gather(L.dtype, L.dtype, 0, 0, M, N, K, x, K, w, K, y, N, b, residual, residual ? N : 0, aux, aux_out, ldaux, act, 0, 1, 1.f,
                           nullptr, 0, nullptr, nullptr, p_drop, seed, st);
}
// dx = epi(dy W) with W [n_out][n_in] as it lies in memory (engine.linear_bwd's data gradient); the weight-gradient product is queued
int linear_dx(const S2TLayerDesc& L, int M, int n_out, int n_in, const void* dy, const void* w, void* dx, const void* aux, int ldaux, int act,
              float alpha, int accumulate, void* st) {
    return s2t_gemm_gather(L.dtype, L.dtype, 0, 1, M, n_in, n_out, dy, n_out, w, n_in, dx, n_in, nullptr, nullptr, 0, aux, nullptr, ldaux, act,
                           accumulate, 1, alpha, nullptr, 0, nullptr, nullptr, 0.f, 0ull, st);
}
int queue_dw(S2TLayerCall& c, const void* dy, const void* x, float* dw, float* db, int n_out, int n_in, int tokens) {
    if (c.n_items >= c.max_items || !c.items) return S2T_EINVAL;
    S2TWgradProblem& p = c.items[c.n_items++];
    p.dY = dy; p.X = x; p.dW = dw; p.db = db; p.n_out = n_out; p.n_in = n_in; p.tokens = tokens; p.ldy = n_out; p.ldx = n_in; p.ldw = n_in;
    return S2T_OK;
}
bool bad(const S2TLayerDesc* L) {
    return !L || L->dtype != S2T_BF16 || L->T <= 0 || L->B <= 0 || L->D <= 0 || L->heads <= 0 || L->D % L->heads || L->ffn <= 0 ||
           (L->decoder && L->Ts <= 0) || (L->D & 7) || (L->ffn & 7);
}
}  // namespace

extern "C" size_t s2t_layer_ws_bytes(const S2TLayerDesc* L, int training) { return bad(L) ? 0 : layout(*L, training).total; }
extern "C" size_t s2t_layer_bwd_tmp_bytes(const S2TLayerDesc* L) { return bad(L) ? 0 : tmp_layout(*L).total; }

extern "C" int s2t_layer_fwd(const S2TLayerDesc* Lp, S2TLayerCall* c, void* st) {
    if (bad(Lp) || !c || !c->x || !c->y || !c->ws || (Lp->decoder && !c->enc)) return S2T_EINVAL;
    const S2TLayerDesc& L = *Lp;
    const Ws w = layout(L, c->training);
    char* ws = static_cast<char*>(c->ws);
    const int M = L.T * L.B, D = L.D, Ff = L.ffn, H = L.heads, d = D / H;
    const float scale = (float)pow((double)d, -0.5);         // kernels.attn_fwd: d ** -0.5
    const float pa = c->training ? L.p_attn : 0.f, p = c->training ? L.p_drop : 0.f, pact = c->training ? L.p_act : 0.f;
    // ---- self-attention block: x + dropout(out_proj(attn(LN(x))))        (transformer_layer.py:103-124 / 296-322)
    float* st1 = reinterpret_cast<float*>(ws + w.st1);
    TRY(s2t_layernorm_fwd(L.dtype, c->x, L.ln1_g, L.ln1_b, ws + w.h1, st1, st1 + M, M, D, L.ln_eps, st));
    TRY(linear(L, M, 3 * D, D, ws + w.h1, L.w_qkv, L.b_qkv, ws + w.qkv, nullptr, nullptr, nullptr, 0, S2T_ACT_NONE, 0.f, 0, st));
    const char* qkv = ws + w.qkv;
    const long ts3 = (long)L.B * 3 * D, bs3 = 3 * D, ts1 = (long)L.B * D, bs1 = D;
    TRY(s2t_attn_fwd(L.dtype, d, L.B, H, L.T, L.T, qkv, ts3, bs3, qkv + (size_t)D * 2, ts3, bs3, qkv + (size_t)2 * D * 2, ts3, bs3, ws + w.ctx, ts1, bs1,
                     reinterpret_cast<float*>(ws + w.lse), c->self_klen, L.causal, L.dist_penalty, scale, pa, c->seed_sa_attn, st));
    TRY(linear(L, M, D, D, ws + w.ctx, L.w_o, L.b_o, ws + w.y1, c->x, nullptr, nullptr, 0, S2T_ACT_NONE, p, c->seed_sa_out, st));
    const char* xin = ws + w.y1;
    // ---- encoder-attention block (decoder layer)                           (transformer_layer.py:324-352)
    if (L.decoder) {
        const int Ms = L.Ts * L.B;
        float* st2 = reinterpret_cast<float*>(ws + w.st2);
        TRY(s2t_layernorm_fwd(L.dtype, xin, L.lnx_g, L.lnx_b, ws + w.h2, st2, st2 + M, M, D, L.ln_eps, st));
        TRY(linear(L, M, D, D, ws + w.h2, L.w_xq, L.b_xq, ws + w.q, nullptr, nullptr, nullptr, 0, S2T_ACT_NONE, 0.f, 0, st));
        TRY(linear(L, Ms, 2 * D, D, c->enc, L.w_xkv, L.b_xkv, ws + w.kv, nullptr, nullptr, nullptr, 0, S2T_ACT_NONE, 0.f, 0, st));
        const char* kv = ws + w.kv;
        const long ts2 = (long)L.B * 2 * D, bs2 = 2 * D;
        TRY(s2t_attn_fwd(L.dtype, d, L.B, H, L.T, L.Ts, ws + w.q, ts1, bs1, kv, ts2, bs2, kv + (size_t)D * 2, ts2, bs2, ws + w.ctx2, ts1, bs1,
                         reinterpret_cast<float*>(ws + w.lse2), c->enc_klen, 0, 0, scale, pa, c->seed_xa_attn, st));
        TRY(linear(L, M, D, D, ws + w.ctx2, L.w_xo, L.b_xo, ws + w.y2, xin, nullptr, nullptr, 0, S2T_ACT_NONE, p, c->seed_xa_out, st));
        xin = ws + w.y2;
    }
    // ---- FFN block: x + dropout(fc2(dropout(act(fc1(LN(x))))))            (transformer_layer.py:128-136)
    float* st3 = reinterpret_cast<float*>(ws + w.st3);
    TRY(s2t_layernorm_fwd(L.dtype, xin, L.ln2_g, L.ln2_b, ws + w.h3, st3, st3 + M, M, D, L.ln_eps, st));
    if (L.gelu) {
        TRY(linear(L, M, Ff, D, ws + w.h3, L.w_fc1, L.b_fc1, ws + w.a, nullptr, nullptr, ws + w.pre, Ff, S2T_ACT_GELU, pact, c->seed_ffn_act, st));
    } else if (w.nmask) {       // the ReLU decision as one bit per activation, written by fc1's epilogue in its own tile order
        TRY(linear(L, M, Ff, D, ws + w.h3, L.w_fc1, L.b_fc1, ws + w.a, nullptr, nullptr, ws + w.amask, 0, S2T_ACT_RELU_MASK, pact, c->seed_ffn_act, st));
    } else {
        TRY(linear(L, M, Ff, D, ws + w.h3, L.w_fc1, L.b_fc1, ws + w.a, nullptr, nullptr, nullptr, 0, S2T_ACT_RELU, pact, c->seed_ffn_act, st));
    }
    TRY(linear(L, M, D, Ff, ws + w.a, L.w_fc2, L.b_fc2, c->y, xin, nullptr, nullptr, 0, S2T_ACT_NONE, p, c->seed_ffn_out, st));
    return S2T_OK;
}

extern "C" int s2t_layer_bwd(const S2TLayerDesc* Lp, S2TLayerCall* c, void* st) {
    if (bad(Lp) || !c || !c->x || !c->ws || !c->dy || !c->dx || !c->tmp || (Lp->decoder && (!c->enc || !c->denc))) return S2T_EINVAL;
    if (c->nxt_p > 0.f && !c->dx_drop) return S2T_EINVAL;
    const S2TLayerDesc& L = *Lp;
    const Ws w = layout(L, c->training);
    const Tmp t = tmp_layout(L);
    const char* ws = static_cast<const char*>(c->ws);
    char* tm = static_cast<char*>(c->tmp);
    const int M = L.T * L.B, D = L.D, Ff = L.ffn, H = L.heads, d = D / H;
    const float scale = (float)pow((double)d, -0.5);         // kernels.attn_fwd: d ** -0.5
    const float pa = c->training ? L.p_attn : 0.f, p = c->training ? L.p_drop : 0.f, pact = c->training ? L.p_act : 0.f;
    const size_t nMD = (size_t)M * D;
    const long ts3 = (long)L.B * 3 * D, bs3 = 3 * D, ts1 = (long)L.B * D, bs1 = D;
    // ---- FFN block (engine.ffn_block_bwd)
    const char* x3 = L.decoder ? ws + w.y2 : ws + w.y1;                    // input of the FFN block
    const void* d3 = c->dy_drop;
    if (!d3) {
        if (p > 0.f) { TRY(s2t_dropout(L.dtype, c->dy, tm + t.d3, nMD, p, c->seed_ffn_out, st)); d3 = tm + t.d3; }
        else d3 = c->dy;
    }
    TRY(queue_dw(*c, d3, ws + w.a, L.g_w_fc2, L.g_b_fc2, D, Ff, M));
    if (!L.gelu) {
        // a = relu(z) * keep / (1 - p): a > 0 <=> active and kept; the 1/(1-p) factor goes in alpha
        const float alpha = 1.f / (1.f - pact);
        if (w.nmask) TRY(linear_dx(L, M, D, Ff, d3, L.w_fc2, tm + t.da, ws + w.amask, 0, S2T_ACT_RELU_BWD_MASK, alpha, 0, st));
        else TRY(linear_dx(L, M, D, Ff, d3, L.w_fc2, tm + t.da, ws + w.a, Ff, S2T_ACT_RELU_BWD, alpha, 0, st));
    } else {
        TRY(linear_dx(L, M, D, Ff, d3, L.w_fc2, tm + t.da, ws + w.pre, Ff, S2T_ACT_GELU_BWD, 1.f, 0, st));
        if (pact > 0.f) TRY(s2t_dropout(L.dtype, tm + t.da, tm + t.da, (size_t)M * Ff, pact, c->seed_ffn_act, st));
    }
    TRY(queue_dw(*c, tm + t.da, ws + w.h3, L.g_w_fc1, L.g_b_fc1, Ff, D, M));
    TRY(linear_dx(L, M, Ff, D, tm + t.da, L.w_fc1, tm + t.dh3, nullptr, 0, S2T_ACT_NONE, 1.f, 0, st));
    const float* st3 = reinterpret_cast<const float*>(ws + w.st3);
    // the LayerNorm backward also writes dropout(dx) with the mask of the block that consumes dx next
    const unsigned long long seed_up = L.decoder ? c->seed_xa_out : c->seed_sa_out;
    TRY(s2t_layernorm_bwd(L.dtype, tm + t.dh3, x3, st3, st3 + M, L.ln2_g, c->dy, tm + t.dx2, L.g_ln2_g, L.g_ln2_b, M, D,
                          p > 0.f ? tm + t.dx2d : nullptr, p, seed_up, st));
    const char* dcur = tm + t.dx2;
    const char* dcur_d = p > 0.f ? tm + t.dx2d : tm + t.dx2;
    // ---- encoder-attention block (engine.cross_attn_block_bwd)
    if (L.decoder) {
        const int Ms = L.Ts * L.B;
        TRY(queue_dw(*c, dcur_d, ws + w.ctx2, L.g_w_xo, L.g_b_xo, D, D, M));
        TRY(linear_dx(L, M, D, D, dcur_d, L.w_xo, tm + t.dctx2, nullptr, 0, S2T_ACT_NONE, 1.f, 0, st));
        const char* kv = ws + w.kv;
        char* dkv = tm + t.dkv;
        const long ts2 = (long)L.B * 2 * D, bs2 = 2 * D;
        TRY(s2t_attn_bwd(L.dtype, d, L.B, H, L.T, L.Ts, ws + w.q, ts1, bs1, kv, ts2, bs2, kv + (size_t)D * 2, ts2, bs2, ws + w.ctx2, ts1, bs1,
                         tm + t.dctx2, ts1, bs1, reinterpret_cast<const float*>(ws + w.lse2), reinterpret_cast<float*>(tm + t.delta),
                         tm + t.dq, ts1, bs1, dkv, ts2, bs2, dkv + (size_t)D * 2, ts2, bs2, c->enc_klen, 0, 0, scale, pa, c->seed_xa_attn, st));
        TRY(queue_dw(*c, dkv, c->enc, L.g_w_xkv, L.g_b_xkv, 2 * D, D, Ms));
        TRY(linear_dx(L, Ms, 2 * D, D, dkv, L.w_xkv, c->denc, nullptr, 0, S2T_ACT_NONE, 1.f, c->denc_accumulate, st));
        TRY(queue_dw(*c, tm + t.dq, ws + w.h2, L.g_w_xq, L.g_b_xq, D, D, M));
        TRY(linear_dx(L, M, D, D, tm + t.dq, L.w_xq, tm + t.dh2, nullptr, 0, S2T_ACT_NONE, 1.f, 0, st));
        const float* st2 = reinterpret_cast<const float*>(ws + w.st2);
        TRY(s2t_layernorm_bwd(L.dtype, tm + t.dh2, ws + w.y1, st2, st2 + M, L.lnx_g, dcur, tm + t.dx1, L.g_lnx_g, L.g_lnx_b, M, D,
                              p > 0.f ? tm + t.dx1d : nullptr, p, c->seed_sa_out, st));
        dcur = tm + t.dx1;
        dcur_d = p > 0.f ? tm + t.dx1d : tm + t.dx1;
    }
    // ---- self-attention block (engine.self_attn_block_bwd)
    TRY(queue_dw(*c, dcur_d, ws + w.ctx, L.g_w_o, L.g_b_o, D, D, M));
    TRY(linear_dx(L, M, D, D, dcur_d, L.w_o, tm + t.dctx, nullptr, 0, S2T_ACT_NONE, 1.f, 0, st));
    const char* qkv = ws + w.qkv;
    char* dqkv = tm + t.dqkv;
    TRY(s2t_attn_bwd(L.dtype, d, L.B, H, L.T, L.T, qkv, ts3, bs3, qkv + (size_t)D * 2, ts3, bs3, qkv + (size_t)2 * D * 2, ts3, bs3, ws + w.ctx, ts1, bs1,
                     tm + t.dctx, ts1, bs1, reinterpret_cast<const float*>(ws + w.lse), reinterpret_cast<float*>(tm + t.delta),
                     dqkv, ts3, bs3, dqkv + (size_t)D * 2, ts3, bs3, dqkv + (size_t)2 * D * 2, ts3, bs3, c->self_klen, L.causal, L.dist_penalty, scale, pa,
                     c->seed_sa_attn, st));
    TRY(queue_dw(*c, dqkv, ws + w.h1, L.g_w_qkv, L.g_b_qkv, 3 * D, D, M));
    TRY(linear_dx(L, M, 3 * D, D, dqkv, L.w_qkv, tm + t.dh1, nullptr, 0, S2T_ACT_NONE, 1.f, 0, st));
    const float* st1 = reinterpret_cast<const float*>(ws + w.st1);
    TRY(s2t_layernorm_bwd(L.dtype, tm + t.dh1, c->x, st1, st1 + M, L.ln1_g, dcur, c->dx, L.g_ln1_g, L.g_ln1_b, M, D,
                          c->nxt_p > 0.f ? c->dx_drop : nullptr, c->nxt_p, c->nxt_seed, st));
    return S2T_OK;
}
