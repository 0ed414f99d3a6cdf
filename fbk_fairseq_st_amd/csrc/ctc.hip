// CTC-based compression of the encoder sequence and the CTC loss.
// Reference: examples/speech_recognition/models/conv_transformer.py:278-291,385-426
//            (average_same_ctc_features / CTCCompressStrategy) and
//            examples/speech_recognition/criterions/CTC_loss.py:101-175 (F.log_softmax + F.ctc_loss,
//            reduction="sum", zero_infinity=True).
// The integer part (arg-max of the softmax output with first-index ties, run-length collapse inside
// src_lengths[b], new lengths) is bit-exact by construction; nothing here goes through a dense
// (B, T, T') weight matrix: compression is a segment-weighted sum, its backward a gather.
#include "common.hpp"

// ------------------------------------------------------------------ softmax arg-max per frame
// logits [T][B][V] (T dtype) -> pred[b][t] = first index of max softmax(logits)[v]; pmax[b][t] = that
// probability (f32).  One wavefront per (t,b) row, three passes over the row (L2-resident).
// Row statistics with 16-byte loads (rows start 16-byte aligned: padded row stride, kernels.py alloc_rows): max, then sum of
// exp(x - max).  The scalar tail covers V % (16 / sizeof(T)) and unaligned rows.
template <typename T>
__device__ __forceinline__ void row_max_sumexp(const T* __restrict__ x, int V, int lane, float& m_out, float& s_out) {
    constexpr int E = 16 / (int)sizeof(T);
    const bool vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const int nv = vec ? V / E : 0;
    float m = -INFINITY;
    for (int c = lane; c < nv; c += 64) {
        T v[E];
        *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(x + c * E);
#pragma unroll
        for (int e = 0; e < E; ++e) m = fmaxf(m, to_f32(v[e]));
    }
    for (int j = nv * E + lane; j < V; j += 64) m = fmaxf(m, to_f32(x[j]));
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < nv; c += 64) {
        T v[E];
        *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(x + c * E);
#pragma unroll
        for (int e = 0; e < E; ++e) s += expf(to_f32(v[e]) - m);
    }
    for (int j = nv * E + lane; j < V; j += 64) s += expf(to_f32(x[j]) - m);
    m_out = m; s_out = wave_sum(s);
}

// Single-pass form for bf16 rows of up to 64 * 8 * NC units whose rows start 16-byte aligned: the whole row is fetched ONCE into
// registers (NC 16-byte vectors per lane, all requested before the first is used) and the three sweeps -- max, sum of exp, arg-max
// of the softmax values -- run on the registers.  (The three-pass kernel below re-read the 10 KB rows from L2 / HBM: 102 us for
// 24,000 x 5,001, 2.4 TB/s of useful traffic.)  Same arithmetic, same results as the three-pass kernel.
template <int NC>
__global__ __launch_bounds__(256) void ctc_argmax_row_kernel(const bf16* __restrict__ logits, int* __restrict__ pred, float* __restrict__ pmax,
                                                             float* __restrict__ lse_out, int Tn, int B, int V, int ld) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)Tn * B) return;
    const int t = (int)(row / B), b = (int)(row % B);
    const bf16* x = logits + row * ld;
    const int nv = (V + 7) / 8;                                    // the last vector may reach into the row padding (ld >= 8 nv)
    u32x4 raw[NC];
    const int tail = V & 7;                                            // real elements of the last vector (0: all eight)
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int c = lane + 64 * k;
        raw[k] = c < nv ? *reinterpret_cast<const u32x4*>(x + c * 8) : (u32x4){0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u};   // -inf
        if (tail && c == nv - 1) {                                     // padding columns of the last vector -> -inf, once
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (2 * w >= tail) raw[k][w] = 0xFF80FF80u;
                else if (2 * w + 1 >= tail) raw[k][w] = (raw[k][w] & 0x0000FFFFu) | 0xFF800000u;
            }
        }
    }
    auto elem = [&](int k, int e) -> float {                       // bf16 -> f32 is a shift / mask of the packed word
        const uint32_t w = raw[k][e >> 1];
        return __builtin_bit_cast(float, (e & 1) ? (w & 0xFFFF0000u) : (w << 16));
    };
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, elem(k, e));
    m = wave_max(m);
    // sum of exp(x - m) with the hardware exp2 (two instructions per element; the libm-accurate expf made this sweep 3/4 of the
    // kernel's time).  Only the common denominator comes from here: which index wins is decided below on accurately evaluated
    // numerators, and the reported probability / log-sum-exp move by ~1e-7 relative.
    const float ml2 = m * 1.44269504088896f;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += __builtin_amdgcn_exp2f(__builtin_fmaf(elem(k, e), 1.44269504088896f, -ml2));     // exp2(-inf) = 0 for the padding
    s = wave_sum(s);
    float best = -1.f; int bi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xv = elem(k, e);
            if (xv >= m - 1e-3f) {                                   // only these can round to the top probability (padding is -inf)
                const float pr = expf(xv - m) / s;
                const int j = (lane + 64 * k) * 8 + e;
                if (pr > best || (pr == best && j < bi)) { best = pr; bi = j; }
            }
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) {
        pred[(long)b * Tn + t] = bi; pmax[(long)b * Tn + t] = best;
        if (lse_out) lse_out[row] = m + logf(s);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ctc_argmax_kernel(const T* __restrict__ logits, int* __restrict__ pred,
                                                         float* __restrict__ pmax, float* __restrict__ lse_out, int Tn, int B, int V, int ld) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)Tn * B) return;
    const int t = (int)(row / B), b = (int)(row % B);
    const T* x = logits + row * ld;
    float m, s;
    row_max_sumexp<T>(x, V, lane, m, s);
    // arg-max of the softmax VALUES (first index among equal f32 probabilities, as torch compares them): only logits within 1e-3 of
    // the maximum can round to the top probability, everything else is skipped without evaluating exp / divide
    float best = -1.f; int bi = 0x7fffffff;
    auto consider = [&](float xv, int j) {
        if (xv >= m - 1e-3f) {
            const float p = expf(xv - m) / s;                    // the softmax value torch would compare
            if (p > best || (p == best && j < bi)) { best = p; bi = j; }
        }
    };
    constexpr int E = 16 / (int)sizeof(T);
    const int nv = (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? V / E : 0;
    for (int c = lane; c < nv; c += 64) {
        T v[E];
        *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(x + c * E);
#pragma unroll
        for (int e = 0; e < E; ++e) consider(to_f32(v[e]), c * E + e);
    }
    for (int j = nv * E + lane; j < V; j += 64) consider(to_f32(x[j]), j);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) {
        pred[(long)b * Tn + t] = bi; pmax[(long)b * Tn + t] = best;
        if (lse_out) lse_out[row] = m + logf(s);              // what row_lse_kernel would store: the CTC loss over the same logits reuses it
    }
}

// ------------------------------------------------------------------ run-length collapse + weights
// One wavefront per utterance.  seg[b][t] = run index (-1 beyond len), run_start/run_len[b][j],
// new_len[b]; w[b][t] = weight of frame t inside its run:
//   strategy 0 avg:      1/run_len            (python double 1.0/n rounded to f32, :394)
//   strategy 1 weighted: p_t / sum_run p      (:407-409)
//   strategy 2 softmax:  softmax_run(p)_t / sum (= softmax over the run, :422-424)
__global__ __launch_bounds__(64) void ctc_rle_kernel(const int* __restrict__ pred, const float* __restrict__ pmax,
                                                     const long long* __restrict__ len, int* __restrict__ seg,
                                                     int* __restrict__ run_start, int* __restrict__ run_len,
                                                     long long* __restrict__ new_len, float* __restrict__ w,
                                                     int Tn, int strategy) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int L = (int)min((long long)Tn, len[b]);
    const int* p = pred + (long)b * Tn;
    int* sg = seg + (long)b * Tn; int* rs = run_start + (long)b * Tn; int* rl = run_len + (long)b * Tn;
    int base = 0;                                   // runs before this 64-frame chunk
    for (int t0 = 0; t0 < Tn; t0 += 64) {
        const int t = t0 + lane;
        const bool valid = t < L;
        const bool head = valid && (t == 0 || p[t] != p[t - 1]);
        const unsigned long long heads = __ballot(head);
        const int before = __popcll(heads & ((1ull << lane) - 1ull)) + (head ? 1 : 0);
        const int id = valid ? base + before - 1 : -1;
        if (t < Tn) { sg[t] = id; rs[t] = 0; rl[t] = 0; }
        base += __popcll(heads);
    }
    __syncthreads();
    for (int t = lane; t < L; t += 64) {             // run boundaries
        const int id = sg[t];
        if (t == 0 || sg[t - 1] != id) rs[id] = t;
        if (t == L - 1 || sg[t + 1] != id) rl[id] = t + 1;   // temporarily the end index
    }
    __syncthreads();
    for (int j = lane; j < base; j += 64) rl[j] = rl[j] - rs[j];
    if (lane == 0) new_len[b] = base;
    __syncthreads();
    float* wb = w + (long)b * Tn;
    const float* pm = pmax + (long)b * Tn;
    for (int t = lane; t < Tn; t += 64) {
        if (t >= L) { wb[t] = 0.f; continue; }
        const int id = sg[t], s0 = rs[id], n = rl[id];
        if (strategy == 0) wb[t] = (float)(1.0 / (double)n);
        else if (strategy == 1) {
            float sum = 0.f;
            for (int k = 0; k < n; ++k) sum += pm[s0 + k];
            wb[t] = pm[t] / sum;
        } else {
            float mx = -INFINITY;
            for (int k = 0; k < n; ++k) mx = fmaxf(mx, pm[s0 + k]);
            float sum = 0.f;
            for (int k = 0; k < n; ++k) sum += expf(pm[s0 + k] - mx);
            const float sm = expf(pm[t] - mx) / sum;
            float tot = 0.f;
            for (int k = 0; k < n; ++k) tot += expf(pm[s0 + k] - mx) / sum;
            wb[t] = sm / tot;
        }
    }
}

// ------------------------------------------------------------------ segment-weighted sum (forward)
// out[j][b][:] = sum_{t in run j} w[b][t] * x[t][b][:]   for j < new_len[b], zeros up to Tout
template <typename T>
__global__ __launch_bounds__(256) void ctc_compress_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                               const int* __restrict__ run_start, const int* __restrict__ run_len,
                                                               const long long* __restrict__ new_len, T* __restrict__ out,
                                                               int Tn, int B, int D, int Tout) {
    const int j = blockIdx.x, b = blockIdx.y;
    T* o = out + ((long)j * B + b) * D;
    if (j >= new_len[b]) { for (int d = threadIdx.x; d < D; d += 256) o[d] = from_f32<T>(0.f); return; }
    const int s0 = run_start[(long)b * Tn + j], n = run_len[(long)b * Tn + j];
    constexpr int E = 16 / (int)sizeof(T);
    if (D % E == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        // a run can span hundreds of frames (every frame of an utterance predicted blank): the frames of the run are dealt to the
        // four waves (a serial walk is one memory latency per frame), 16 bytes of channels per lane, partial sums meet in LDS
        __shared__ float part[4][64 * E];
        const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
        for (int d0 = lane * E; d0 < D; d0 += 64 * E) {
            float acc[E];
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = 0.f;
            for (int k = slot; k < n; k += 4) {
                const float wt = w[(long)b * Tn + s0 + k];
                T v[E];
                *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(x + ((long)(s0 + k) * B + b) * D + d0);
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] += wt * to_f32(v[e]);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) part[slot][lane * E + e] = acc[e];
            __syncthreads();
            if (slot == 0) {
                T r[E];
#pragma unroll
                for (int e = 0; e < E; ++e)
                    r[e] = from_f32<T>((part[0][lane * E + e] + part[1][lane * E + e]) + (part[2][lane * E + e] + part[3][lane * E + e]));
                *reinterpret_cast<u32x4*>(o + d0) = *reinterpret_cast<const u32x4*>(r);
            }
            __syncthreads();
        }
        return;
    }
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc += w[(long)b * Tn + s0 + k] * to_f32(x[((long)(s0 + k) * B + b) * D + d]);
        o[d] = from_f32<T>(acc);
    }
}

// backward: dx[t][b][:] (+)= w[b][t] * dout[seg[b][t]][b][:]   (weights carry no gradient, :280)
template <typename T>
__global__ __launch_bounds__(256) void ctc_compress_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ w,
                                                               const int* __restrict__ seg, T* __restrict__ dx,
                                                               int Tn, int B, int D, int accumulate) {
    const int t = blockIdx.x, b = blockIdx.y;
    const int id = seg[(long)b * Tn + t];
    const float wt = id >= 0 ? w[(long)b * Tn + t] : 0.f;
    T* o = dx + ((long)t * B + b) * D;
    for (int d = threadIdx.x; d < D; d += 256) {
        float v = id >= 0 ? wt * to_f32(dout[((long)id * B + b) * D + d]) : 0.f;
        if (accumulate) v += to_f32(o[d]);
        o[d] = from_f32<T>(v);
    }
}

// ------------------------------------------------------------------ CTC loss
// Pass 1: per (t,b) row log-sum-exp of the logits (f32) -> lse[t][b].
template <typename T>
__global__ __launch_bounds__(256) void row_lse_kernel(const T* __restrict__ logits, float* __restrict__ lse, long rows, int V, int ld) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* x = logits + row * ld;
    float m, s;
    row_max_sumexp<T>(x, V, lane, m, s);
    if (lane == 0) lse[row] = m + logf(s);
}

// The alpha / beta recursion runs in LOG2 units with a finite "never" (CTC_NEVER) instead of -inf: a dependent chain of 375 steps
// per utterance whose latency IS the kernel's duration (a lone wave issues one instruction per ~8 cycles), so every instruction
// in the step counts.  log2(2^a + 2^b [+ 2^c]) on the hardware exp2 / log2 units takes no scaling multiplies; with CTC_NEVER the
// all-unreachable case needs no test (2^0 three times, log2 3 added to -1e30 is absorbed: one ulp there is 7.6e22) and no inf - inf
// can arise.  Absolute error ~1e-6 per step on values of magnitude 1e2: far inside the 1e-4 relative budget of the loss.
#define CTC_NEVER (-1.0e30f)
#define CTC_NEVER_TEST (-1.0e29f)          // "reachable" = above this (unreachable values only move further down)
__device__ __forceinline__ float l2ae2(float a, float b) {
    return fmaxf(a, b) + __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(-fabsf(a - b)));
}
__device__ __forceinline__ float l2ae3(float a, float b, float c) {
    const float m = fmaxf(fmaxf(a, b), c);
    return m + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m) + __builtin_amdgcn_exp2f(c - m));
}

// Lane i <- lane i-1 (lane 0 <- never) and lane i <- lane i+1 (lane 63 <- never) as DPP wave shifts: ~2 VALU issues where a
// ds_bpermute shuffle is an LDS round trip, and the neighbour's value sits on the recursion's dependent chain at every step.
__device__ __forceinline__ float lane_before(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(CTC_NEVER), __float_as_int(v), 0x138, 0xf, 0xf, false));   // wave_shr:1
}
__device__ __forceinline__ float lane_after(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(CTC_NEVER), __float_as_int(v), 0x130, 0xf, 0xf, false));   // wave_shl:1
}
// wave maximum, wave-uniform: an inclusive row scan (row_shr 1, 2, 4, 8), the rows' last lanes carried across (row_bcast 15, 31)
__device__ __forceinline__ float wave_max_dpp(float v) {
    // in place: lanes whose source lies outside the row / wave keep their value (bound_ctrl off disables the write).  A VALU write
    // followed by a DPP read of the same register needs two wait states, which hipcc cannot see inside an asm: the s_nop 1's.
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1" : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Pass 2: alpha (wave 0) and beta (wave 1) recursions of one utterance in log2 space.
// Extended target ext[s], s < S = 2L+1 <= 64*SPL; lane owns SPL consecutive s; the neighbours needed from the adjacent lane travel
// by DPP wave shift.  Emissions lp[t][ext[s]] = (logit - lse) log2 e are gathered CH time steps at a time into LDS by the whole
// workgroup (CTC_NEVER for s >= S and frames outside the utterance: such positions then stay "never" by themselves, no per-step
// test); with one position per thread (SPL <= 2) the NEXT chunk's gather is in flight in registers while this chunk's steps run.
// Outputs: la/lb [B][T][64 SPL] (f32 log2 alpha/beta, each step's vector shifted by a per-step offset; rows are padded to the
// lanes' positions so that a step is one unconditional aligned store per lane), nll[b] in nats (+inf if infeasible).
// CTC_SPL extended-target positions per lane (1 .. 16: transcripts of up to 511 units), CTC_CH time steps of emissions per LDS chunk
// (the chunk buffer is 2 x CTC_CH x 64 x CTC_SPL floats: CTC_CH shrinks as CTC_SPL grows to stay inside 64 KB of static LDS)
//
// Round 4 took the step from ~2,900 cycles to ~?: (1) DPP shifts instead of shuffles; (2) the normalisation no longer sits on the
// chain: every fourth step is shifted by the wave maximum of the vector it STARTS from (a DPP reduction beside that step's sums),
// not every step by its own; (3) t = 0 / t = Tb-1 are the ordinary step applied to a virtual state delta(s = 0) / delta(s = S-1);
// (4) the chunk gather is prefetched; (5) log2 units, a finite "never", two-term sums at the blanks (even s never skip).
template <typename T, int CTC_SPL, int CTC_CH = (CTC_SPL <= 4 ? 16 : (CTC_SPL == 8 ? 8 : 4))>
__global__ __launch_bounds__(128) void ctc_alphabeta_kernel(const T* __restrict__ logits, const float* __restrict__ lse,
                                                            const long long* __restrict__ targets, const long long* __restrict__ tgt_len,
                                                            const int* __restrict__ in_len, float* __restrict__ la,
                                                            float* __restrict__ lb, float* __restrict__ nll, int Tn, int B,
                                                            int V, int ld, int Lmax, int blank) {
    constexpr int P = 64 * CTC_SPL;                         // positions of the wave = row stride of la / lb
    constexpr float L2E = 1.44269504088896f;
    __shared__ float em[2][CTC_CH][P];
    __shared__ int ext_s[P];
    constexpr bool PF = CTC_SPL <= 2;                       // S <= 128: thread = position, one gather per thread and chunk
    const int b = blockIdx.x, lane = threadIdx.x & 63, dir = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int L = (int)tgt_len[b], S = 2 * L + 1, Tb = min(in_len[b], Tn);
    for (int s = threadIdx.x; s < P; s += 128)
        ext_s[s] = (s < S) ? ((s & 1) ? (int)targets[(long)b * Lmax + (s >> 1)] : blank) : -1;
    __syncthreads();
    float a[CTC_SPL];
    bool skip2[CTC_SPL];          // alpha: may come from s-2 ; beta: may go to s+2 (odd positions only: blanks never skip)
#pragma unroll
    for (int i = 0; i < CTC_SPL; ++i) {
        const int s = lane * CTC_SPL + i;
        const int e = ext_s[s];
        // the state "before" the first step: alpha_{-1} = delta(s = 0), beta_{Tb} = delta(s = S-1).  One ordinary step from it gives
        // alpha_0 = {lp(blank), lp(l_1), never ...} and beta_{Tb-1} likewise (l2ae(0, never[, never]) = 0 exactly).
        a[i] = (s == (dir == 0 ? 0 : S - 1)) ? 0.f : CTC_NEVER;
        if (dir == 0) skip2[i] = (s >= 2 && s < S && e != blank && e != ext_s[s - 2]);
        else skip2[i] = (s + 2 < S && ext_s[s + 2] != blank && ext_s[s + 2] != e);
    }
    // Every step's vector is stored RELATIVE to a per-step offset and the offsets are summed in double.  In plain log space alpha and
    // beta reach -3,000 over 375 frames x 5,001 units, where one f32 ulp is 2.4e-4: alpha + beta - log P then carries ~1e-3 of
    // relative error into every posterior (measured against float64: 8e-4 in |dg|/|g|, torch's own f32 ctc_loss 6e-4).  The offset
    // of step t is the sum of the wave maxima taken every fourth step before it: the vector's maximum then stays within four steps'
    // change (each at most +log 3, at least the largest emission) of zero, and the gradient kernel normalises alpha + beta -
    // emission per frame, so the offsets never enter the posteriors; only the loss needs their sum.
    double offsum = 0.0;
    const int nch = (Tb + CTC_CH - 1) / CTC_CH;
    float xv[2 * CTC_CH], lv[2 * CTC_CH];
    // all 32 gathered loads of a position are issued from clamped (always valid) rows before any is used: guarded loads get sunk
    // under their bounds test and waited for one by one (seen in the ISA of conv1: one memory latency each)
    auto fetch = [&](int ch, int s) {
        const int col = max(ext_s[s], 0);
        const int ta0 = ch * CTC_CH, tb0 = Tb - 1 - ch * CTC_CH;
#pragma unroll
        for (int dk = 0; dk < 2 * CTC_CH; ++dk) {
            const int d = dk / CTC_CH, k = dk % CTC_CH;
            const int t = min(max(d == 0 ? ta0 + k : tb0 - k, 0), max(Tb - 1, 0));
            xv[dk] = to_f32(logits[((long)t * B + b) * ld + col]);
            lv[dk] = lse[(long)t * B + b];
        }
    };
    auto deposit = [&](int ch, int s) {
        const int ta0 = ch * CTC_CH, tb0 = Tb - 1 - ch * CTC_CH;
#pragma unroll
        for (int dk = 0; dk < 2 * CTC_CH; ++dk) {
            const int d = dk / CTC_CH, k = dk % CTC_CH;
            const int t = d == 0 ? ta0 + k : tb0 - k;
            em[d][k][s] = (s < S && t >= 0 && t < Tb) ? (xv[dk] - lv[dk]) * L2E : CTC_NEVER;
        }
    };
    if (PF && nch > 0 && (int)threadIdx.x < P) fetch(0, threadIdx.x);
    // this wave's output row of the first step and its stride (alpha walks forward, beta backward)
    float* out = (dir == 0 ? la : lb) + ((long)b * Tn + (dir == 0 ? 0 : max(Tb - 1, 0))) * P + lane * CTC_SPL;
    const long ostep = dir == 0 ? P : -P;
    for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();
        if (PF) { if ((int)threadIdx.x < P) deposit(ch, threadIdx.x); }
        else {
            for (int s = threadIdx.x; s < P; s += 128) {      // no index divisions: (direction, step) pairs outside, position = thread
                fetch(ch, s);
#pragma unroll
                for (int dk = 0; dk < 2 * CTC_CH; ++dk) asm volatile("" : "+v"(xv[dk]), "+v"(lv[dk]));
                deposit(ch, s);
            }
        }
        __syncthreads();
        if (PF && ch + 1 < nch && (int)threadIdx.x < P) fetch(ch + 1, threadIdx.x);
        // emissions travel one step ahead of their use (an LDS read issued at the top of a step is ~100 cycles on the chain)
        float e_next[CTC_SPL];
#pragma unroll
        for (int i = 0; i < CTC_SPL; ++i) e_next[i] = em[dir][0][lane * CTC_SPL + i];
        const int nk = min(CTC_CH, Tb - ch * CTC_CH);
        for (int k = 0; k < nk; ++k) {
            float e[CTC_SPL];
#pragma unroll
            for (int i = 0; i < CTC_SPL; ++i) e[i] = e_next[i];
            {
                const int kn = min(k + 1, CTC_CH - 1);
#pragma unroll
                for (int i = 0; i < CTC_SPL; ++i) e_next[i] = em[dir][kn][lane * CTC_SPL + i];
            }
            if ((k & 3) == 0) {
                float loc = a[0];
#pragma unroll
                for (int i = 1; i < CTC_SPL; ++i) loc = fmaxf(loc, a[i]);
                float sh = wave_max_dpp(loc);
                sh = sh > CTC_NEVER_TEST ? sh : 0.f;                      // (nothing reachable: an infeasible alignment, nothing to shift)
                offsum += (double)sh;
#pragma unroll
                for (int i = 0; i < CTC_SPL; ++i) e[i] -= sh;
            }
            float n[CTC_SPL];
            if (dir == 0) {
                // alpha[s-1], alpha[s-2] of the previous lane: lanes own whole (blank, unit) pairs when SPL is even, and then only
                // s-1 of the lane's first position (a blank: no skip) crosses lanes
                const float p1 = lane_before(a[CTC_SPL - 1]);
                const float p2 = CTC_SPL >= 2 ? CTC_NEVER : lane_before(p1);
#pragma unroll
                for (int i = 0; i < CTC_SPL; ++i) {
                    const float m1 = (i >= 1) ? a[i - 1] : p1;
                    const float m2 = (i >= 2) ? a[i - 2] : (i == 1 ? p1 : p2);
                    if (CTC_SPL >= 2 && (i & 1) == 0) n[i] = l2ae2(a[i], m1) + e[i];
                    else n[i] = l2ae3(a[i], m1, skip2[i] ? m2 : CTC_NEVER) + e[i];
                }
            } else {
                // beta[s+1], beta[s+2] of the next lane: its first position is a blank (s+1 of this lane's last, a unit) and its
                // second a unit (s+2 of it: the skip)
                const float p1 = lane_after(a[0]);
                const float p2 = CTC_SPL >= 2 ? lane_after(a[CTC_SPL >= 2 ? 1 : 0]) : lane_after(p1);
#pragma unroll
                for (int i = 0; i < CTC_SPL; ++i) {
                    const float m1 = (i + 1 < CTC_SPL) ? a[i + 1] : p1;
                    const float m2 = (i + 2 < CTC_SPL) ? a[i + 2] : (i + 2 == CTC_SPL ? p1 : p2);
                    if (CTC_SPL >= 2 && (i & 1) == 0) n[i] = l2ae2(a[i], m1) + e[i];
                    else n[i] = l2ae3(a[i], m1, skip2[i] ? m2 : CTC_NEVER) + e[i];
                }
            }
#pragma unroll
            for (int i = 0; i < CTC_SPL; ++i) a[i] = n[i];
            if constexpr (CTC_SPL == 1) out[0] = a[0];
            else if constexpr (CTC_SPL == 2) *reinterpret_cast<float2*>(out) = float2{a[0], a[1]};
            else {
#pragma unroll
                for (int i = 0; i < CTC_SPL; i += 4) *reinterpret_cast<float4*>(out + i) = float4{a[i], a[i + 1], a[i + 2], a[i + 3]};
            }
            out += ostep;
        }
    }
    if (dir == 0) {
        // log-likelihood = logaddexp(alpha[Tb-1][S-1], alpha[Tb-1][S-2])
        float v = CTC_NEVER;
#pragma unroll
        for (int i = 0; i < CTC_SPL; ++i) {
            const int s = lane * CTC_SPL + i;
            if (s < S && s >= S - 2) v = l2ae2(v, a[i]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = l2ae2(v, __shfl_xor(v, o));
        if (lane == 0) nll[b] = (Tb > 0 && v > CTC_NEVER_TEST) ? (float)(-((double)v + offsum) * 0.693147180559945309) : INFINITY;
    }
}

// Pass 3: gradient w.r.t. the logits, one workgroup per (t,b) row:
//   g[c] = softmax[c] - sum_{s: ext[s]=c} post[s],   post[s] = 2^(la + lb - lp2[ext s]) / sum_s' 2^(la + lb - lp2[ext s'])   (log2 units)
// for t < in_len[b] and finite nll, else 0.  (sum_s alpha_t(s) beta_t(s) / y_t(ext s) = P for every t: normalising per frame is
// the same posterior as exp(la + lb - lp + nll) and is indifferent to the per-step offsets the recursion subtracts.)
// Also the summed loss (atomic add of the finite nll's, done by the t = 0 rows).
template <typename T>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const T* __restrict__ logits, const float* __restrict__ lse,
                                                       const long long* __restrict__ targets, const long long* __restrict__ tgt_len,
                                                       const int* __restrict__ in_len, const float* __restrict__ la,
                                                       const float* __restrict__ lb, const float* __restrict__ nll,
                                                       T* __restrict__ grad, float* __restrict__ loss_sum, int Tn, int B,
                                                       int V, int ld, int Lmax, int Srow, int blank, float gscale,
                                                       const float* __restrict__ gscale_dev) {
    extern __shared__ float occ[];                      // [V]
    if (gscale_dev) gscale *= gscale_dev[0];            // upstream gradient of the loss (a device scalar produced by autograd)
    const int t = blockIdx.x, b = blockIdx.y;
    const long row = (long)t * B + b;
    T* g = grad + row * ld;
    const float nl = nll[b];
    const bool live = t < min(in_len[b], Tn) && nl < INFINITY;
    if (loss_sum && t == 0 && threadIdx.x == 0 && nl < INFINITY) atomicAdd(loss_sum, nl);
    if (!live) { for (int c = threadIdx.x; c < V; c += 256) g[c] = from_f32<T>(0.f); return; }
    const float ls = lse[row];
    const T* x = logits + row * ld;
    for (int c = threadIdx.x; c < V; c += 256) occ[c] = 0.f;
    __syncthreads();
    const int S = 2 * (int)tgt_len[b] + 1;
    __shared__ float redm[4], reds[4];
    // w[s] = la + lb - lp: up to four states per thread (S <= 1023); frame maximum, then frame sum
    float wv[4]; int wc[4];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int s = threadIdx.x + 256 * k;
        wv[k] = -INFINITY; wc[k] = 0;
        if (s < S) {
            wc[k] = (s & 1) ? (int)targets[(long)b * Lmax + (s >> 1)] : blank;
            const float lab = la[((long)b * Tn + t) * Srow + s] + lb[((long)b * Tn + t) * Srow + s];      // log2 units
            if (lab > CTC_NEVER_TEST) wv[k] = lab - (to_f32(x[wc[k]]) - ls) * 1.44269504088896f;
        }
        mx = fmaxf(mx, wv[k]);
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    float ev[4], sm = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { ev[k] = wv[k] > -INFINITY ? __builtin_amdgcn_exp2f(wv[k] - mx) : 0.f; sm += ev[k]; }
    sm = wave_sum(sm);
    if ((threadIdx.x & 63) == 0) reds[threadIdx.x >> 6] = sm;
    __syncthreads();
    sm = (reds[0] + reds[1]) + (reds[2] + reds[3]);
    const float inv = sm > 0.f ? 1.f / sm : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (ev[k] > 0.f) atomicAdd(&occ[wc[k]], ev[k] * inv);
    __syncthreads();
    constexpr int E = 16 / (int)sizeof(T);
    const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g)) & 15) == 0;
    const int nv = vec ? V / E : 0;
    // softmax(x)[c] = exp2(x log2 e - lse log2 e) on the hardware exp2 (one quarter-rate instruction; expf costs ~15 VALU per element, and
    // this loop runs over T x B x V = 120 M of them): relative error ~2e-7, far below what the posteriors carry
    constexpr float L2E = 1.44269504088896f;
    const float ls2 = ls * L2E;
    for (int c = threadIdx.x; c < nv; c += 256) {
        T v[E], o[E];
        *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(x + c * E);
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = from_f32<T>((__builtin_amdgcn_exp2f(__builtin_fmaf(to_f32(v[e]), L2E, -ls2)) - occ[c * E + e]) * gscale);
        *reinterpret_cast<u32x4*>(g + c * E) = *reinterpret_cast<const u32x4*>(o);
    }
    for (int c = nv * E + threadIdx.x; c < V; c += 256)
        g[c] = from_f32<T>((__builtin_amdgcn_exp2f(__builtin_fmaf(to_f32(x[c]), L2E, -ls2)) - occ[c]) * gscale);
}

// ------------------------------------------------------------------ C ABI
extern "C" int s2t_ctc_argmax(int dtype, const void* logits, int* pred, float* pmax, float* lse, int T, int B, int V, int ld, void* stream) {
    const long rows = (long)T * B;
    if (rows <= 0) return S2T_OK;
    if (!logits || !pred || !pmax || V <= 0 || ld < V) return S2T_EINVAL;
    dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    const int nv8 = (V + 7) / 8;
    if (dtype == S2T_BF16 && !((uintptr_t)logits & 15) && !(ld & 7) && ld >= nv8 * 8 && nv8 <= 64 * 16) {
        // the row fits the registers of one wave: single pass (NC = vectors per lane)
        if (nv8 <= 64 * 4) hipLaunchKernelGGL(ctc_argmax_row_kernel<4>, grid, dim3(256), 0, st, (const bf16*)logits, pred, pmax, lse, T, B, V, ld);
        else if (nv8 <= 64 * 10) hipLaunchKernelGGL(ctc_argmax_row_kernel<10>, grid, dim3(256), 0, st, (const bf16*)logits, pred, pmax, lse, T, B, V, ld);
        else hipLaunchKernelGGL(ctc_argmax_row_kernel<16>, grid, dim3(256), 0, st, (const bf16*)logits, pred, pmax, lse, T, B, V, ld);
    }
    else if (dtype == S2T_BF16) hipLaunchKernelGGL(ctc_argmax_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)logits, pred, pmax, lse, T, B, V, ld);
    else if (dtype == S2T_F32) hipLaunchKernelGGL(ctc_argmax_kernel<float>, grid, dim3(256), 0, st, (const float*)logits, pred, pmax, lse, T, B, V, ld);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_ctc_rle(const int* pred, const float* pmax, const long long* len, int* seg, int* run_start,
                           int* run_len, long long* new_len, float* w, int T, int B, int strategy, void* stream) {
    if (B <= 0 || T <= 0) return S2T_OK;
    if (!pred || !pmax || !len || !seg || !run_start || !run_len || !new_len || !w || strategy < 0 || strategy > 2) return S2T_EINVAL;
    hipLaunchKernelGGL(ctc_rle_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, pred, pmax, len, seg, run_start, run_len, new_len, w, T, strategy);
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_ctc_compress_fwd(int dtype, const void* x, const float* w, const int* run_start, const int* run_len,
                                    const long long* new_len, void* out, int T, int B, int D, int Tout, void* stream) {
    if (B <= 0 || Tout <= 0) return S2T_OK;
    if (!x || !w || !run_start || !run_len || !new_len || !out || Tout > T) return S2T_EINVAL;
    dim3 grid(Tout, B);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2T_BF16) hipLaunchKernelGGL(ctc_compress_fwd_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, w, run_start, run_len, new_len, (bf16*)out, T, B, D, Tout);
    else if (dtype == S2T_F32) hipLaunchKernelGGL(ctc_compress_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, w, run_start, run_len, new_len, (float*)out, T, B, D, Tout);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_ctc_compress_bwd(int dtype, const void* dout, const float* w, const int* seg, void* dx, int T, int B,
                                    int D, int accumulate, void* stream) {
    if (B <= 0 || T <= 0) return S2T_OK;
    if (!dout || !w || !seg || !dx) return S2T_EINVAL;
    dim3 grid(T, B);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S2T_BF16) hipLaunchKernelGGL(ctc_compress_bwd_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)dout, w, seg, (bf16*)dx, T, B, D, accumulate);
    else if (dtype == S2T_F32) hipLaunchKernelGGL(ctc_compress_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)dout, w, seg, (float*)dx, T, B, D, accumulate);
    else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

// Fused log-softmax + CTC loss (sum, zero_infinity) + gradient w.r.t. the logits.
//   logits / grad rows have stride ld >= V elements (pad ld to a multiple of 8 so GEMMs can vector-load them)
//   logits [T][B][V]; targets [B][Lmax] int64 (first tgt_len[b] entries); in_len [B] int32
//   workspaces: lse [T*B] f32, la/lb [B*T*S2T_CTC_ROW(Lmax)] f32 (rows padded to the recursion wave's positions), nll [B] f32
//   outputs: grad [T][B][V] (dtype), loss_sum[0] += sum of finite nll  (caller zeroes it)
// forward-only calls: loss_sum[0] += sum of the finite nll's (the gradient pass does it when both run in one call)
__global__ void ctc_loss_sum_kernel(const float* __restrict__ nll, int B, float* __restrict__ loss_sum) {
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) { const float v = nll[b]; if (v < INFINITY) s += v; }
    s = wave_sum(s);
    if (threadIdx.x == 0) atomicAdd(loss_sum, s);
}

extern "C" int s2t_ctc_loss(int dtype, const void* logits, const long long* targets, const long long* tgt_len,
                            const int* in_len, float* lse, float* la, float* lb, float* nll, void* grad,
                            float* loss_sum, int T, int B, int V, int ld, int Lmax, int blank, float grad_scale,
                            int phase, const float* grad_scale_dev, void* stream) {
    if (B <= 0 || T <= 0) return S2T_OK;
    const bool lse_given = (phase & 4) != 0;               // lse already holds the rows' log-sum-exps (s2t_ctc_argmax wrote them)
    phase &= 3;
    if (!logits || !targets || !tgt_len || !in_len || !lse || !la || !lb || !nll || ld < V || phase < 0 || phase > 2) return S2T_EINVAL;
    if ((phase != 2 && !loss_sum) || (phase != 1 && !grad)) return S2T_EINVAL;
    const bool fwd = phase != 2, bwd = phase != 1;
    float* const lsum_grad = phase == 0 ? loss_sum : nullptr;      // the gradient pass adds up the loss only when it is the same call
    const int Smax = 2 * Lmax + 1;
    if (Smax > 64 * 16) return S2T_ENOTSUP;               // transcripts longer than 511 units (S2T_CTC_MAX_TARGET in the header)
    const int spl = Smax <= 64 ? 1 : (Smax <= 128 ? 2 : (Smax <= 256 ? 4 : (Smax <= 512 ? 8 : 16)));   // extended-target positions per lane
    if ((size_t)V * 4 > 160 * 1024 - 1024) return S2T_ENOTSUP;   // the gradient kernel keeps one f32 row of the vocabulary in LDS (V <= 40,704)
    hipStream_t st = (hipStream_t)stream;
    const long rows = (long)T * B;
    dim3 g1((unsigned)((rows + 3) / 4)), g3(T, B);
    const size_t lds = (size_t)V * 4;
    if (dtype == S2T_BF16) {
        if (fwd) {
        if (!lse_given) hipLaunchKernelGGL(row_lse_kernel<bf16>, g1, dim3(256), 0, st, (const bf16*)logits, lse, rows, V, ld);
        if (spl == 1) hipLaunchKernelGGL((ctc_alphabeta_kernel<bf16, 1>), dim3(B), dim3(128), 0, st, (const bf16*)logits, lse, targets, tgt_len, in_len, la, lb, nll, T, B, V, ld, Lmax, blank);
        else if (spl == 2) hipLaunchKernelGGL((ctc_alphabeta_kernel<bf16, 2>), dim3(B), dim3(128), 0, st, (const bf16*)logits, lse, targets, tgt_len, in_len, la, lb, nll, T, B, V, ld, Lmax, blank);
        else if (spl == 4) hipLaunchKernelGGL((ctc_alphabeta_kernel<bf16, 4>), dim3(B), dim3(128), 0, st, (const bf16*)logits, lse, targets, tgt_len, in_len, la, lb, nll, T, B, V, ld, Lmax, blank);
        else if (spl == 8) hipLaunchKernelGGL((ctc_alphabeta_kernel<bf16, 8>), dim3(B), dim3(128), 0, st, (const bf16*)logits, lse, targets, tgt_len, in_len, la, lb, nll, T, B, V, ld, Lmax, blank);
        else hipLaunchKernelGGL((ctc_alphabeta_kernel<bf16, 16>), dim3(B), dim3(128), 0, st, (const bf16*)logits, lse, targets, tgt_len, in_len, la, lb, nll, T, B, V, ld, Lmax, blank);
        if (phase == 1) hipLaunchKernelGGL(ctc_loss_sum_kernel, dim3(1), dim3(64), 0, st, nll, B, loss_sum);
        }
        static bool attr = false;
        if (!attr && lds > 65536) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ctc_grad_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); attr = true; }
        if (bwd) hipLaunchKernelGGL(ctc_grad_kernel<bf16>, g3, dim3(256), lds, st, (const bf16*)logits, lse, targets, tgt_len, in_len, la, lb, nll, (bf16*)grad, lsum_grad, T, B, V, ld, Lmax, 64 * spl, blank, grad_scale, grad_scale_dev);
    } else if (dtype == S2T_F32) {
        if (fwd) {
        if (!lse_given) hipLaunchKernelGGL(row_lse_kernel<float>, g1, dim3(256), 0, st, (const float*)logits, lse, rows, V, ld);
        if (spl == 1) hipLaunchKernelGGL((ctc_alphabeta_kernel<float, 1>), dim3(B), dim3(128), 0, st, (const float*)logits, lse, targets, tgt_len, in_len, la, lb, nll, T, B, V, ld, Lmax, blank);
        else if (spl == 2) hipLaunchKernelGGL((ctc_alphabeta_kernel<float, 2>), dim3(B), dim3(128), 0, st, (const float*)logits, lse, targets, tgt_len, in_len, la, lb, nll, T, B, V, ld, Lmax, blank);
        else if (spl == 4) hipLaunchKernelGGL((ctc_alphabeta_kernel<float, 4>), dim3(B), dim3(128), 0, st, (const float*)logits, lse, targets, tgt_len, in_len, la, lb, nll, T, B, V, ld, Lmax, blank);
        else if (spl == 8) hipLaunchKernelGGL((ctc_alphabeta_kernel<float, 8>), dim3(B), dim3(128), 0, st, (const float*)logits, lse, targets, tgt_len, in_len, la, lb, nll, T, B, V, ld, Lmax, blank);
        else hipLaunchKernelGGL((ctc_alphabeta_kernel<float, 16>), dim3(B), dim3(128), 0, st, (const float*)logits, lse, targets, tgt_len, in_len, la, lb, nll, T, B, V, ld, Lmax, blank);
        if (phase == 1) hipLaunchKernelGGL(ctc_loss_sum_kernel, dim3(1), dim3(64), 0, st, nll, B, loss_sum);
        }
        static bool attr = false;
        if (!attr && lds > 65536) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ctc_grad_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); attr = true; }
        if (bwd) hipLaunchKernelGGL(ctc_grad_kernel<float>, g3, dim3(256), lds, st, (const float*)logits, lse, targets, tgt_len, in_len, la, lb, nll, (float*)grad, lsum_grad, T, B, V, ld, Lmax, 64 * spl, blank, grad_scale, grad_scale_dev);
    } else return S2T_ENOTSUP;
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}
