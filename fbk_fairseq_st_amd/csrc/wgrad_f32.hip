// All weight gradients of a backward pass in f32 mode, one launch: dW += dY^T X (+ db += column sums of dY) over a list of problems.
// Replaces, for f32 operands, the per-Linear autograd products of fairseq/modules/transformer_layer.py:123-136,243-377 (nn.Linear
// backward: grad_weight = grad_output^T input, grad_bias = column sums) that the bf16 path hands to wgrad_group.hip.
//
// f32 mode is the parity mode of this build (exact-f32 MFMA, v_mfma_f32_16x16x4_f32: the f32 VECTOR rate, 1/16 of bf16), so a product
// is bound by its MFMA time and the only question is whether every CU has some: the per-product split-K launches of round 5 left
// half the chip idle on the s preset's 256-wide outputs (and summed their slices with f32 atomics, so two runs differed in the last
// bits).  Here every 128 x 128 tile of every dW is one work item owned by ONE workgroup over all its tokens -- no atomics, a fixed
// summation order -- and the items of all products are dealt to 2 workgroups per CU, longest reductions first.  One exception, as in
// wgrad_group.hip: when the longest class of tiles does not fill its last round of 512 workgroups, the tiles of that round are cut along
// the token range into as many pieces as fill it, and those pieces meet in f32 atomics (1,060 equal tiles would otherwise take three
// rounds on 512 workgroups where 2.07 are needed).
// Operands are read as they lie ([tokens][columns], K-major for both: the f32 MFMA takes one value per lane, A[row = lane & 15][k = lane >> 4],
// so dY^T needs no transposition, only rows of 16 consecutive columns): 32-token stages, register-staged (16-byte loads, next stage in
// flight during the MFMAs), LDS rows padded to 144 floats so that the four k-groups of a read fall on distinct banks.
#include "common.hpp"
#include "prof.hpp"
#include "s2t_hip.h"
#include <algorithm>
#include <cstring>
#include <vector>

namespace {
constexpr int TM = 128, TN = 128, BT = 32, SA = 144;            // tile, tokens per stage, LDS row stride (floats)
struct ProbF { const float *dY, *X; float *dW, *db; int n_out, n_in, tokens, ldy, ldx, ldw; };
struct ItemF { int prob, tm, tn, s0, s1, atomic; };              // stages [s0, s1) of 32 tokens; atomic: a piece of a tile cut along the tokens

__global__ __launch_bounds__(256, 2) void wgrad_f32_kernel(const ProbF* __restrict__ probs, const ItemF* __restrict__ items, int n_items) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [2 buffers][A | B][BT][SA]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const int pc = tid & 31, pr = tid >> 5;                      // this thread's 16-byte piece (columns 4 pc .. 4 pc + 3) of rows pr + 8 i
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        const ItemF I = items[it];
        const ProbF P = probs[I.prob];
        const int row0 = I.tm * TM, col0 = I.tn * TN;
        const bool okA = row0 + 4 * pc < P.n_out, okB = col0 + 4 * pc < P.n_in;          // pieces beyond the last column: zeros
        const float* gA = P.dY + row0 + 4 * pc;
        const float* gB = P.X + col0 + 4 * pc;
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 bsum = {0.f, 0.f, 0.f, 0.f};                        // column sums of dY (items with tn == 0 only)
        f32x4 ra[4], rb[4];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        auto fetch = [&](int k0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = k0 + pr + 8 * i;
                const bool in = k < P.tokens;
                ra[i] = (in && okA) ? *reinterpret_cast<const f32x4*>(gA + (size_t)k * P.ldy) : zero;
                rb[i] = (in && okB) ? *reinterpret_cast<const f32x4*>(gB + (size_t)k * P.ldx) : zero;
            }
        };
        auto stash = [&](int buf) {
            float* sA = lds + buf * 2 * BT * SA;
            float* sB = sA + BT * SA;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x4*>(sA + (pr + 8 * i) * SA + 4 * pc) = ra[i];
                *reinterpret_cast<f32x4*>(sB + (pr + 8 * i) * SA + 4 * pc) = rb[i];
                bsum += ra[i];
            }
        };
        const int nst = I.s1 - I.s0, k_base = I.s0 * BT;
        fetch(k_base);
        __syncthreads();                                          // the previous item's last reads of buffer 0
        stash(0);
        __syncthreads();
        for (int s = 0; s < nst; ++s) {
            if (s + 1 < nst) fetch(k_base + (s + 1) * BT);
            const float* sA = lds + (s & 1) * 2 * BT * SA + (lane >> 4) * SA + wm * 64 + (lane & 15);
            const float* sB = lds + (s & 1) * 2 * BT * SA + BT * SA + (lane >> 4) * SA + wn * 64 + (lane & 15);
#pragma unroll
            for (int k4 = 0; k4 < BT / 4; ++k4) {
                float a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { a[i] = sA[k4 * 4 * SA + 16 * i]; b[i] = sB[k4 * 4 * SA + 16 * i]; }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (s + 1 < nst) stash((s + 1) & 1);                  // buffer (s+1)&1 was last read in iteration s-1: everyone passed the barrier below
            __syncthreads();
        }
        // ---- epilogue: dW[row][col] += acc (this workgroup is the tile's only writer)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + wm * 64 + 16 * i + 4 * (lane >> 4) + r;
                if (row < P.n_out) {
                    float* dst = P.dW + (size_t)row * P.ldw + col0 + wn * 64 + (lane & 15);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (col0 + wn * 64 + 16 * j + (lane & 15) < P.n_in) {
                            if (I.atomic) atomicAdd(dst + 16 * j, acc[i][j][r]);
                            else dst[16 * j] += acc[i][j][r];
                        }
                    }
                }
            }
        }
        if (P.db && I.tn == 0) {                                  // column sums: 8 row groups per piece meet in LDS (buffers free: last barrier passed)
            float* red = lds;
            *reinterpret_cast<f32x4*>(red + (size_t)tid * 4) = bsum;
            __syncthreads();
            if (tid < TM) {
                float v = 0.f;
#pragma unroll
                for (int g = 0; g < 8; ++g) v += red[(g * 32 + (tid >> 2)) * 4 + (tid & 3)];
                if (row0 + tid < P.n_out) { if (I.atomic) atomicAdd(P.db + row0 + tid, v); else P.db[row0 + tid] += v; }
            }
            __syncthreads();
        }
    }
}
}  // namespace

extern "C" int s2t_wgrad_group_f32(int n, const S2TWgradProblem* probs, void* stream) {
    if (n == 0) return S2T_OK;
    if (n < 0 || !probs) return S2T_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    std::vector<ProbF> pv(n);
    std::vector<ItemF> iv;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        const S2TWgradProblem& s = probs[i];
        if (!s.dY || !s.X || !s.dW || s.n_out <= 0 || s.n_in <= 0 || s.tokens <= 0) return S2T_EINVAL;
        // 16-byte pieces of four columns: aligned rows whose padding covers the last piece
        if ((s.ldy & 3) || (s.ldx & 3) || ((uintptr_t)s.dY & 15) || ((uintptr_t)s.X & 15)) return S2T_EINVAL;
        if (((s.n_out + 3) & ~3) > s.ldy || ((s.n_in + 3) & ~3) > s.ldx) return S2T_EINVAL;
        ProbF& p = pv[i];
        p.dY = (const float*)s.dY; p.X = (const float*)s.X; p.dW = s.dW; p.db = s.db;
        p.n_out = s.n_out; p.n_in = s.n_in; p.tokens = s.tokens; p.ldy = s.ldy; p.ldx = s.ldx; p.ldw = s.ldw;
        for (int a = 0; a < (s.n_out + TM - 1) / TM; ++a)
            for (int b = 0; b < (s.n_in + TN - 1) / TN; ++b) iv.push_back(ItemF{i, a, b, 0, (s.tokens + BT - 1) / BT, 0});
        flops += 2.0 * s.n_out * (double)s.n_in * s.tokens;
        bytes += 4.0 * s.tokens * ((double)s.n_out + s.n_in) + 8.0 * s.n_out * (double)s.n_in;
    }
    // longest reductions first, dealt round-robin to the workgroups: a launch takes as long as its most loaded workgroup
    constexpr int SLOTS = 512;
    auto len = [](const ItemF& t) { return t.s1 - t.s0; };
    std::stable_sort(iv.begin(), iv.end(), [&](const ItemF& x, const ItemF& y) { return len(x) > len(y); });
    {   // the long class = items at least half as long as the longest; its partly filled last round is cut to fill the round
        size_t L = 0;
        while (L < iv.size() && 2 * len(iv[L]) >= len(iv[0])) ++L;
        const size_t rem = L % SLOTS;
        if (L > SLOTS && rem > 0 && rem <= SLOTS / 2 && len(iv[0]) >= 16) {
            const int f = (int)std::min<size_t>(8, SLOTS / rem);
            std::vector<ItemF> cut;
            for (size_t i = L - rem; i < L; ++i) {
                const int n = len(iv[i]), per = (n + f - 1) / f;
                for (int c = 0; c < n; c += per) cut.push_back(ItemF{iv[i].prob, iv[i].tm, iv[i].tn, c, std::min(n, c + per), 1});
            }
            iv.erase(iv.begin() + (L - rem), iv.begin() + L);
            iv.insert(iv.end(), cut.begin(), cut.end());
            std::stable_sort(iv.begin(), iv.end(), [&](const ItemF& x, const ItemF& y) { return len(x) > len(y); });
        }
    }
    const size_t pb = pv.size() * sizeof(ProbF), ib = iv.size() * sizeof(ItemF);
    hipError_t se = hipSuccess;
    char* dev = (char*)s2t_scratch(S2T_SCRATCH_WGRAD_F32, st, pb + ib, &se);
    if (!dev) return S2T_EHIP(se);
    std::vector<char> host(pb + ib);
    memcpy(host.data(), pv.data(), pb);
    memcpy(host.data() + pb, iv.data(), ib);
    // stream-ordered upload from pageable memory (staged by the runtime before the call returns); the previous launch that read this
    // scratch was enqueued on the same stream
    se = hipMemcpyAsync(dev, host.data(), pb + ib, hipMemcpyHostToDevice, st);
    if (se != hipSuccess) return S2T_EHIP(se);
    const size_t lds = (size_t)2 * 2 * BT * SA * sizeof(float);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    const int grid = (int)std::min<size_t>(iv.size(), SLOTS);
    ProfScope prof("wgrad_group_f32", st, flops, bytes);
    hipLaunchKernelGGL(wgrad_f32_kernel, dim3(grid), dim3(256), lds, st, (const ProbF*)dev, (const ItemF*)(dev + pb), (int)iv.size());
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}
