// GEMM argument block and the shared LDS-staged epilogue (used by gemm.hip and gemm_v2.hip).
#pragma once
#include "common.hpp"

struct GemmArgs {
    const void* A; const void* B; void* C;
    const float* bias; const void* residual; const void* aux; void* aux_out;
    int M, N, K, lda, ldb, ldc, ldr, ldaux;
    int act, accumulate, splitk;
    float alpha;
    // optional row gather / scatter (implicit-GEMM convolution, subsample.hip):
    //  mapA (A direct):  A(row r, k) = Asrc[ mapA[(k/periodA)*M + r] ][ k % periodA ]   (-1 -> zeros)
    //  mapB (B stored [K][N]): B(k, :) = Bsrc[ mapB[k] ][:]                                 (-1 -> zeros)
    //  mapC: output row r is written to C row mapC[r]
    const int* mapA; int periodA; const int* mapB; const int* mapC;
    float p_drop; unsigned long long seed;   // dropout on the activated value, before the residual add
    float* rowsum;                           // optional (TA products): rowsum[m] += sum_k op(A)[m][k]  (bias gradient of a Linear)
};


// Finishing loop of the epilogue, specialised on the activation: every thread completes CW consecutive columns
// of one row per iteration (16 bytes of output) from the f32 tile parked in LDS.
template <typename TO, int BM, int BN, int ACT, int NTH>
__device__ __forceinline__ void gemm_finish(const GemmArgs& p, const char* smem, int row0, int col0) {
    TO* C = reinterpret_cast<TO*>(p.C);
    const TO* R = reinterpret_cast<const TO*>(p.residual);
    const TO* AUX = reinterpret_cast<const TO*>(p.aux);
    TO* AUXO = reinterpret_cast<TO*>(p.aux_out);
    constexpr int RS = BN * 4 + 16;
    constexpr int CW = 16 / (int)sizeof(TO);                  // output elements per 16-byte chunk
    constexpr int CPR = BN / CW;
    const uint32_t drop_th = (uint32_t)fminf(p.p_drop * 4294967296.f, 4294967295.f);
    const float drop_inv = 1.f / (1.f - p.p_drop);
    const bool vC = (p.ldc % CW == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
    const bool vR = !R || ((p.ldr % CW == 0) && ((reinterpret_cast<uintptr_t>(R) & 15) == 0));
    const bool vX = !AUX || ((p.ldaux % CW == 0) && ((reinterpret_cast<uintptr_t>(AUX) & 15) == 0));
    const bool vO = !AUXO || ((p.ldaux % CW == 0) && ((reinterpret_cast<uintptr_t>(AUXO) & 15) == 0));
    for (int c = threadIdx.x; c < BM * CPR; c += NTH) {
        const int lr = c / CPR, cc = c % CPR;
        const int row = row0 + lr, col = col0 + cc * CW;
        if (row >= p.M || col >= p.N) continue;
        const size_t orow = p.mapC ? (size_t)p.mapC[row] : (size_t)row;
        float v[CW];
#pragma unroll
        for (int k = 0; k < CW / 4; ++k) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(smem + lr * RS + (cc * CW + 4 * k) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * k + e] = a0[e];
        }
        const bool fast = (col + CW <= p.N) && vC && vR && vX && vO;
        TO rres[CW], raux[CW];
        if (fast) {
            if (R) *reinterpret_cast<u32x4*>(rres) = *reinterpret_cast<const u32x4*>(R + (size_t)row * p.ldr + col);
            if (AUX) *reinterpret_cast<u32x4*>(raux) = *reinterpret_cast<const u32x4*>(AUX + (size_t)row * p.ldaux + col);
        } else {
#pragma unroll
            for (int e = 0; e < CW; ++e) {
                const bool ok = col + e < p.N;
                rres[e] = (R && ok) ? R[(size_t)row * p.ldr + col + e] : from_f32<TO>(0.f);
                raux[e] = (AUX && ok) ? AUX[(size_t)row * p.ldaux + col + e] : from_f32<TO>(0.f);
            }
        }
        float bv[CW];
        if (p.bias && fast && ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0)) {
#pragma unroll
            for (int k = 0; k < CW / 4; ++k) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + col + 4 * k);
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[4 * k + e] = b4[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < CW; ++e) bv[e] = (p.bias && col + e < p.N) ? p.bias[col + e] : 0.f;
        }
        // one hash per element quad when rows start on a multiple of four elements
        const bool quad_ok = (p.N & 3) == 0;
        u32x2 dh[CW / 4];
        if (p.p_drop > 0.f && quad_ok) {
            const uint64_t base = ((uint64_t)orow * p.N + col) >> 2;        // element index in the OUTPUT tensor (scattered rows too)
#pragma unroll
            for (int k = 0; k < CW / 4; ++k) dh[k] = drop_hash4(p.seed, base + k);
        }
        TO pre[CW], o[CW];
#pragma unroll
        for (int e = 0; e < CW; ++e) {
            float x = v[e] + bv[e];
            if constexpr (ACT == ACT_RELU) x = fmaxf(x, 0.f);
            else if constexpr (ACT == ACT_GELU) { pre[e] = from_f32<TO>(x); x = gelu_f(x); }
            else if constexpr (ACT == ACT_RELU_BWD) x = (to_f32(raux[e]) > 0.f) ? x : 0.f;
            else if constexpr (ACT == ACT_GELU_BWD) x *= gelu_grad_f(to_f32(raux[e]));
            if (p.p_drop > 0.f) {
                bool keep;
                if (quad_ok) keep = drop_field(dh[e >> 2], e & 3) >= (drop_th >> 16);
                else keep = dropout_keep(p.seed, (uint64_t)orow * p.N + col + e, drop_th);
                x = keep ? x * drop_inv : 0.f;
            }
            if (R) x += to_f32(rres[e]);
            o[e] = from_f32<TO>(x);
        }
        TO* dst = C + orow * p.ldc + col;
        if (fast) {
            if (p.accumulate) {
                TO old[CW];
                *reinterpret_cast<u32x4*>(old) = *reinterpret_cast<const u32x4*>(dst);
#pragma unroll
                for (int e = 0; e < CW; ++e) o[e] = from_f32<TO>(to_f32(o[e]) + to_f32(old[e]));
            }
            *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(o);
            if (ACT == ACT_GELU && AUXO) *reinterpret_cast<u32x4*>(AUXO + (size_t)row * p.ldaux + col) = *reinterpret_cast<const u32x4*>(pre);
        } else {
#pragma unroll
            for (int e = 0; e < CW; ++e) {
                if (col + e >= p.N) continue;
                dst[e] = p.accumulate ? from_f32<TO>(to_f32(o[e]) + to_f32(dst[e])) : o[e];
                if (ACT == ACT_GELU && AUXO) AUXO[(size_t)row * p.ldaux + col + e] = pre[e];
            }
        }
    }
}

// Shared epilogue (see the comment inside): acc -> LDS -> 16-byte vector finish / split-K atomics.
template <typename TO, int BM, int BN, int MT, int NT, int NTH>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, const f32x4 (&acc)[MT][NT], char* smem,
                                              int row0, int col0, int wrow0, int wcol0, int q, int r16) {
    // ---------------- epilogue
    // Park the f32 accumulators in LDS (free after the main loop); then every thread finishes CW consecutive
    // columns of one row (16 bytes of output): bias / activation / dropout / residual on vectors, one 16-byte
    // store.  Split-K partial sums leave as f32 atomics with the lanes of a wave on 64 consecutive columns
    // (256 contiguous bytes per wave-instruction: the full-rate shape, MI355X_MICROARCH.md "Global float atomics").
    TO* C = reinterpret_cast<TO*>(p.C);
    const TO* R = reinterpret_cast<const TO*>(p.residual);
    const TO* AUX = reinterpret_cast<const TO*>(p.aux);
    TO* AUXO = reinterpret_cast<TO*>(p.aux_out);
    constexpr int RS = BN * 4 + 16;                           // LDS row stride (bytes); 528 B = 4 banks mod 32: b128 writes conflict-free
    // the MFMAs ran as (W-rows x X-rows): acc[i][j][r] = C[m = 16i + r16][n = 16j + 4q + r]  ->  one 16-byte LDS write each
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int lr = wrow0 + 16 * i + r16, lc = wcol0 + 16 * j + 4 * q;
            *reinterpret_cast<f32x4*>(smem + lr * RS + lc * 4) = acc[i][j] * p.alpha;
        }
    __syncthreads();
    if constexpr (sizeof(TO) == 4) {
        if (p.splitk > 1) {
            for (int idx = threadIdx.x; idx < BM * BN; idx += NTH) {
                const int lr = idx / BN, lc = idx % BN;
                const int row = row0 + lr, col = col0 + lc;
                if (row >= p.M || col >= p.N) continue;
                const size_t orow = p.mapC ? (size_t)p.mapC[row] : (size_t)row;
                atomicAdd(reinterpret_cast<float*>(C) + orow * p.ldc + col, *reinterpret_cast<const float*>(smem + lr * RS + lc * 4));
            }
            return;
        }
    }
    switch (p.act) {          // the activation is compile-time inside the finishing loop (27 -> ~6 VALU per element)
        case ACT_RELU: gemm_finish<TO, BM, BN, ACT_RELU, NTH>(p, smem, row0, col0); break;
        case ACT_GELU: gemm_finish<TO, BM, BN, ACT_GELU, NTH>(p, smem, row0, col0); break;
        case ACT_RELU_BWD: gemm_finish<TO, BM, BN, ACT_RELU_BWD, NTH>(p, smem, row0, col0); break;
        case ACT_GELU_BWD: gemm_finish<TO, BM, BN, ACT_GELU_BWD, NTH>(p, smem, row0, col0); break;
        default: gemm_finish<TO, BM, BN, ACT_NONE, NTH>(p, smem, row0, col0); break;
    }
}

