// Library info, event-based per-family timing and the host-side CTC unit-error-rate helper.
#include "common.hpp"
#include "prof.hpp"
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>
#include <cstring>

int g_s2t_prof_on = 0;
int g_s2t_opt_gemm256 = 1, g_s2t_opt_attn_v1 = 0, g_s2t_opt_attn_v2_min_tq = 16, g_s2t_opt_gemm256_min_tiles = 0, g_s2t_opt_gemm256_sched = 0;
int g_s2t_opt_ln_small = 1;                                 // 1: the LayerNorm backward of small activations requests a wave's rows four at a time (norm_optim.hip)
int g_s2t_opt_gemm_deep = 1;                                // 1: the 64 x 64 bf16 products request DEPTH k-tiles before the first MFMA (gemm.hip)
int g_s2t_opt_attn_bwd_fused = 0;                          // 1: the one-kernel attention backward for Tk <= 384 (attention.hip)
int g_s2t_opt_small_nt = 192, g_s2t_opt_small_kt = 40;      // bf16 products: below that many 128 x 128 tiles the 64 x 64 form (NT / NN and TN)
int g_s2t_opt_f32_small_nt = 1024, g_s2t_opt_f32_small_kt = 512, g_s2t_opt_f32_narrow = 0;   // tile-shape thresholds of f32 products (gemm.hip)
int g_s2t_opt_reserve_cus = 0;            // CUs the persistent one-workgroup-per-CU kernels (gemm256, wgrad_group) leave to a concurrent collective
extern int g_s2t_opt_decode_stop_after;
namespace {
struct Rec { hipEvent_t a, b; double flops, bytes; };
struct Fam { std::vector<Rec> recs; double ms = 0, flops = 0, bytes = 0; long long launches = 0; };
std::map<std::string, Fam> g_fams;
std::vector<hipEvent_t> g_pool;
std::mutex g_mu;
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
void drain(Fam& f) {
    for (auto& r : f.recs) {
        (void)hipEventSynchronize(r.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { f.ms += ms; f.flops += r.flops; f.bytes += r.bytes; f.launches += 1; }
        g_pool.push_back(r.a); g_pool.push_back(r.b);
    }
    f.recs.clear();
}
}  // namespace

void s2t_prof_push(const char* family, hipStream_t st, double flops, double bytes, bool begin) {
    std::lock_guard<std::mutex> lk(g_mu);
    Fam& f = g_fams[family];
    if (begin) {
        Rec r{get_event(), get_event(), flops, bytes};
        (void)hipEventRecord(r.a, st);
        f.recs.push_back(r);
    } else if (!f.recs.empty()) {
        (void)hipEventRecord(f.recs.back().b, st);
    }
}

void* s2t_scratch(int slot, hipStream_t st, size_t bytes, hipError_t* err) {
    struct Buf { void* p = nullptr; size_t cap = 0; };
    static std::map<std::tuple<int, int, hipStream_t>, Buf> bufs;
    static std::mutex mu;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    Buf& b = bufs[std::make_tuple(slot, dev, st)];
    if (b.cap < bytes) {
        if (b.p) (void)hipFree(b.p);                   // waits for the kernels that may still read it
        b.p = nullptr; b.cap = 0;
        hipError_t e = hipMalloc(&b.p, bytes);
        if (e != hipSuccess) { b.p = nullptr; if (err) *err = e; return nullptr; }
        b.cap = bytes;
    }
    return b.p;
}

// ---------------------------------------------------------------- stand-in for a collective (bench.py data_parallel.dry_run)
// What an RCCL all-reduce does to the kernels it runs beside, without a second GPU: `workgroups` workgroups stay resident for the time a
// ring all-reduce of `bytes` over `ranks` ranks takes at `bus_gbps` (2 (ranks - 1) / ranks * bytes / bus), and move 2 x bytes through HBM in
// each direction meanwhile, evenly paced (each 64 KiB chunk waits for its slot on the 100 MHz real-time counter).  It measures the cost of
// SHARING CUs and HBM with a collective; it moves no data between GPUs and proves nothing about xGMI.
__global__ __launch_bounds__(256) void comm_standin_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, int passes,
                                                           unsigned long long ticks_total) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
    constexpr size_t CH = 4096;                                   // 16-byte pieces per chunk
    const unsigned long long nchunks = (unsigned long long)passes * ((hi > lo ? hi - lo : 0) + CH - 1) / CH;
    unsigned long long idx = 0;
    for (int p = 0; p < passes; ++p) {
        for (size_t c0 = lo; c0 < hi; c0 += CH, ++idx) {
            const unsigned long long due = nchunks ? idx * ticks_total / nchunks : 0;
            while (__builtin_amdgcn_s_memrealtime() - t0 < due) __builtin_amdgcn_s_sleep(32);
            const size_t c1 = c0 + CH < hi ? c0 + CH : hi;
            for (size_t i = c0 + threadIdx.x; i < c1; i += 256) dst[i] = src[i];
        }
    }
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks_total) __builtin_amdgcn_s_sleep(32);
}
extern "C" int s2t_comm_standin(const void* src, void* dst, size_t bytes, int workgroups, int ranks, float bus_gbps, void* stream) {
    if (!src || !dst || workgroups < 1 || workgroups > 256 || ranks < 2 || !(bus_gbps > 0.f) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return S2T_EINVAL;
    if (bytes < 16) return S2T_OK;
    const double seconds = 2.0 * (ranks - 1) / ranks * (double)bytes / ((double)bus_gbps * 1e9);
    hipLaunchKernelGGL(comm_standin_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst, bytes / 16, 2,
                       (unsigned long long)(seconds * 1e8));
    S2T_LAUNCH_CHECK();
    return S2T_OK;
}

extern "C" int s2t_abi_version(void) { return 9; }
extern "C" const char* s2t_build_info(void) { return "libs2t_hip gfx950 (CDNA4, wave64, MFMA) built " __DATE__ " " __TIME__; }
extern "C" int s2t_set_option(const char* key, int value) {
    if (!key) return S2T_EINVAL;
    int* slot = !strcmp(key, "gemm256") ? &g_s2t_opt_gemm256 : !strcmp(key, "attn_v1") ? &g_s2t_opt_attn_v1
              : !strcmp(key, "attn_v2_min_tq") ? &g_s2t_opt_attn_v2_min_tq : !strcmp(key, "gemm256_min_tiles") ? &g_s2t_opt_gemm256_min_tiles
              : !strcmp(key, "gemm256_sched") ? &g_s2t_opt_gemm256_sched : !strcmp(key, "decode_stop_after") ? &g_s2t_opt_decode_stop_after
              : !strcmp(key, "reserve_cus") ? &g_s2t_opt_reserve_cus : !strcmp(key, "gemm_f32_small_nt") ? &g_s2t_opt_f32_small_nt
              : !strcmp(key, "gemm_f32_small_kt") ? &g_s2t_opt_f32_small_kt : !strcmp(key, "gemm_f32_narrow") ? &g_s2t_opt_f32_narrow
              : !strcmp(key, "gemm_small_nt") ? &g_s2t_opt_small_nt : !strcmp(key, "gemm_small_kt") ? &g_s2t_opt_small_kt
              : !strcmp(key, "attn_bwd_fused") ? &g_s2t_opt_attn_bwd_fused : !strcmp(key, "gemm_deep") ? &g_s2t_opt_gemm_deep
              : !strcmp(key, "ln_small") ? &g_s2t_opt_ln_small : nullptr;
    if (slot == &g_s2t_opt_reserve_cus && (value < 0 || value > 128)) return S2T_EINVAL;
    if (!slot) return S2T_EINVAL;
    const int old = *slot;
    *slot = value;
    return old;
}
extern "C" int s2t_prof_enable(int on) { g_s2t_prof_on = on ? 1 : 0; return S2T_OK; }
extern "C" int s2t_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_fams) { drain(kv.second); kv.second = Fam(); }
    return S2T_OK;
}
extern "C" int s2t_prof_read(const char* family, double* ms, long long* launches, double* flops, double* bytes) {
    if (!family) return S2T_EINVAL;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_fams.find(family);
    if (it == g_fams.end()) { if (ms) *ms = 0; if (launches) *launches = 0; if (flops) *flops = 0; if (bytes) *bytes = 0; return S2T_OK; }
    drain(it->second);
    if (ms) *ms = it->second.ms;
    if (launches) *launches = it->second.launches;
    if (flops) *flops = it->second.flops;
    if (bytes) *bytes = it->second.bytes;
    return S2T_OK;
}

// ---------------------------------------------------------------- host: CTC unit error rate
// Greedy path (collapse repeats, drop blanks) then the alignment of wer_utils.EditDistance(False):
// substitution 4, insertion/deletion 3, diagonal preferred on ties, then insertion, then deletion;
// errors = number of non-match steps on the back-trace.  Directions are kept in a byte matrix.
static int align_errors(const std::vector<int>& hyp /*decoded prediction = "refs"*/, const long long* tgt, int nt) {
    const int nr = (int)hyp.size();
    if (nr == 0 && nt == 0) return 0;
    const int cols = nt + 1;
    std::vector<double> prev(cols), cur(cols);
    std::vector<unsigned char> dir((size_t)(nr + 1) * cols, 0);   // 0 diag, 1 left (insertion), 2 up (deletion)
    for (int j = 0; j < cols; ++j) { prev[j] = 3.0 * j; dir[j] = 1; }
    for (int i = 1; i <= nr; ++i) {
        cur[0] = 3.0 * i; dir[(size_t)i * cols] = 2;
        for (int j = 1; j < cols; ++j) {
            double best = prev[j - 1] + (hyp[i - 1] == (int)tgt[j - 1] ? 0.0 : 4.0);
            unsigned char d = 0;
            const double ins = cur[j - 1] + 3.0;
            if (ins < best) { best = ins; d = 1; }
            const double del = prev[j] + 3.0;
            if (del < best) { best = del; d = 2; }
            cur[j] = best; dir[(size_t)i * cols + j] = d;
        }
        std::swap(prev, cur);
    }
    int errors = 0, i = nr, j = nt;
    while (i > 0 || j > 0) {
        const unsigned char d = dir[(size_t)i * cols + j];
        if (d == 0) { if (hyp[i - 1] != (int)tgt[j - 1]) ++errors; --i; --j; }
        else if (d == 1) { ++errors; --j; }
        else { ++errors; --i; }
    }
    return errors;
}

extern "C" int s2t_host_ctc_uer(const int* pred, const long long* input_len, int B, int T, const long long* targets,
                                const long long* target_len, int L, int blank, double* errors, double* total) {
    if (!pred || !input_len || !targets || !target_len || !errors || !total) return S2T_EINVAL;
    double e = 0, n = 0;
    std::vector<int> dec;
    for (int b = 0; b < B; ++b) {
        dec.clear();
        const int* p = pred + (size_t)b * T;
        const int len = (int)(input_len[b] < T ? input_len[b] : T);
        for (int t = 0; t < len; ++t) {
            if (t > 0 && p[t] == p[t - 1]) continue;
            if (p[t] != blank) dec.push_back(p[t]);
        }
        const int tl = (int)target_len[b];
        e += align_errors(dec, targets + (size_t)b * L, tl);
        n += tl;
    }
    *errors = e; *total = n;
    return S2T_OK;
}


// Frame-budget batching (host): fairseq/data/data_utils_fast.pyx:16-68 batch_by_size_fast.  `indices` is the ordered list of
// dataset indices, lens[idx] the number of frames of utterance idx.  A batch is closed when adding the next utterance would make
// (batch size + 1) x (longest member) exceed max_tokens, or the batch holds max_sentences; closed batches are trimmed to a
// multiple of bsz_mult (the remainder opens the next batch).  Output: the batches concatenated in out_flat[n], their bounds in
// out_offsets[*n_batches + 1].  max_tokens / max_sentences <= 0 disable the respective limit.
extern "C" int s2t_host_batch_by_size(const long long* indices, long long n, const long long* lens, long long max_tokens,
                                      long long max_sentences, int bsz_mult, long long* out_flat, long long* out_offsets,
                                      long long* n_batches) {
    if (n < 0 || bsz_mult < 1 || !out_offsets || !n_batches || (n > 0 && (!indices || !lens || !out_flat))) return S2T_EINVAL;
    std::vector<long long> batch, blen;
    long long nb = 0, written = 0, sample_len = 0;
    out_offsets[0] = 0;
    for (long long i = 0; i < n; ++i) {
        const long long idx = indices[i], nt = lens[idx];
        blen.push_back(nt);
        sample_len = sample_len > nt ? sample_len : nt;
        if (max_tokens > 0 && sample_len > max_tokens) return S2T_EINVAL;          // "sentence ... exceeds max_tokens limit"
        const long long num_tokens = ((long long)batch.size() + 1) * sample_len;
        const bool full = !batch.empty() && ((max_sentences > 0 && (long long)batch.size() == max_sentences) ||
                                             (max_tokens > 0 && num_tokens > max_tokens));
        if (full) {
            const long long sz = (long long)batch.size();
            const long long a = bsz_mult * (sz / bsz_mult), b = sz % bsz_mult, mod_len = a > b ? a : b;
            for (long long k = 0; k < mod_len; ++k) out_flat[written++] = batch[k];
            out_offsets[++nb] = written;
            batch.erase(batch.begin(), batch.begin() + mod_len);
            blen.erase(blen.begin(), blen.begin() + mod_len);
            sample_len = 0;
            for (long long v : blen) sample_len = sample_len > v ? sample_len : v;
        }
        batch.push_back(idx);
    }
    if (!batch.empty()) {
        for (long long v : batch) out_flat[written++] = v;
        out_offsets[++nb] = written;
    }
    *n_batches = nb;
    return S2T_OK;
}
