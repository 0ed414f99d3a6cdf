// Optional HIP-event bracketing of kernel families (used by bench.py for the roofline figure).
#pragma once
#include <hip/hip_runtime.h>
void s2t_prof_push(const char* family, hipStream_t st, double flops, double bytes, bool begin);
extern int g_s2t_prof_on;
struct ProfScope {
    const char* fam; hipStream_t st; double flops, bytes;
    ProfScope(const char* f, hipStream_t s, double fl, double by) : fam(f), st(s), flops(fl), bytes(by) {
        if (g_s2t_prof_on) s2t_prof_push(fam, st, flops, bytes, true);
    }
    ~ProfScope() { if (g_s2t_prof_on) s2t_prof_push(fam, st, flops, bytes, false); }
};

// Per-(device, stream) scratch of the entry points that hand workgroup partial sums to a finishing kernel: two calls of one entry
// point on different streams (or devices) get different buffers; calls on ONE stream are ordered by the stream.  Grows, never
// shrinks; lives until process exit.  Returns nullptr and sets *err when the allocation fails.
enum { S2T_SCRATCH_LSCE = 0, S2T_SCRATCH_KD, S2T_SCRATCH_GNORM, S2T_SCRATCH_CONV1_BWD, S2T_SCRATCH_WGRAD_F32, S2T_SCRATCH_SLOTS };
void* s2t_scratch(int slot, hipStream_t st, size_t bytes, hipError_t* err);
