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
