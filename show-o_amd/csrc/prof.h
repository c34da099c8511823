// Optional per-launch HIP-event timing of the hot kernels (used by bench.py's roofline leg).
#pragma once
#include <hip/hip_runtime.h>
namespace showo {
enum { PROF_GEMM = 0, PROF_ATTN = 1, PROF_CONV = 2, PROF_KINDS = 3 };
struct ProfScope {
    int idx;
    hipStream_t s;
    ProfScope(int kind, double work, hipStream_t stream);
    ~ProfScope();
};
}  // namespace showo
