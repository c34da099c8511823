// Probe (round 6): does MODE.FP16_OVFL (bit 23 of HW_REG_MODE) make gfx950's fp32 -> fp16 converts saturate at +-65504 (keeping true
// inf / NaN) -- i.e. can the v_med3_f32 clamp in front of every fp16 convert of precision 2 (common.h sat_f16) be dropped?
// Build: hipcc --offload-arch=gfx950 -O3 probe_ovfl.hip -o probe_ovfl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void probe(float* out, float big, float inf, float nan, int set) {
    if (set) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1" ::: "memory");
    const f2 a = {big, -big}, b = {inf, nan}, c = {65519.f, 65520.f};
    const h2 ha = __builtin_convertvector(a, h2), hb = __builtin_convertvector(b, h2), hc = __builtin_convertvector(c, h2);
    const _Float16 s = (_Float16)big;
    if (threadIdx.x == 0) {
        out[0] = (float)ha[0]; out[1] = (float)ha[1]; out[2] = (float)hb[0]; out[3] = (float)hb[1];
        out[4] = (float)hc[0]; out[5] = (float)hc[1]; out[6] = (float)s;
    }
}
int main() {
    float* d; hipMalloc(&d, 64);
    for (int set = 0; set < 2; ++set) {
        probe<<<1, 64>>>(d, 1.0e6f, INFINITY, NAN, set);
        float h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d: pk(1e6, -1e6) = %g %g | pk(inf, nan) = %g %g | pk(65519, 65520) = %g %g | scalar(1e6) = %g\n", set, h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
    }
    return 0;
}
