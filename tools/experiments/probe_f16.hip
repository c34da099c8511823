// Probe (round 6): how gfx950 treats fp16 subnormals in the places precision 2 depends on --
//   (1) the compiler's fp32 -> fp16 fptrunc (v_cvt_pk_f16_f32 / v_cvt_f16_f32), (2) v_mfma_f32_16x16x32_f16 / 32x32x16_f16 operands,
//   (3) v_dot2_f32_f16 operands, (4) saturation of the convert.  Build: hipcc --offload-arch=gfx950 -O3 probe_f16.hip -o probe_f16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void probe(float* out, float tiny, float big) {
    const int lane = threadIdx.x;
    // (1) conversion of a value below the fp16 normal range (6.1e-5)
    const f2 v = {tiny, tiny * 3.f};
    const h2 hv = __builtin_convertvector(v, h2);
    if (lane == 0) { out[0] = (float)hv[0]; out[1] = (float)hv[1]; }
    // (2) MFMA with subnormal A, B = 1: c[i][j] = sum_k a[i][k] b[k][j] = 32 * tiny
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = hv[0]; b[i] = (_Float16)1.0f; }
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (lane == 0) out[2] = c[0];
    f16v c2;
    for (int i = 0; i < 16; ++i) c2[i] = 0.f;
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    if (lane == 0) out[3] = c2[0];
    // subnormal x subnormal-free: B subnormal instead
    c = (f4){0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c, 0, 0, 0);
    if (lane == 0) out[4] = c[0];
    // (3) dot2
    const h2 w = {hv[0], hv[0]}, one = {(_Float16)1.0f, (_Float16)1.0f};
    if (lane == 0) out[5] = __builtin_amdgcn_fdot2(w, one, 0.f, false);
    // (4) out-of-range convert
    const f2 bv = {big, -big};
    const h2 hb = __builtin_convertvector(bv, h2);
    if (lane == 0) { out[6] = (float)hb[0]; out[7] = (float)hb[1]; }
}
int main() {
    float* d; hipMalloc(&d, 64);
    probe<<<1, 64>>>(d, 1.0e-6f, 1.0e5f);
    float h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("cvt(1e-6) = %.9g  cvt(3e-6) = %.9g   (fp16 subnormal spacing 5.96e-8; 0 = flushed)\n", h[0], h[1]);
    printf("mfma16x16x32_f16 A subnormal: %.9g (expect 32 x cvt = %.9g)\n", h[2], 32 * h[0]);
    printf("mfma32x32x16_f16 A subnormal: %.9g (expect 16 x cvt = %.9g)\n", h[3], 16 * h[0]);
    printf("mfma16x16x32_f16 B subnormal: %.9g\n", h[4]);
    printf("dot2_f32_f16 subnormal: %.9g (expect 2 x cvt = %.9g)\n", h[5], 2 * h[0]);
    printf("cvt(1e5) = %g cvt(-1e5) = %g\n", h[6], h[7]);
    return 0;
}
