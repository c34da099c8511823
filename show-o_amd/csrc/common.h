// Shared device/host helpers for the gfx950 (MI355X, CDNA4) Show-o hot path.
// Wave = 64 lanes everywhere in this tree; nothing here is portable to 32-wide hardware by design.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace showo {

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__host__ __device__ inline float bf2f(bf16_t v) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)v) << 16;
    return c.f;
}

// round-to-nearest-even.  Device code uses the gfx950 conversion instruction (v_cvt_pk_bf16_f32: one instruction per PAIR instead of
// ~7 VALU instructions per value for the integer form; bit-identical on every finite value and on infinities, NaN stays NaN with
// the hardware's quiet pattern) -- the integer form was 25-45 % of the VALU work of every bf16-producing GEMM epilogue
// (profiles/r2_gemm_harness.txt, r3e).  The host keeps the integer form (NaN preserved as quiet NaN).
// The conversion is written as the COMPILER's fptrunc (selected to v_cvt_pk_bf16_f32 on gfx950), not as inline assembly: converted
// values feed MFMA operands in the attention kernels, and the hazard recognizer cannot see a VALU write inside an asm block (round 5:
// an asm-written P fragment consumed by the next-but-one MFMA gave bf16-level errors in one variant of the split attention).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__host__ __device__ inline bf16_t f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(SHOWO_PACK_ASM)  // A/B build of the round-4 form (scripts/gpu_r5_l.sh); never the shipped library
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(r) : "v"(f));
    return (bf16_t)(r & 0xffffu);
#elif defined(__HIP_DEVICE_COMPILE__)
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
#else
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
#endif
}

__device__ inline uint32_t pack_bf2(float lo, float hi) {
#if defined(SHOWO_PACK_ASM)
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
#else
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
#endif
}

// ---- the 16-bit OPERAND TYPE of the MFMA / dot2 kernels, as a compile-time property of a kernel instance (round 6) ----------------
// Op16<false> = bfloat16: the operands of precision 0 (the default) and the (hi, lo) halves of precision 1.
// Op16<true>  = IEEE binary16 ("fp16", Showo.set_precision(2)): the same instructions at the same rate and peak
//               (v_mfma_f32_*_f16 / v_dot2_f32_f16, 2.5 PFLOP/s dense) with 11 instead of 8 significand bits -- operand rounding 2^-12
//               instead of 2^-9, which puts the logits within north_star's 1e-3 of the fp32 reference (oracle/predict_rounding.py)
//               where bf16 operands sit at 7e-3.  Range 6.1e-5 .. 65504: every convert SATURATES (|x| > 65504 -> +-65504), values below the normal range keep their subnormal encoding (conversions and MFMA operands are not flushed on
//               gfx950: tools/experiments/probe_f16.hip), and showo_engine_set_range_check counts saturated elements per layer.
// Buffers stay raw uint16_t (bf16_t) images either way: only the instructions that PRODUCE (pack2 / cvt), CONSUME (mfma16 / mfma32 /
// dot2) or WIDEN (tof) an element depend on the type; DMA, LDS layouts, swizzles and stores are type-blind.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
constexpr float F16_MAX = 65504.0f;
// one v_med3_f32 per value.  (A NaN comes out as -65504 -- v_med3 returns the minimum when an input is NaN; the fp32 residual stream and
// the fp32 GEMM accumulators are never converted, so a NaN that reaches x still reaches the logits; the range check counts |x| = 65504.)
#if defined(SHOWO_F16_NOSAT)  // A/B build only (what the saturation costs); never the shipped library
__device__ inline float sat_f16(float x) { return x; }
#else
__device__ inline float sat_f16(float x) { return __builtin_amdgcn_fmed3f(x, -F16_MAX, F16_MAX); }
#endif
template <bool F16>
struct Op16;
template <>
struct Op16<false> {
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf2(lo, hi); }
    static __device__ __forceinline__ uint32_t pack2_bounded(float lo, float hi) { return pack_bf2(lo, hi); }
    static __device__ __forceinline__ bf16_t cvt(float f) { return f2bf(f); }
    static __device__ __forceinline__ float tof(bf16_t v) { return bf2f(v); }
    static __device__ __forceinline__ float lo_of(uint32_t pk) { return __uint_as_float(pk << 16); }          // element 0 / 1 of a packed pair
    static __device__ __forceinline__ float hi_of(uint32_t pk) { return __uint_as_float(pk & 0xffff0000u); }
    static __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ float dot2(uint32_t w, uint32_t a, float acc) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), __builtin_bit_cast(bf16x2_t, a), acc, false);
    }
};
template <>
struct Op16<true> {
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        const f32x2_t v = {sat_f16(lo), sat_f16(hi)};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));  // the compiler's own fptrunc (RNE), visible to the hazard recognizer
    }
    // values bounded by construction (soft-max numerators <= e^8, convex combinations of already converted values): no clamp --
    // in the attention kernels the clamp of P cost 5 % of the kernel (profiles/r6_f16_saturation_ab.txt)
    static __device__ __forceinline__ uint32_t pack2_bounded(float lo, float hi) {
        const f32x2_t v = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
    }
    static __device__ __forceinline__ bf16_t cvt(float f) { return __builtin_bit_cast(bf16_t, (_Float16)sat_f16(f)); }
    static __device__ __forceinline__ float tof(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
    static __device__ __forceinline__ float lo_of(uint32_t pk) { return (float)__builtin_bit_cast(f16x2_t, pk)[0]; }
    static __device__ __forceinline__ float hi_of(uint32_t pk) { return (float)__builtin_bit_cast(f16x2_t, pk)[1]; }
    static __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float dot2(uint32_t w, uint32_t a, float acc) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, w), __builtin_bit_cast(f16x2_t, a), acc, false);
    }
};
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// DPP lane moves (VALU only, no trip through the LDS crossbar).  On gfx9 a 16-lane row offers rotations and quad permutes, not a
// general xor: partner 8 away IS the rotation by 8; partner 4 away is the rotation by 4 once the values have period 8 within the row
// (true after the step-8 addition of a butterfly reduction: v[i] == v[i ^ 8]); partners 2 and 1 are quad permutes.
template <int CTRL>
__device__ inline float dpp_move(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_ROR8 = 0x128, DPP_ROR4 = 0x124, DPP_XOR2 = 0x4E, DPP_XOR1 = 0xB1;  // row_ror:8, row_ror:4, quad_perm [2,3,0,1], [1,0,3,2]
constexpr int DPP_HALF_MIRROR = 0x141;  // lane i of every 8 <- lane 7 - i
// the last four steps (partners 8, 4, 2, 1) of a butterfly sum / max on DPP: the partner VALUES of __shfl_xor, hence the same bits
__device__ inline float row_sum_dpp(float v) {
    v += dpp_move<DPP_ROR8>(v);
    v += dpp_move<DPP_ROR4>(v);
    v += dpp_move<DPP_XOR2>(v);
    v += dpp_move<DPP_XOR1>(v);
    return v;
}
__device__ inline float row_max_dpp(float v) {
    v = fmaxf(v, dpp_move<DPP_ROR8>(v));
    v = fmaxf(v, dpp_move<DPP_ROR4>(v));
    v = fmaxf(v, dpp_move<DPP_XOR2>(v));
    v = fmaxf(v, dpp_move<DPP_XOR1>(v));
    return v;
}
// wave_sum entirely on the VALU: partners 32 and 16 on gfx950's row-swap instructions, 8 .. 1 on DPP -- the same partner values in the
// same order of additions as wave_sum: identical bits (tests/test_kernels_gpu.py).  The decode GEMVs reduce once per weight row per
// wave and the single-query attention runs five dependent reductions: a ds_bpermute step costs an LDS round trip (~100 cycles) each.
__device__ inline float wave_sum_swap(float v) {
    const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
    const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
    return row_sum_dpp(v);
}
__device__ inline float wave_max_swap(float v) {
    const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
    const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
    return row_max_dpp(v);
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// gelu_new(x) = 0.5 x (1 + tanh(u)) = x * sigmoid(2u) = x / (1 + 2^(x (C1 + C3 x^2))),  u = sqrt(2/pi) (x + 0.044715 x^3),
// C1 = -2 log2(e) sqrt(2/pi), C3 = 0.044715 C1: 3 multiplies + 1 fma + 1 add around v_exp_f32 / v_rcp_f32 (round 6; the literal
// transcription of the formula cost 8 -- the fc1 half of the projection epilogue is VALU-bound on it: DESIGN.md).  The packed form
// evaluates two values per v_pk_*_f32 instruction and gives the SAME bits per element (IEEE fp32 operations either way).
constexpr float GELU_C1 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
constexpr float GELU_C3 = GELU_C1 * 0.044715f;
__device__ inline float gelu_new_fast(float x) {
    const float p = __builtin_fmaf(x * x, GELU_C3, GELU_C1);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
}
__device__ inline f32x2_t gelu_new_fast2(f32x2_t x) {
    const f32x2_t p = __builtin_elementwise_fma(x * x, (f32x2_t){GELU_C3, GELU_C3}, (f32x2_t){GELU_C1, GELU_C1});
    const f32x2_t t = x * p;
    const f32x2_t e = (f32x2_t){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + (f32x2_t){1.0f, 1.0f};
    return x * (f32x2_t){__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
}
__device__ __forceinline__ void gelu4(float (&v)[4]) {
    const f32x2_t a = gelu_new_fast2((f32x2_t){v[0], v[1]}), c = gelu_new_fast2((f32x2_t){v[2], v[3]});
    v[0] = a[0]; v[1] = a[1]; v[2] = c[0]; v[3] = c[1];
}
// v[r] = gelu_new(v[r] + b[r]), r = 0..3, on the packed forms (the bias add included: v_pk_add_f32)
__device__ __forceinline__ void gelu4_bias(float (&v)[4], const float (&b)[4]) {
    const f32x2_t a = gelu_new_fast2((f32x2_t){v[0], v[1]} + (f32x2_t){b[0], b[1]});
    const f32x2_t c = gelu_new_fast2((f32x2_t){v[2], v[3]} + (f32x2_t){b[2], b[3]});
    v[0] = a[0]; v[1] = a[1]; v[2] = c[0]; v[3] = c[1];
}

// acc + w[0] a[0] + w[1] a[1] on packed bf16 pairs: v_dot2c_f32_bf16 (gfx950), no bf16 -> fp32 conversions (the converting FMA form
// costs 3 VALU instructions per product; the decode GEMVs were VALU-bound next to their weight stream).  dot8_bf16 -- the four pairs of
// a 16-byte group in ascending order into ONE accumulator -- is THE accumulation order of every GEMV of the decode step (gemv_kernel,
// ln_gemv2 / out_gemv2, the batched kernels, the co-scheduled fc2 roles): they all give the same bits per output column.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ inline float dot2_bf16(uint32_t w, uint32_t a, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), __builtin_bit_cast(bf16x2_t, a), acc, false);
}
__device__ inline float dot8_bf16(const uint4& w, const uint4& a, float acc) {
    acc = dot2_bf16(w.x, a.x, acc);
    acc = dot2_bf16(w.y, a.y, acc);
    acc = dot2_bf16(w.z, a.z, acc);
    acc = dot2_bf16(w.w, a.w, acc);
    return acc;
}
// the same chain on either operand type (precision 2: v_dot2_f32_f16)
template <bool F16>
__device__ inline float dot8_op(const uint4& w, const uint4& a, float acc) {
    acc = Op16<F16>::dot2(w.x, a.x, acc);
    acc = Op16<F16>::dot2(w.y, a.y, acc);
    acc = Op16<F16>::dot2(w.z, a.z, acc);
    acc = Op16<F16>::dot2(w.w, a.w, acc);
    return acc;
}

// 16-byte streaming load with the non-temporal hint (global_load_dwordx4 ... nt): weights that ONE wave reads once (decode GEMVs)
// should not displace reusable lines; measured on MI355X decode layers: issued -> landed -18 % (MI355X_MICROARCH.md, "nt-weights")
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ inline uint4 ldg_nt16(const void* p) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// direct HBM -> LDS copy, 16 B per lane; the LDS destination is wave-uniform base + lane * 16 (lane-linear image)
__device__ inline void glds16(const bf16_t* src, bf16_t* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

// The same DMA issued from inline assembly: the compiler does not see an LDS write in flight.  With the builtin, its wait-count pass
// puts `s_waitcnt vmcnt(0)` in front of the first LDS read that MAY alias a pending DMA destination -- always the case when the buffer
// index is a run-time value (buf ^= 1 loops) or the read is ds_read_b64_tr_b16 -- which turns a prefetch into a blocking load.  The
// caller orders the data itself: an explicit s_waitcnt vmcnt + barrier before the tile is read.  lds_wave_base: LDS BYTE address
// (lds_addr_of), wave-uniform.  No other M0 user may live in a kernel that uses this (LDS instructions do not need M0 on gfx9+).
__device__ inline uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const char*)p);
}
__device__ inline void glds16_untracked(const bf16_t* src, uint32_t lds_wave_base) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_wave_base);  // the value is wave-uniform; this tells the compiler so
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
}

// Philox4x32-10 counter RNG (Salmon et al. 2011) — the on-device noise source of the sampler.
struct Philox {
    uint32_t k0, k1;
    __device__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
    __device__ inline void round(uint32_t (&c)[4], uint32_t ka, uint32_t kb) const {
        const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
        uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
        uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
        uint32_t n0 = hi1 ^ c[1] ^ ka, n1 = lo1, n2 = hi0 ^ c[3] ^ kb, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    __device__ inline void gen(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&out)[4]) const {
        uint32_t c[4] = {c0, c1, c2, c3};
        uint32_t ka = k0, kb = k1;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            round(c, ka, kb);
            ka += 0x9E3779B9u;
            kb += 0xBB67AE85u;
        }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
    }
};
// uniform in (0,1): never 0 or 1
__device__ inline float u32_to_unit(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

}  // namespace showo

#define SHOWO_CHECK_HIP(expr)                                  \
    do {                                                       \
        hipError_t _e = (expr);                                \
        if (_e != hipSuccess) return showo::set_error_hip(_e, #expr, __FILE__, __LINE__); \
    } while (0)

namespace showo {
int set_error_hip(hipError_t e, const char* what, const char* file, int line);
int set_error_msg(int code, const char* msg);
}

// multi-tensor AdamW (train_kernels.hip adamw_multi_kernel; table built by train_engine.hip)
namespace showo {
struct AdamSeg { float *p, *m, *v; const float* g; bf16_t* dst16; float* dst32; int64_t n; int decay; };
constexpr int64_t ADAM_CHUNK = 65536;  // elements per block
int adamw_multi_launch(const AdamSeg* segs, const int* seg_of, const int64_t* start_of, int nchunks, float lr, float beta1, float beta2,
                       float eps, float weight_decay, int step, hipStream_t s);
}  // namespace showo
