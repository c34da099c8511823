// Accuracy mode of the transformer engine (showo_engine_set_precision(e, 1)): the reference's fp32 inference
// (inference_t2i.py:67 loads the model in fp32; models/phi.py:1182-1183 returns fp32 logits) reproduced to ~1e-4 end to end.
//
// Every GEMM runs on the split-precision MFMA kernel (showo_gemm_bf16x3: each operand is a (hi, lo) bf16 pair, x = hi + lo to 2^-17,
// products hi*hi + hi*lo + lo*hi accumulated in fp32 -> ~1e-5 relative per GEMM instead of bf16's 4e-3), and everything between the
// GEMMs stays fp32: LayerNorm output, q / k / v, the attention (scores, softmax, P V on the vector ALU with IEEE expf), the exact
// tanh form of gelu_new.  Nothing here is tuned for speed: it is the mode tests and `smoke()` use to show that the engine's
// arithmetic IS the reference's (logits within north_star's 1e-3 of the fp32 reference at full size), while the timed default keeps
// bf16 operands.  Kernels below: LayerNorm -> (hi, lo); q/k LayerNorm + partial RoPE + head-major relayout in fp32
// (models/phi.py:661-694); masked attention in fp32 (phi.py:715-722 / the eager twin :311-397); gelu_new -> (hi, lo).
#include "engine.h"

namespace showo {
namespace {

__device__ inline void split_store(float v, bf16_t* hi, bf16_t* lo, int64_t i) {
    const bf16_t h = f2bf(v);
    hi[i] = h;
    lo[i] = f2bf(v - bf2f(h));
}

// nn.LayerNorm (phi.py:744,776,1065) in fp32, output as a (hi, lo) bf16 pair; one wave per row; row_index gathers input rows
__global__ __launch_bounds__(256) void ln_split_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                       const int32_t* __restrict__ row_index, bf16_t* __restrict__ hi,
                                                       bf16_t* __restrict__ lo, int rows, int H, float eps) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const float* xr = x + (row_index ? (int64_t)row_index[r] : (int64_t)r) * H;
    float s = 0.f;
    for (int i = lane; i < H; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
    for (int i = lane; i < H; i += 64) { const float d = xr[i] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
    for (int i = lane; i < H; i += 64) split_store((xr[i] - mean) * rstd * w[i] + b[i], hi, lo, (int64_t)r * H + i);
}

// The same LayerNorm with the output laid out for the K-concatenated split GEMM on the production kernel: row r of out is
// [hi(H) | lo(H) | hi(H)] (ld = 3 H), to be multiplied with weight rows [w_hi | w_hi | w_lo] (hi*hi + lo*hi + hi*lo in ONE bf16 GEMM
// over K = 3 H with fp32 accumulation).  One wave per row, 8 values per lane per step.
__global__ __launch_bounds__(256) void ln_split3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                        const int32_t* __restrict__ row_index, bf16_t* __restrict__ out, int rows, int H,
                                                        float eps) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const float* xr = x + (row_index ? (int64_t)row_index[r] : (int64_t)r) * H;
    float s = 0.f;
    for (int i = lane * 4; i < H; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + i);
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
    for (int i = lane * 4; i < H; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + i);
        const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
    bf16_t* o = out + (int64_t)r * 3 * H;
    for (int i = lane * 4; i < H; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + i);
        const float4 wv = *reinterpret_cast<const float4*>(w + i);
        const float4 bv = *reinterpret_cast<const float4*>(b + i);
        const float y[4] = {(v.x - mean) * rstd * wv.x + bv.x, (v.y - mean) * rstd * wv.y + bv.y, (v.z - mean) * rstd * wv.z + bv.z,
                            (v.w - mean) * rstd * wv.w + bv.w};
        uint2 hi, lo;
        hi.x = pack_bf2(y[0], y[1]);
        hi.y = pack_bf2(y[2], y[3]);
        lo.x = pack_bf2(y[0] - bf2f((bf16_t)(hi.x & 0xffffu)), y[1] - bf2f((bf16_t)(hi.x >> 16)));
        lo.y = pack_bf2(y[2] - bf2f((bf16_t)(hi.y & 0xffffu)), y[3] - bf2f((bf16_t)(hi.y >> 16)));
        *reinterpret_cast<uint2*>(o + i) = hi;
        *reinterpret_cast<uint2*>(o + H + i) = lo;
        *reinterpret_cast<uint2*>(o + 2 * H + i) = hi;
    }
}

// qkv fp32 [T, 3 nH 64] -> Q fp32 [B,nH,L,64], K / V fp32 [B,nH,Lcap,64] (rows pos0 + l): per-head LayerNorm(64) of q and k, partial
// rotary over dims [0, 32) with rotate_half pairing d <-> d + 16 (phi.py:163-167,681-694).  One wave per (token, head, q|k|v), lane = dim.
__global__ __launch_bounds__(256) void qk_prep_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ qw, const float* __restrict__ qb,
                                                          const float* __restrict__ kw, const float* __restrict__ kb,
                                                          const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                          float* __restrict__ Q, float* __restrict__ K, float* __restrict__ V, int B, int L,
                                                          int nH, float eps, int pos0, int Lcap) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t total = (int64_t)B * L * nH * 3;
    if (item >= total) return;
    const int which = (int)(item % 3);
    const int64_t th = item / 3;
    const int head = (int)(th % nH);
    const int64_t t = th / nH;
    const int b = (int)(t / L), l = (int)(t - (int64_t)b * L), pos = pos0 + l;
    const int Hq = nH * 64;
    float v = qkv[t * 3 * Hq + (int64_t)which * Hq + head * 64 + lane];
    if (which == 2) {
        V[(((int64_t)b * nH + head) * Lcap + pos) * 64 + lane] = v;
        return;
    }
    const float mean = wave_sum(v) * (1.0f / 64.0f);
    const float d = v - mean;
    const float rstd = 1.0f / sqrtf(wave_sum(d * d) * (1.0f / 64.0f) + eps);
    v = d * rstd * (which ? kw : qw)[lane] + (which ? kb : qb)[lane];
    const float partner = __shfl_xor(v, 16, 64);
    if (lane < 32) {
        const float c = cosT[(int64_t)pos * 32 + lane], sn = sinT[(int64_t)pos * 32 + lane];
        v = lane < 16 ? v * c - partner * sn : v * c + partner * sn;
    }
    if (which == 0) Q[(((int64_t)b * nH + head) * L + l) * 64 + lane] = v;
    else K[(((int64_t)b * nH + head) * Lcap + pos) * 64 + lane] = v;
}

// softmax(Q K^T / 8 + mask) V in fp32: one wave per (b, head, query row).  Visibility: two intervals per row (iv), the dense additive
// mask when *flag != 0, or causal (iv == NULL: query r sees keys <= r + Lk - Lq).  Scores of the row live in LDS (Lk <= 2048).
__global__ __launch_bounds__(64) void attn_f32_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                      const int32_t* __restrict__ iv, const int32_t* __restrict__ flag,
                                                      const float* __restrict__ dense, float* __restrict__ O, int nH, int Lq, int Lk, int Lcap,
                                                      int ldo) {
    __shared__ float sc[2048];
    __shared__ float qs[64];
    const int lane = threadIdx.x, r = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int64_t bh = (int64_t)b * nH + h;
    qs[lane] = Q[(bh * Lq + r) * 64 + lane];
    __syncthreads();
    int lo1 = 0, hi1 = r + Lk - Lq + 1, lo2 = 0, hi2 = 0;
    const bool use_dense = flag && *flag != 0 && dense;
    if (iv && !use_dense) {
        const int4 v = reinterpret_cast<const int4*>(iv)[(int64_t)b * Lq + r];
        lo1 = v.x; hi1 = v.y; lo2 = v.z; hi2 = v.w;
    }
    const float* Kb = K + bh * Lcap * 64;
    const float* Vb = V + bh * Lcap * 64;
    float mx = -INFINITY;
    for (int c0 = 0; c0 < Lk; c0 += 64) {
        const int c = c0 + lane;
        float s = -INFINITY;
        if (c < Lk) {
            const bool vis = use_dense || (c >= lo1 && c < hi1) || (c >= lo2 && c < hi2);
            if (vis) {
                const float4* kr = reinterpret_cast<const float4*>(Kb + (int64_t)c * 64);
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float4 kv = kr[j];
                    acc += qs[4 * j] * kv.x;
                    acc += qs[4 * j + 1] * kv.y;
                    acc += qs[4 * j + 2] * kv.z;
                    acc += qs[4 * j + 3] * kv.w;
                }
                s = acc * 0.125f;
                if (use_dense) s += dense[((int64_t)b * Lq + r) * Lk + c];
            }
            sc[c] = s;
        }
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < Lk; c += 64) {
        const float p = sc[c] == -INFINITY ? 0.f : expf(sc[c] - mx);
        sc[c] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    __syncthreads();
    float o = 0.f;
    for (int c = 0; c < Lk; ++c) {
        const float p = sc[c];
        if (p != 0.f) o += p * Vb[(int64_t)c * 64 + lane];
    }
    O[((int64_t)b * Lq + r) * ldo + h * 64 + lane] = o / sum;
}

// gelu_new (transformers NewGELUActivation, phi.py:204-212) with tanhf, output as a (hi, lo) pair
__global__ void gelu_split_kernel(const float* __restrict__ f, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float x = f[i];
        const float g = 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
        split_store(g, hi, lo, i);
    }
}

}  // namespace

int precise_ln_split(const float* x, const float* w, const float* b, const int32_t* row_index, bf16_t* hi, bf16_t* lo, int rows, int H,
                     float eps, hipStream_t s) {
    if (rows <= 0) return 0;
    ln_split_kernel<<<dim3((rows + 3) / 4), dim3(256), 0, s>>>(x, w, b, row_index, hi, lo, rows, H, eps);
    return hipGetLastError() == hipSuccess ? 0 : set_error_msg(7, "precise: ln_split launch failed");
}
int precise_ln_split3(const float* x, const float* w, const float* b, const int32_t* row_index, bf16_t* out, int rows, int H, float eps,
                      hipStream_t s) {
    if (rows <= 0) return 0;
    if (H % 4) return set_error_msg(1, "precise: ln_split3 needs H % 4 == 0");
    ln_split3_kernel<<<dim3((rows + 3) / 4), dim3(256), 0, s>>>(x, w, b, row_index, out, rows, H, eps);
    return hipGetLastError() == hipSuccess ? 0 : set_error_msg(7, "precise: ln_split3 launch failed");
}
int precise_qk_prep(const float* qkv, const float* qw, const float* qb, const float* kw, const float* kb, const float* cosT,
                    const float* sinT, float* Q, float* K, float* V, int B, int L, int nH, float eps, int pos0, int Lcap, hipStream_t s) {
    const int64_t items = (int64_t)B * L * nH * 3;
    if (items <= 0) return 0;
    qk_prep_f32_kernel<<<dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s>>>(qkv, qw, qb, kw, kb, cosT, sinT, Q, K, V, B, L, nH, eps, pos0, Lcap);
    return hipGetLastError() == hipSuccess ? 0 : set_error_msg(7, "precise: qk_prep launch failed");
}
int precise_attention(const float* Q, const float* K, const float* V, const int32_t* iv, const int32_t* flag, const float* dense, float* O,
                      int B, int nH, int Lq, int Lk, int Lcap, int ldo, hipStream_t s) {
    if (B <= 0 || Lq <= 0) return 0;
    if (Lk > 2048) return set_error_msg(5, "precise attention: at most 2048 keys");
    if (dense && Lq != Lk) return set_error_msg(1, "precise attention: a dense mask needs Lq == Lk");
    attn_f32_kernel<<<dim3(Lq, nH, B), dim3(64), 0, s>>>(Q, K, V, iv, flag, dense, O, nH, Lq, Lk, Lcap, ldo);
    return hipGetLastError() == hipSuccess ? 0 : set_error_msg(7, "precise: attention launch failed");
}
int precise_gelu_split(const float* f, bf16_t* hi, bf16_t* lo, int64_t n, hipStream_t s) {
    if (n <= 0) return 0;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    gelu_split_kernel<<<dim3(blocks), dim3(256), 0, s>>>(f, hi, lo, n);
    return hipGetLastError() == hipSuccess ? 0 : set_error_msg(7, "precise: gelu launch failed");
}

}  // namespace showo
