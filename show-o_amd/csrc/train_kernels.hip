// HBM-bound kernels of the training step for gfx950.
// Replaces (reference): the autograd of LayerNorm / gelu_new / q,k LayerNorm + rotary / embedding / cross-entropy that
// loss.backward() runs (training/train.py:612; forward definitions models/phi.py:208-212, 661-694, 774-790,
// models/modeling_showo.py:80-98) and torch.optim.AdamW.step (training/train.py:225-231, 617).
// All reductions are fixed-order (per-block partials + a finalize pass): the same inputs give the same bits.
#include "common.h"
#include <cstdlib>
#include "../../include/showo_hip.h"

using namespace showo;

namespace {

// forward gelu_new of the training path = the GEMM epilogues' form (common.h: x * sigmoid(2u), within ~1e-6 relative of the tanh form,
// far below the bf16 rounding of its result): showo_gelu_bf16, the gelu(f)^T recomputation of the dW2 operand and the
// save-for-backward projection epilogue (showo_gemm_qkv_fc1_save_bf16) all produce the same bits
__device__ __forceinline__ float gelu_new_f(float x) { return gelu_new_fast(x); }
// d/dx [0.5 x (1 + tanh(u))],  u = c (x + 0.044715 x^3)
__device__ __forceinline__ float gelu_new_grad(float x) {
    const float c = 0.7978845608028654f;
    const float u = c * (x + 0.044715f * x * x * x);
    // tanh(u) = 1 - 2 / (1 + e^(2u)) with the hardware exp / rcp (the forward's gelu_new_fast uses the same pair): ~1e-6 relative, far
    // below the bf16 rounding of the product; libm's tanhf made the dgelu pass VALU-bound (132 us where its HBM floor is ~100)
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * u));
    return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * c * (1.0f + 3.0f * 0.044715f * x * x);
}

// ------------------------------------------------------------------------------------------------
// X bf16 [T, C] (row stride ld) -> XT bf16 [C, Tp] (columns >= T zero).  mode 1: XT = gelu_new(X)^T.
// colpart (optional): fp32 [gridDim.y, C] per-row-block column sums of X (bias gradients), summed by colsum_finalize.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ xt, float* __restrict__ colpart,
                                                        int T, int C, int ld, int Tp, int mode) {
    __shared__ bf16_t t[64][66];
    __shared__ float cs[4][64];
    const int tid = threadIdx.x, c0 = blockIdx.x * 64, t0 = blockIdx.y * 64;
    {
        const int r = tid >> 2, cq = (tid & 3) * 16;  // row r, 16 columns
        float v[16];
        const bool rok = t0 + r < T;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4 u = make_uint4(0, 0, 0, 0);
            if (rok && c0 + cq + 8 * h < C) u = *reinterpret_cast<const uint4*>(x + (int64_t)(t0 + r) * ld + c0 + cq + 8 * h);
            const bf16_t* e = reinterpret_cast<const bf16_t*>(&u);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[8 * h + j] = bf2f(e[j]);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float w = v[j];
            if (mode == 1) w = rok ? gelu_new_f(w) : 0.f;
            t[r][cq + j] = f2bf(w);
        }
    }
    __syncthreads();
    if (colpart) {  // column sums of the (untransformed) tile, fixed order: 4 partial sums of 16 rows, then 4 -> 1
        const int c = tid & 63, part = tid >> 6;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += bf2f(t[part * 16 + r][c]);
        cs[part][c] = s;
    }
    {
        const int c = tid >> 2, ts = (tid & 3) * 16;
        if (c0 + c < C) {
            uint32_t w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = (uint32_t)t[ts + 2 * j][c] | ((uint32_t)t[ts + 2 * j + 1][c] << 16);
            uint4* o = reinterpret_cast<uint4*>(xt + (int64_t)(c0 + c) * Tp + t0 + ts);
            o[0] = make_uint4(w[0], w[1], w[2], w[3]);
            o[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
    }
    if (colpart) {
        __syncthreads();
        if (tid < 64 && c0 + tid < C) colpart[(int64_t)blockIdx.y * C + c0 + tid] = (cs[0][tid] + cs[1][tid]) + (cs[2][tid] + cs[3][tid]);
    }
}
// out[c] (+)= sum over blocks of part[blk][c], bit-reproducible: block k belongs to subset k % 8; inside a subset wave w of
// the reducing block sums its blocks k = j + 8 (w + 4 i) in order and the four waves combine as (0+1)+(2+3); the 8 subset
// sums are written to scratch[8][C] (the tail of the partial buffer) and combined as ((0+1)+(2+3))+((4+5)+(6+7)).
__global__ __launch_bounds__(256) void colsum_subset_kernel(const float* __restrict__ part, float* __restrict__ sub8, int nblk, int C) {
    __shared__ float red[4][64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = blockIdx.y;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < C)
        for (int k = j + 8 * w; k < nblk; k += 32) s += part[(int64_t)k * C + c];
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < C) sub8[(int64_t)j * C + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
__global__ void colsum_combine_kernel(const float* __restrict__ sub8, float* __restrict__ out, float* __restrict__ out2, int C1, int C,
                                      int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = ((sub8[c] + sub8[(int64_t)C + c]) + (sub8[2 * (int64_t)C + c] + sub8[3 * (int64_t)C + c])) +
                    ((sub8[4 * (int64_t)C + c] + sub8[5 * (int64_t)C + c]) + (sub8[6 * (int64_t)C + c] + sub8[7 * (int64_t)C + c]));
    float* o = c < C1 ? out + c : out2 + (c - C1);
    *o = accumulate ? *o + s : s;
}
// part: [nblk][C] partials followed by 8*C floats of scratch.  Columns >= C1 go to out2[c - C1] when out2 is given (two destinations
// from one partial buffer: the LayerNorm backward).  (A one-launch form -- 64 columns per 512-thread block, eight waves over the
// partial rows -- measured 1.5-3 ms per training step SLOWER than these two launches: C / 64 blocks leave most CUs idle.)
static inline void colsum_reduce(const float* part, float* out, int nblk, int C, int accumulate, hipStream_t s, float* out2 = nullptr, int C1 = 0) {
    float* sub8 = const_cast<float*>(part) + (int64_t)nblk * C;
    colsum_subset_kernel<<<dim3((C + 63) / 64, 8), dim3(256), 0, s>>>(part, sub8, nblk, C);
    colsum_combine_kernel<<<dim3((C + 255) / 256), dim3(256), 0, s>>>(sub8, out, out2, out2 ? C1 : C, C, accumulate);
}

// Column sums of a token-major bf16 matrix X [T, C] (row stride ld): the bias gradient of a Linear is the column sum of its dY.
// Block (cb, rb) sums rows 32 rb .. 32 rb + 31 of columns 2048 cb ..; a thread owns 8 consecutive columns (one 16-byte load per row,
// a row of the block is 4 KiB contiguous; 32 rows: ~8 resident blocks per CU at the [T, 3H] shape, 64 left it at 2 and 2.5 TB/s).
// part[rb][c] partials in row order, reduced by colsum_reduce: no atomics.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, int T, int C, int ld) {
    const int c0 = (blockIdx.x * 256 + threadIdx.x) * 8, r0 = blockIdx.y * 32;
    if (c0 >= C) return;
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    const int rend = min(r0 + 32, T);
    const bf16_t* p = x + (int64_t)r0 * ld + c0;
    const bool full = c0 + 8 <= C;
#pragma unroll 8
    for (int r = r0; r < rend; ++r, p += ld) {
        if (full) {
            const uint4 u = *reinterpret_cast<const uint4*>(p);
            const bf16_t* e = reinterpret_cast<const bf16_t*>(&u);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += bf2f(e[j]);
        } else {
            for (int j = 0; j < 8; ++j)
                if (c0 + j < C) a[j] += bf2f(p[j]);
        }
    }
    float* o = part + (int64_t)blockIdx.y * C + c0;
    for (int j = 0; j < 8; ++j)
        if (c0 + j < C) o[j] = a[j];
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward fused with the residual add of the parallel block (y = x + f(LN(x)), phi.py:774-790):
//   dx = dy + LN'(x)^T dh.   One wave per row; a block walks LNB_ROWS rows and keeps per-lane column partials of
//   dgamma / dbeta, written as part[blk][2][H].  dx goes to dx32 (may alias dy) and, rounded, to dx16 (GEMM operand).
// ------------------------------------------------------------------------------------------------
constexpr int LNB_ROWS = 16;  // 702 blocks at the stage-1 batch: two resident blocks on every CU (32 rows left a third of the slots empty)
// DXS: also the column sums of the ROUNDED dx (dx16) -> part2[blk][H]: dx is the output gradient of the block below, and the bias
// gradients of its dense and fc2 are exactly these column sums (saves a pass over dx16)
template <bool DXS>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ dh, const float* dy, float* dx32,
                                                     bf16_t* __restrict__ dx16, float* __restrict__ part, float* __restrict__ part2, int T,
                                                     int H, float eps) {
    extern __shared__ float red[];  // [4][2][H] cross-wave reduction of the column partials
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nv = (H + 255) / 256;  // float4 groups per lane (H % 4 == 0; lanes past the row end idle)
    float ag[8][4], ab[8][4], ad[DXS ? 8 : 1][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; if (DXS) ad[DXS ? i : 0][j] = 0.f; }
    const int r0 = blockIdx.x * LNB_ROWS;
    for (int rr = wave; rr < LNB_ROWS; rr += 4) {
        const int r = r0 + rr;
        if (r >= T) break;
        const float* xr = x + (int64_t)r * H;
        const float* gr = dh + (int64_t)r * H;
        // the three rows this iteration reads (x, dh, dy: 24 KB) are requested up front: one HBM round trip per row instead of three
        float4 xv[8], gv[8], ov[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nv && (i * 64 + lane) * 4 < H) {
                const int c = (i * 64 + lane) * 4;
                xv[i] = *reinterpret_cast<const float4*>(xr + c);
                gv[i] = *reinterpret_cast<const float4*>(gr + c);
                ov[i] = *reinterpret_cast<const float4*>(dy + (int64_t)r * H + c);
            }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nv && (i * 64 + lane) * 4 < H) s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
        const float mean = wave_sum(s) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nv && (i * 64 + lane) * 4 < H) {
                xv[i].x -= mean; xv[i].y -= mean; xv[i].z -= mean; xv[i].w -= mean;
                q += (xv[i].x * xv[i].x + xv[i].y * xv[i].y) + (xv[i].z * xv[i].z + xv[i].w * xv[i].w);
            }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
        float s1 = 0.f, s2 = 0.f;  // sum(g), sum(g * xhat) with g = dh * gamma
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nv && (i * 64 + lane) * 4 < H) {
                const int c = (i * 64 + lane) * 4;
                const float4 d = gv[i];
                const float4 w = *reinterpret_cast<const float4*>(gamma + c);
                xv[i].x *= rstd; xv[i].y *= rstd; xv[i].z *= rstd; xv[i].w *= rstd;  // xhat
                ag[i][0] += d.x * xv[i].x; ag[i][1] += d.y * xv[i].y; ag[i][2] += d.z * xv[i].z; ag[i][3] += d.w * xv[i].w;
                ab[i][0] += d.x; ab[i][1] += d.y; ab[i][2] += d.z; ab[i][3] += d.w;
                gv[i] = make_float4(d.x * w.x, d.y * w.y, d.z * w.z, d.w * w.w);
                s1 += (gv[i].x + gv[i].y) + (gv[i].z + gv[i].w);
                s2 += (gv[i].x * xv[i].x + gv[i].y * xv[i].y) + (gv[i].z * xv[i].z + gv[i].w * xv[i].w);
            }
        const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nv && (i * 64 + lane) * 4 < H) {
                const int c = (i * 64 + lane) * 4;
                float4 o = ov[i];
                o.x += rstd * (gv[i].x - m1 - xv[i].x * m2);
                o.y += rstd * (gv[i].y - m1 - xv[i].y * m2);
                o.z += rstd * (gv[i].z - m1 - xv[i].z * m2);
                o.w += rstd * (gv[i].w - m1 - xv[i].w * m2);
                *reinterpret_cast<float4*>(dx32 + (int64_t)r * H + c) = o;
                if (dx16) {
                    uint2 pk;
                    pk.x = pack_bf2(o.x, o.y);
                    pk.y = pack_bf2(o.z, o.w);
                    *reinterpret_cast<uint2*>(dx16 + (int64_t)r * H + c) = pk;
                    if (DXS) {
                        ad[DXS ? i : 0][0] += bf2f((bf16_t)(pk.x & 0xffffu)); ad[DXS ? i : 0][1] += bf2f((bf16_t)(pk.x >> 16));
                        ad[DXS ? i : 0][2] += bf2f((bf16_t)(pk.y & 0xffffu)); ad[DXS ? i : 0][3] += bf2f((bf16_t)(pk.y >> 16));
                    }
                }
            }
    }
    // block partials: red[wave][0|1][c]
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i < nv && (i * 64 + lane) * 4 < H) {
            const int c = (i * 64 + lane) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                red[(wave * 2 + 0) * H + c + j] = ag[i][j];
                red[(wave * 2 + 1) * H + c + j] = ab[i][j];
            }
        }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * H; c += 256) {
        const int which = c / H, col = c - which * H;
        const float v = (red[(0 * 2 + which) * H + col] + red[(1 * 2 + which) * H + col]) +
                        (red[(2 * 2 + which) * H + col] + red[(3 * 2 + which) * H + col]);
        part[((int64_t)blockIdx.x * (DXS ? 3 : 2) + which) * H + col] = v;
    }
    if (DXS) {  // second use of the reduction buffer: the column sums of dx16
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < nv && (i * 64 + lane) * 4 < H) {
                const int c = (i * 64 + lane) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) red[wave * H + c + j] = ad[DXS ? i : 0][j];
            }
        __syncthreads();
        for (int col = threadIdx.x; col < H; col += 256)
            part2[(int64_t)blockIdx.x * 3 * H + col] = (red[col] + red[H + col]) + (red[2 * H + col] + red[3 * H + col]);
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of q/k LayerNorm(64) + partial rotary (rotary_dim 32) + the 1/8 fold of Q (phi.py:661-694).
//   in : dq, dk bf16 token-major [T, ldg] (gradients w.r.t. the STORED Q (= rope(ln(q)) / 8) and K), raw qkv bf16 [T, 3H]
//   out: dqkv bf16 [T, 3H] q and k sections (the v section is written by the attention backward);
//        part[blk][4][64] = per-block partials of (dqln_w, dqln_b, dkln_w, dkln_b).
//   One wave per (token, head) row, lane = head dim.
// ------------------------------------------------------------------------------------------------
// Round 3 layout: a (token, head) row of 64 dims is held by 8 lanes x 8 consecutive dims (one 16-B load per operand), so a wave
// handles 8 rows -- the 8 consecutive heads of a token = 1 KiB contiguous per load instruction -- and every 64-dim reduction is 3
// DPP steps over the 8 lanes of a row instead of 6 over a whole wave (round 2's one-lane-per-dim form ran at 5 x its byte floor:
// 2-byte loads and 8 wave-wide reductions per row).  The rotate_half partner d <-> d + 16 is lane ^ 2.
// Algorithmic bytes per token: dq, dk, q, k in (4 x 2H) + dq', dk' out (2 x 2H) = 12 H B = 276 MB at T = 11 223, H = 2048.
constexpr int QKB_ROWS = 64;  // rows per wave (8 iterations of 8 rows); a 4-wave block covers 256 rows
__device__ __forceinline__ float row8_sum(float v) {  // sum over the 8 lanes that share a row (lane bits 0..2)
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}
__device__ __forceinline__ void unpack8(const uint4 u, float (&f)[8]) {
    f[0] = bf2f((bf16_t)(u.x & 0xffffu)); f[1] = bf2f((bf16_t)(u.x >> 16));
    f[2] = bf2f((bf16_t)(u.y & 0xffffu)); f[3] = bf2f((bf16_t)(u.y >> 16));
    f[4] = bf2f((bf16_t)(u.z & 0xffffu)); f[5] = bf2f((bf16_t)(u.z >> 16));
    f[6] = bf2f((bf16_t)(u.w & 0xffffu)); f[7] = bf2f((bf16_t)(u.w >> 16));
}
__global__ __launch_bounds__(256) void qkln_rope_bwd_kernel(const bf16_t* __restrict__ dq, const bf16_t* __restrict__ dk, int ldg,
                                                            const bf16_t* __restrict__ qkv, const float* __restrict__ qw,
                                                            const float* __restrict__ kw, const float* __restrict__ cosT,
                                                            const float* __restrict__ sinT, bf16_t* __restrict__ dqkv,
                                                            float* __restrict__ part, int T, int L, int nH, float eps) {
    __shared__ float red[4][4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sub = lane & 7, slot = lane >> 3;  // dims 8 sub .. 8 sub + 7 of row `slot` of the iteration
    const int H = nH * 64;
    const int64_t rows = (int64_t)T * nH;
    float acc[4][8];  // per-lane partials of (dq_ln_w, dq_ln_b, dk_ln_w, dk_ln_b) for the lane's 8 dims
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[p][j] = 0.f;
    float wq[8], wk[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { wq[j] = qw[8 * sub + j]; wk[j] = kw[8 * sub + j]; }
    const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * QKB_ROWS;
    for (int it = 0; it < QKB_ROWS / 8; ++it) {
        const int64_t row = base + it * 8 + slot;
        const bool live = row < rows;
        const int64_t rr = live ? row : rows - 1;  // clamped rows compute on valid memory; they neither store nor accumulate
        const int tok = (int)(rr / nH), head = (int)(rr - (int64_t)tok * nH);
        const int pos = tok % L;  // training sequences start at position 0 (phi.py:998-1003)
        float c[8], sn[8];
        if (sub < 4) {
            const float4 c0 = *reinterpret_cast<const float4*>(cosT + (int64_t)pos * 32 + 8 * sub), c1 = *reinterpret_cast<const float4*>(cosT + (int64_t)pos * 32 + 8 * sub + 4);
            const float4 s0 = *reinterpret_cast<const float4*>(sinT + (int64_t)pos * 32 + 8 * sub), s1 = *reinterpret_cast<const float4*>(sinT + (int64_t)pos * 32 + 8 * sub + 4);
            c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w; c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
            sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { c[j] = 1.f; sn[j] = 0.f; }
        }
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            float g[8], x[8];
            unpack8(*reinterpret_cast<const uint4*>((which ? dk : dq) + (int64_t)tok * ldg + head * 64 + 8 * sub), g);
            unpack8(*reinterpret_cast<const uint4*>(qkv + (int64_t)tok * 3 * H + which * H + head * 64 + 8 * sub), x);
            const float sc = which ? 1.0f : 0.125f;
            float dy[8];
            float sx = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                g[j] *= sc;
                // rope^T: y'[d] = y[d] c[d] + rot(y)[d] s[d], rot(y)[d] = -y[d+16] (d<16), +y[d-16] (16<=d<32)
                //   => dy[d] = g[d] c[d] + (d < 16 ? g[d+16] s[d+16] : -g[d-16] s[d-16])
                const float other = __shfl_xor(g[j] * sn[j], 2, 64);  // dims d +- 16 live two lanes away
                dy[j] = g[j] * c[j];
                if (sub < 2) dy[j] += other; else if (sub < 4) dy[j] -= other;
                sx += x[j];
            }
            const float mean = row8_sum(sx) * (1.0f / 64.0f);
            float sq = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { x[j] -= mean; sq += x[j] * x[j]; }
            const float rstd = 1.0f / sqrtf(row8_sum(sq) * (1.0f / 64.0f) + eps);
            float s1 = 0.f, s2 = 0.f;
            float gg[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[j] *= rstd;  // xhat
                if (live) { acc[2 * which][j] += dy[j] * x[j]; acc[2 * which + 1][j] += dy[j]; }
                gg[j] = dy[j] * (which ? wk[j] : wq[j]);
                s1 += gg[j];
                s2 += gg[j] * x[j];
            }
            const float m1 = row8_sum(s1) * (1.0f / 64.0f), m2 = row8_sum(s2) * (1.0f / 64.0f);
            uint4 o;
            o.x = pack_bf2(rstd * (gg[0] - m1 - x[0] * m2), rstd * (gg[1] - m1 - x[1] * m2));
            o.y = pack_bf2(rstd * (gg[2] - m1 - x[2] * m2), rstd * (gg[3] - m1 - x[3] * m2));
            o.z = pack_bf2(rstd * (gg[4] - m1 - x[4] * m2), rstd * (gg[5] - m1 - x[5] * m2));
            o.w = pack_bf2(rstd * (gg[6] - m1 - x[6] * m2), rstd * (gg[7] - m1 - x[7] * m2));
            if (live) *reinterpret_cast<uint4*>(dqkv + (int64_t)tok * 3 * H + which * H + head * 64 + 8 * sub) = o;
        }
    }
    // fixed-order reductions: over the 8 row slots of the wave (lane bits 3..5), then over the 4 waves through LDS
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[p][j];
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (slot == 0) red[wave][p][8 * sub + j] = v;
        }
    __syncthreads();
    {
        const int which = threadIdx.x >> 6;
        part[((int64_t)blockIdx.x * 4 + which) * 64 + lane] =
            (red[0][which][lane] + red[1][which][lane]) + (red[2][which][lane] + red[3][which][lane]);
    }
}

// ------------------------------------------------------------------------------------------------
// Cross-entropy of Showo.forward (modeling_showo.py:80-98).  A logits row r = (b, l) can carry up to two targets:
//   A = labels[b, l]     with weight wa  (mask-token prediction rows: b < b_t2i, l > max_seq_len)
//   B = labels[b, l+1]   with weight wb  (next-token rows of the lm and mmu slices, l < L-1)
// ce_rows_kernel builds (labA, labB, group bits) and the three valid-label counts; ce_kernel computes per row
//   lse, ceA = lse - z[A], ceB = lse - z[B] and dlogits = (wa + wb) softmax - wa onehot(A) - wb onehot(B)  (bf16, ld Vp);
// ce_finalize sums the row losses per group in row order (mean over valid labels, like F.cross_entropy).
// ------------------------------------------------------------------------------------------------
struct CeRow { int labA, labB, bits; };  // bits: 1 = t2i row, 2 = lm row, 4 = mmu row (valid label only)
__global__ void ce_rows_kernel(const int64_t* __restrict__ labels, CeRow* __restrict__ rows, int* __restrict__ counts, int B, int L,
                               int b_t2i, int b_lm, int b_mmu, int max_seq_len) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B * L) return;
    const int b = r / L, l = r - b * L;
    CeRow o;
    o.labA = -100; o.labB = -100; o.bits = 0;
    if (b < b_t2i && l >= max_seq_len + 1) {
        const int64_t t = labels[r];
        if (t != -100) { o.labA = (int)t; o.bits |= 1; atomicAdd(&counts[0], 1); }
    }
    if (l < L - 1) {
        const int64_t t = labels[r + 1];
        if (t != -100) {
            const bool in_lm = b >= b_t2i && b < b_t2i + b_lm;
            const bool in_mmu = b_mmu == 0 ? true : b >= B - b_mmu;  // logits[-0:] selects the whole batch (reference quirk)
            if (in_lm) { o.bits |= 2; atomicAdd(&counts[1], 1); }
            if (in_mmu) { o.bits |= 4; atomicAdd(&counts[2], 1); }
            if (in_lm || in_mmu) o.labB = (int)t;
        }
    }
    rows[r] = o;
}

__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, int ldl, const CeRow* __restrict__ rows,
                                                 const int* __restrict__ counts, float g_t2i, float g_lm, float g_mmu,
                                                 bf16_t* __restrict__ dlogits, int ldd, float* __restrict__ rowloss, int V) {
    __shared__ float sred[4];
    __shared__ float sbc;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const CeRow cr = rows[r];
    const float* z = logits + (int64_t)r * ldl;
    bf16_t* dz = dlogits ? dlogits + (int64_t)r * ldd : nullptr;
    float wa = 0.f, wb = 0.f;
    if (cr.bits & 1) wa = g_t2i / (float)counts[0];
    if (cr.bits & 2) wb += g_lm / (float)counts[1];
    if (cr.bits & 4) wb += g_mmu / (float)counts[2];
    if (cr.bits == 0) {  // no target: zero gradient row, no loss
        if (dz)
            for (int i = tid * 8; i < ldd; i += 256 * 8) *reinterpret_cast<uint4*>(dz + i) = make_uint4(0, 0, 0, 0);
        if (tid == 0) { rowloss[2 * r] = 0.f; rowloss[2 * r + 1] = 0.f; }
        return;
    }
    // rows are read as 8-byte pairs (ldl even, base 8-byte aligned: every row then starts on a pair boundary), four pairs in flight per
    // thread: the three passes are latency-bound on a 234 KB row otherwise.  V2 = number of whole pairs read that way.
    const bool pairs = ((ldl & 1) == 0) && ((reinterpret_cast<uintptr_t>(logits) & 7) == 0);
    const int V2 = pairs ? (V >> 1) : 0;
    const float2* z2 = reinterpret_cast<const float2*>(z);
    float mx = -INFINITY;
#pragma unroll 4
    for (int i = tid; i < V2; i += 256) { const float2 p = z2[i]; mx = fmaxf(mx, fmaxf(p.x, p.y)); }
    for (int i = 2 * V2 + tid; i < V; i += 256) mx = fmaxf(mx, z[i]);
    mx = wave_max(mx);
    if (lane == 0) sred[wave] = mx;
    __syncthreads();
    if (tid == 0) sbc = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
    __syncthreads();
    mx = sbc;
    float s = 0.f;
#pragma unroll 4
    for (int i = tid; i < V2; i += 256) { const float2 p = z2[i]; s += expf(p.x - mx) + expf(p.y - mx); }
    for (int i = 2 * V2 + tid; i < V; i += 256) s += expf(z[i] - mx);
    s = wave_sum(s);
    __syncthreads();
    if (lane == 0) sred[wave] = s;
    __syncthreads();
    if (tid == 0) sbc = (sred[0] + sred[1]) + (sred[2] + sred[3]);
    __syncthreads();
    const float tot = sbc;
    const float lse = mx + logf(tot);
    if (tid == 0) {
        // a label outside [0, V) (other than the ignore index, which never sets a bit) is a caller bug -- e.g. image tokens that were
        // not offset by len(tokenizer); F.cross_entropy raises a device assert in the reference.  Never read out of bounds: the
        // row's loss becomes NaN, which poisons the mean of its group.
        const bool okA = !(cr.bits & 1) || (cr.labA >= 0 && cr.labA < V);
        const bool okB = !(cr.bits & 6) || (cr.labB >= 0 && cr.labB < V);
        rowloss[2 * r] = (cr.bits & 1) ? (okA ? lse - z[cr.labA] : __builtin_nanf("")) : 0.f;
        rowloss[2 * r + 1] = (cr.bits & 6) ? (okB ? lse - z[cr.labB] : __builtin_nanf("")) : 0.f;
    }
    if (dz) {
        // same rule for the gradient: a row whose label is out of range gets a NaN gradient row, so the wrong batch cannot update the
        // weights silently (the NaN reaches every parameter through the lm_head dgrad / wgrad; the reference aborts with a device assert)
        const bool bad = ((cr.bits & 1) && (cr.labA < 0 || cr.labA >= V)) || ((cr.bits & 6) && (cr.labB < 0 || cr.labB >= V));
        const float w = bad ? __builtin_nanf("") : wa + wb, inv = 1.0f / tot;
        for (int i0 = tid * 8; i0 < ldd; i0 += 256 * 8) {
            float zz[8];
            if (i0 + 8 <= 2 * V2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float2 p = z2[(i0 >> 1) + j]; zz[2 * j] = p.x; zz[2 * j + 1] = p.y; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) zz[j] = i0 + j < V ? z[i0 + j] : 0.f;
            }
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j;
                float g = 0.f;
                if (i < V) {
                    g = w * expf(zz[j] - mx) * inv;
                    if (i == cr.labA) g -= wa;
                    if (i == cr.labB) g -= wb;
                }
                v[j] = g;
            }
            uint4 u;
            u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
            *reinterpret_cast<uint4*>(dz + i0) = u;
        }
    }
}
// losses[g] = sum over the rows of group g / count[g].  Fixed order: thread t sums rows t, t+256, .. (double), then a
// fixed binary tree over the 256 partials.
__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* __restrict__ rowloss, const CeRow* __restrict__ rows,
                                                          const int* __restrict__ counts, float* __restrict__ losses, int R) {
    __shared__ double red[3][256];
    const int t = threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int r = t; r < R; r += 256) {
        const int bits = rows[r].bits;
        if (bits & 1) s0 += rowloss[2 * r];
        if (bits & 2) s1 += rowloss[2 * r + 1];
        if (bits & 4) s2 += rowloss[2 * r + 1];
    }
    red[0][t] = s0; red[1][t] = s1; red[2][t] = s2;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) { red[0][t] += red[0][t + w]; red[1][t] += red[1][t + w]; red[2][t] += red[2][t + w]; }
        __syncthreads();
    }
    if (t < 3) losses[t] = (float)(red[t][0] / (double)counts[t]);  // 0/0 -> nan like F.cross_entropy on an empty selection
}

// ------------------------------------------------------------------------------------------------
// Embedding backward, deterministic: tokens are ranked by (id, position) with a counting pass, then every run of equal
// ids is summed in position order by one block (4 waves x 512 columns at H = 2048).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_rank_kernel(const int64_t* __restrict__ ids, int* __restrict__ order, int* __restrict__ runstart, int T) {
    // O(T^2) counting rank, T <= ~16k tokens per step.  A block ranks 64 tokens (lane = token); its four waves each scan a quarter of
    // the positions (the scanned id is wave-uniform: scalar loads) and the four partial counts are added through LDS.
    __shared__ int srank[4][64], searlier[4][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const int64_t id = t < T ? ids[t] : 0;
    const int per = (T + 3) / 4, u0 = part * per, u1 = min(T, u0 + per);
    int rank = 0, earlier = 0;
    for (int u = u0; u < u1; ++u) {
        const int64_t o = ids[u];
        const int eq_before = (o == id) & (u < t);
        rank += (o < id) | eq_before;
        earlier |= eq_before;
    }
    srank[part][lane] = rank;
    searlier[part][lane] = earlier;
    __syncthreads();
    if (part == 0 && t < T) {
        rank = (srank[0][lane] + srank[1][lane]) + (srank[2][lane] + srank[3][lane]);
        earlier = searlier[0][lane] | searlier[1][lane] | searlier[2][lane] | searlier[3][lane];
        order[rank] = t;
        runstart[rank] = earlier ? 0 : 1;
    }
}
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ ids, const int* __restrict__ order,
                                                        const int* __restrict__ runstart, const float* __restrict__ dx,
                                                        float* __restrict__ dE, int T, int H) {
    const int i = blockIdx.x;
    if (!runstart[i]) return;
    int end = i + 1;
    while (end < T && !runstart[end]) ++end;
    const int64_t id = ids[order[i]];
    for (int c = threadIdx.x * 4; c < H; c += 1024) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int j = i;
        for (; j + 8 <= end; j += 8) {  // 8 independent row loads in flight, summed in position order
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(dx + (int64_t)order[j + u] * H + c);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; j < end; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(dx + (int64_t)order[j] * H + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(dE + id * H + c) = s;
    }
}

// ------------------------------------------------------------------------------------------------
// AdamW (torch.optim.AdamW semantics: decoupled decay, bias-corrected moments) on fp32 master weights.
// ------------------------------------------------------------------------------------------------
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             int64_t n, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float w = p[i];
        const float gr = g[i];
        w *= 1.0f - lr * wd;
        const float mm = beta1 * m[i] + (1.0f - beta1) * gr;
        const float vv = beta2 * v[i] + (1.0f - beta2) * gr * gr;
        m[i] = mm;
        v[i] = vv;
        const float denom = sqrtf(vv) / sqrtf(bc2) + eps;
        p[i] = w - (lr / bc1) * (mm / denom);
    }
}

// Multi-tensor form: ONE launch updates every bound parameter and writes the engine's refreshed weight image (bf16 cast or fp32 copy)
// from the register that holds the new master value -- 437 adamw launches + 181 casts + 256 copies per step become one kernel, and the
// 5.8 GB of master weights are not read a second time by the cast.  Same expressions as adamw_kernel element by element (same bits).
// chunk c = (segment seg_of[c], elements [start_of[c], start_of[c] + ADAM_CHUNK)).
__global__ __launch_bounds__(256) void adamw_multi_kernel(const showo::AdamSeg* __restrict__ segs, const int* __restrict__ seg_of,
                                                          const int64_t* __restrict__ start_of, float lr, float beta1, float beta2, float eps,
                                                          float wd_all, float bc1, float bc2) {
    const showo::AdamSeg sg = segs[seg_of[blockIdx.x]];
    const int64_t i0 = start_of[blockIdx.x];
    const int64_t i1 = i0 + showo::ADAM_CHUNK < sg.n ? i0 + showo::ADAM_CHUNK : sg.n;
    const float wd = sg.decay ? wd_all : 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        float w = sg.p[i];
        const float gr = sg.g[i];
        w *= 1.0f - lr * wd;
        const float mm = beta1 * sg.m[i] + (1.0f - beta1) * gr;
        const float vv = beta2 * sg.v[i] + (1.0f - beta2) * gr * gr;
        sg.m[i] = mm;
        sg.v[i] = vv;
        const float denom = sqrtf(vv) / sqrtf(bc2) + eps;
        const float pn = w - (lr / bc1) * (mm / denom);
        sg.p[i] = pn;
        if (sg.dst16) sg.dst16[i] = f2bf(pn);
        else if (sg.dst32) sg.dst32[i] = pn;
    }
}

// ---- global-norm clipping of the flat gradient buffer (reference training/train.py:614-615 accelerator.clip_grad_norm_) ------------
// Fixed-order reduction: block b sums the squares of its contiguous slice (per-thread fp32 chains over a fixed stride, then a fixed
// tree in double), ONE block sums the partials in index order -> total norm and the clip coefficient in device memory; the scale
// pass reads the coefficient there and leaves the buffer untouched when it is >= 1.  Two runs give identical bits.
constexpr int CLIP_PARTS = 2048;
__global__ __launch_bounds__(256) void sumsq_part_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ part) {
    __shared__ double sh[256];
    const int64_t chunk = ((n + CLIP_PARTS - 1) / CLIP_PARTS + 3) & ~(int64_t)3;
    const int64_t i0 = (int64_t)blockIdx.x * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
    if (i0 >= n) {  // slices beyond the end (the chunk is rounded up to a multiple of 4)
        if (threadIdx.x == 0) part[blockIdx.x] = 0.0;
        return;
    }
    float a = 0.f;
    for (int64_t i = i0 + threadIdx.x * 4; i + 3 < i1; i += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(g + i);
        a += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (threadIdx.x == 0) for (int64_t i = i0 + ((i1 - i0) & ~(int64_t)3); i < i1; ++i) a += g[i] * g[i];  // tail of the last slice
    sh[threadIdx.x] = (double)a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void clip_coef_kernel(const double* __restrict__ part, float max_norm, float* __restrict__ out2) {
    __shared__ double sh[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < CLIP_PARTS; i += 256) a += part[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float total = (float)sqrt(sh[0]);
        out2[0] = total;
        // torch.nn.utils.clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max = 1).  A NaN norm gives a NaN coefficient there (clamp keeps
        // NaN) and every gradient becomes NaN; fminf would return 1 and let AdamW update from the finite part of a poisoned gradient
        // (ADVICE r5): propagate it.  (An infinite norm gives 0 like the reference: max_norm / inf.)
        out2[1] = (total != total) ? total : fminf(max_norm / (total + 1e-6f), 1.0f);
    }
}
__global__ __launch_bounds__(256) void scale_by_dev_kernel(float* __restrict__ g, int64_t n, const float* __restrict__ coef) {
    const float c = *coef;
    if (c >= 1.0f) return;  // (false for a NaN coefficient: the scale pass then runs and poisons every gradient, as the reference does)
    const int64_t stride = (int64_t)gridDim.x * 1024;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < n; i += stride) {
        float4 v = *reinterpret_cast<float4*>(g + i);
        v.x *= c; v.y *= c; v.z *= c; v.w *= c;
        *reinterpret_cast<float4*>(g + i) = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) for (int64_t i = n & ~(int64_t)3; i < n; ++i) g[i] *= c;
}

__global__ void scale_f32_kernel(float* __restrict__ x, int64_t n, float s) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) x[i] *= s;
}

__global__ void gelu_kernel(const bf16_t* __restrict__ f, bf16_t* __restrict__ a, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
    for (; i < n; i += stride) {
        const uint4 b = *reinterpret_cast<const uint4*>(f + i);
        const bf16_t* eb = reinterpret_cast<const bf16_t*>(&b);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = gelu_new_f(bf2f(eb[j]));
        uint4 u;
        u.x = pack_bf2(o[0], o[1]); u.y = pack_bf2(o[2], o[3]); u.z = pack_bf2(o[4], o[5]); u.w = pack_bf2(o[6], o[7]);
        *reinterpret_cast<uint4*>(a + i) = u;
    }
}

// dgrad epilogue helper: d_f bf16 = d_a (bf16) * gelu_new'(f)   (elementwise, when not fused in a GEMM epilogue)
__global__ void dgelu_kernel(const bf16_t* __restrict__ da, const bf16_t* __restrict__ f, bf16_t* __restrict__ df, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
    for (; i < n; i += stride) {
        const uint4 a = *reinterpret_cast<const uint4*>(da + i);
        const uint4 b = *reinterpret_cast<const uint4*>(f + i);
        const bf16_t* ea = reinterpret_cast<const bf16_t*>(&a);
        const bf16_t* eb = reinterpret_cast<const bf16_t*>(&b);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = bf2f(ea[j]) * gelu_new_grad(bf2f(eb[j]));
        uint4 u;
        u.x = pack_bf2(o[0], o[1]); u.y = pack_bf2(o[2], o[3]); u.z = pack_bf2(o[4], o[5]); u.w = pack_bf2(o[6], o[7]);
        *reinterpret_cast<uint4*>(df + i) = u;
    }
}

// d_f = d_a * gelu_new'(f) on a [T, C] matrix (row stride ld) with the column sums of the ROUNDED result (the fc1 bias gradient) as
// per-64-row partials: block (cb, rb) = rows 64 rb .. x columns 2048 cb ..; a thread owns 8 consecutive columns.  df may alias da.
__global__ __launch_bounds__(256) void dgelu_colsum_kernel(const bf16_t* da, const bf16_t* __restrict__ f, bf16_t* df, float* __restrict__ part,
                                                          int T, int C, int ld) {
    const int c0 = (blockIdx.x * 256 + threadIdx.x) * 8, r0 = blockIdx.y * 64;
    if (c0 >= C) return;
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    const int rend = min(r0 + 64, T);
#pragma unroll 4
    for (int r = r0; r < rend; ++r) {
        const int64_t o = (int64_t)r * ld + c0;
        const uint4 ua = *reinterpret_cast<const uint4*>(da + o);
        const uint4 ub = *reinterpret_cast<const uint4*>(f + o);
        const bf16_t* ea = reinterpret_cast<const bf16_t*>(&ua);
        const bf16_t* eb = reinterpret_cast<const bf16_t*>(&ub);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bf2f(ea[j]) * gelu_new_grad(bf2f(eb[j]));
        uint4 u;
        u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
        *reinterpret_cast<uint4*>(df + o) = u;
        const bf16_t* eo = reinterpret_cast<const bf16_t*>(&u);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += bf2f(eo[j]);
    }
    float* p = part + (int64_t)blockIdx.y * C + c0;
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = a[j];
}

}  // namespace

extern "C" int showo_transpose_bf16(const uint16_t* x, int ld, uint16_t* xt, int T, int C, int Tp, int mode, float* colpart,
                                    float* colsum, int accumulate, void* stream) {
    if (T <= 0 || C <= 0) return 0;
    if ((Tp % 64) || Tp < T || (ld % 8) || (C % 8)) return set_error_msg(1, "transpose: Tp % 64 == 0, Tp >= T, ld % 8 == 0, C % 8 == 0 required");
    if (colsum && !colpart) return set_error_msg(1, "transpose: column sums need the partial buffer [Tp/64, C]");
    hipStream_t s = (hipStream_t)stream;
    transpose_kernel<<<dim3((C + 63) / 64, Tp / 64), dim3(256), 0, s>>>(x, xt, colsum ? colpart : nullptr, T, C, ld, Tp, mode);
    if (colsum) colsum_reduce(colpart, colsum, Tp / 64, C, accumulate, s);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

// colsum[c] (+)= sum_t x[t][c] of a token-major bf16 matrix (bias gradients next to showo_gemm_tn_bf16).  colpart: fp32 scratch of
// (ceil(T / 32) + 8) * C floats.  Deterministic two-level sum.
extern "C" int showo_colsum_bf16(const uint16_t* x, int ld, int T, int C, float* colpart, float* colsum, int accumulate, void* stream) {
    if (T <= 0 || C <= 0) return 0;
    if (!x || !colpart || !colsum) return set_error_msg(1, "colsum: null argument");
    if ((ld % 8) || (((uintptr_t)x) & 15)) return set_error_msg(1, "colsum: x must be 16-byte aligned with ld a multiple of 8");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (T + 31) / 32;
    colsum_bf16_kernel<<<dim3((C + 2047) / 2048, nblk), dim3(256), 0, s>>>(x, colpart, T, C, ld);
    colsum_reduce(colpart, colsum, nblk, C, accumulate, s);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

static int ln_bwd_impl(const float* x, const float* gamma, const float* dh, const float* dy, float* dx32, uint16_t* dx16, float* part,
                       float* dgb, float* dxsum, int T, int H, float eps, void* stream) {
    if (T <= 0) return 0;
    if ((H % 4) || H > 2048) return set_error_msg(1, "ln_bwd: H must be a multiple of 4 and <= 2048");
    if (dxsum && !dx16) return set_error_msg(1, "ln_bwd: the column sums are those of dx16");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (T + LNB_ROWS - 1) / LNB_ROWS;  // the partial buffer holds nblk + 8 rows (showo_ln_bwd_blocks)
    static bool attr_set = false;
    if (!attr_set) {
        SHOWO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ln_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        SHOWO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ln_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        attr_set = true;
    }
    if (dxsum) {
        // part is [nblk][3][H] (dgamma | dbeta | column sums of dx16): ONE reduce, columns < 2H -> dgb, the rest -> dxsum
        ln_bwd_kernel<true><<<dim3(nblk), dim3(256), (size_t)8 * H * sizeof(float), s>>>(x, gamma, dh, dy, dx32, dx16, part, part + 2 * H, T, H, eps);
        colsum_reduce(part, dgb, nblk, 3 * H, 0, s, dxsum, 2 * H);
    } else {
        // part is [nblk][2][H]: reduce it as a [nblk, 2H] matrix -> dgb = (dgamma[H], dbeta[H])
        ln_bwd_kernel<false><<<dim3(nblk), dim3(256), (size_t)8 * H * sizeof(float), s>>>(x, gamma, dh, dy, dx32, dx16, part, nullptr, T, H, eps);
        colsum_reduce(part, dgb, nblk, 2 * H, 0, s);
    }
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
extern "C" int showo_ln_bwd(const float* x, const float* gamma, const float* dh, const float* dy, float* dx32, uint16_t* dx16,
                            float* part, float* dgb, int T, int H, float eps, void* stream) {
    return ln_bwd_impl(x, gamma, dh, dy, dx32, dx16, part, dgb, nullptr, T, H, eps, stream);
}
// the same, plus dxsum[H] = column sums of dx16 (the bias gradients of the projections that consume dx as their dY); part: scratch
// fp32 [showo_ln_bwd_blocks(T), 3, H]
extern "C" int showo_ln_bwd_colsum(const float* x, const float* gamma, const float* dh, const float* dy, float* dx32, uint16_t* dx16,
                                   float* part, float* dgb, float* dxsum, int T, int H, float eps, void* stream) {
    if (!dxsum) return set_error_msg(1, "ln_bwd_colsum: dxsum required");
    return ln_bwd_impl(x, gamma, dh, dy, dx32, dx16, part, dgb, dxsum, T, H, eps, stream);
}
extern "C" int showo_ln_bwd_blocks(int T) { return (T + LNB_ROWS - 1) / LNB_ROWS + 8; }  // + 8 rows of reduction scratch

extern "C" int showo_qkln_rope_bwd(const uint16_t* dq, const uint16_t* dk, int ldg, const uint16_t* qkv, const float* qw,
                                   const float* kw, const float* cos_tab, const float* sin_tab, uint16_t* dqkv, float* part,
                                   float* dparams, int T, int L, int nH, int rot, float eps, void* stream) {
    if (T <= 0) return 0;
    if (rot != 32) return set_error_msg(1, "qkln_rope_bwd: rotary_dim 32 only");
    if ((ldg % 8) || (((uintptr_t)dq) & 15) || (((uintptr_t)dk) & 15) || (((uintptr_t)qkv) & 15) || (((uintptr_t)dqkv) & 15))
        return set_error_msg(1, "qkln_rope_bwd: dq / dk / qkv / dqkv must be 16-byte aligned with ldg a multiple of 8 (16-byte row loads)");
    hipStream_t s = (hipStream_t)stream;
    const int64_t rows = (int64_t)T * nH;
    const int nblk = (int)((rows + 4 * QKB_ROWS - 1) / (4 * QKB_ROWS));
    qkln_rope_bwd_kernel<<<dim3(nblk), dim3(256), 0, s>>>(dq, dk, ldg, qkv, qw, kw, cos_tab, sin_tab, dqkv, part, T, L, nH, eps);
    colsum_reduce(part, dparams, nblk, 256, 0, s);  // (dqw, dqb, dkw, dkb) x 64
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
extern "C" int showo_qkln_rope_bwd_blocks(int T, int nH) { return (int)(((int64_t)T * nH + 4 * QKB_ROWS - 1) / (4 * QKB_ROWS)) + 8; }

extern "C" int showo_ce_loss(const float* logits, int ldl, const int64_t* labels, int B, int L, int V, int b_t2i, int b_lm,
                             int b_mmu, int max_seq_len, float g_t2i, float g_lm, float g_mmu, void* rows_ws, int* counts,
                             float* rowloss, uint16_t* dlogits, int ldd, float* losses, void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if (dlogits && ((ldd % 8) || ldd < V)) return set_error_msg(1, "ce_loss: ldd must be a multiple of 8 and >= V");
    hipStream_t s = (hipStream_t)stream;
    const int R = B * L;
    CeRow* rows = reinterpret_cast<CeRow*>(rows_ws);
    SHOWO_CHECK_HIP(hipMemsetAsync(counts, 0, 3 * sizeof(int), s));
    ce_rows_kernel<<<dim3((R + 255) / 256), dim3(256), 0, s>>>(labels, rows, counts, B, L, b_t2i, b_lm, b_mmu, max_seq_len);
    ce_kernel<<<dim3(R), dim3(256), 0, s>>>(logits, ldl, rows, counts, g_t2i, g_lm, g_mmu, dlogits, ldd, rowloss, V);
    if (losses) ce_finalize_kernel<<<dim3(1), dim3(256), 0, s>>>(rowloss, rows, counts, losses, R);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int showo_embed_bwd(const int64_t* ids, const float* dx, float* dE, int* order_ws, int T, int H, int V, void* stream) {
    if (T <= 0) return 0;
    if (H % 4) return set_error_msg(1, "embed_bwd: H % 4 == 0 required");
    (void)V;
    hipStream_t s = (hipStream_t)stream;
    int* order = order_ws;
    int* runstart = order_ws + T;
    embed_rank_kernel<<<dim3((T + 63) / 64), dim3(256), 0, s>>>(ids, order, runstart, T);
    embed_bwd_kernel<<<dim3(T), dim3(256), 0, s>>>(ids, order, runstart, dx, dE, T, H);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int showo_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                           float weight_decay, int step, void* stream) {
    if (n <= 0) return 0;
    if (step < 1) return set_error_msg(1, "adamw: step counts from 1");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    int blocks = (int)((n + 255) / 256);
    if (blocks > 65536) blocks = 65536;
    adamw_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

namespace showo {
int adamw_multi_launch(const AdamSeg* segs, const int* seg_of, const int64_t* start_of, int nchunks, float lr, float beta1, float beta2,
                       float eps, float weight_decay, int step, hipStream_t s) {
    if (nchunks <= 0) return 0;
    if (step < 1) return set_error_msg(1, "adamw: step counts from 1");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    adamw_multi_kernel<<<dim3(nchunks), dim3(256), 0, s>>>(segs, seg_of, start_of, lr, beta1, beta2, eps, weight_decay, bc1, bc2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error_hip(e, "adamw_multi launch", __FILE__, __LINE__);
    return 0;
}
}  // namespace showo

extern "C" int showo_scale_f32(float* x, int64_t n, float s, void* stream) {
    if (n <= 0) return 0;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 65536) blocks = 65536;
    scale_f32_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(x, n, s);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- bf16 gradient wire (data-parallel exchange, SURVEY §8e: 2.90 GB per rank per step instead of 5.79 GB fp32).
// pack: wire[i] = bf16(grad[i] * scale) (scale = 1 / world, applied BEFORE rounding so the summed wire is the mean);
// unpack: grad[i] = float(wire[i]).  4 elements per thread, 16-B loads / 8-B stores (and the converse).
namespace {
__global__ __launch_bounds__(256) void wire_pack_kernel(const float* __restrict__ g, bf16_t* __restrict__ w, int64_t n, float scale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(g + i);
            uint2 o;
            o.x = pack_bf2(v.x * scale, v.y * scale);
            o.y = pack_bf2(v.z * scale, v.w * scale);
            *reinterpret_cast<uint2*>(w + i) = o;
        } else {
            for (int64_t j = i; j < n; ++j) w[j] = f2bf(g[j] * scale);
        }
    }
}
__global__ __launch_bounds__(256) void wire_unpack_kernel(const bf16_t* __restrict__ w, float* __restrict__ g, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const uint2 v = *reinterpret_cast<const uint2*>(w + i);
            *reinterpret_cast<float4*>(g + i) = make_float4(bf2f((bf16_t)(v.x & 0xffff)), bf2f((bf16_t)(v.x >> 16)),
                                                            bf2f((bf16_t)(v.y & 0xffff)), bf2f((bf16_t)(v.y >> 16)));
        } else {
            for (int64_t j = i; j < n; ++j) g[j] = bf2f(w[j]);
        }
    }
}
}  // namespace

extern "C" int showo_grad_wire_pack(const float* grad, uint16_t* wire, int64_t n, float scale, void* stream) {
    if (n <= 0) return 0;
    if ((((uintptr_t)grad) & 15) || (((uintptr_t)wire) & 7)) return set_error_msg(1, "grad_wire_pack: grad must be 16B and wire 8B aligned");
    int64_t blocks = (n / 4 + 255) / 256 + 1;
    if (blocks > 256 * 32) blocks = 256 * 32;
    wire_pack_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(grad, wire, n, scale);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int showo_grad_wire_unpack(const uint16_t* wire, float* grad, int64_t n, void* stream) {
    if (n <= 0) return 0;
    if ((((uintptr_t)grad) & 15) || (((uintptr_t)wire) & 7)) return set_error_msg(1, "grad_wire_unpack: grad must be 16B and wire 8B aligned");
    int64_t blocks = (n / 4 + 255) / 256 + 1;
    if (blocks > 256 * 32) blocks = 256 * 32;
    wire_unpack_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(wire, grad, n);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int showo_dgelu_bf16(const uint16_t* da, const uint16_t* f, uint16_t* df, int64_t n, void* stream) {
    if (n <= 0) return 0;
    if (n % 8) return set_error_msg(1, "dgelu: n % 8 == 0 required");
    int blocks = (int)((n / 8 + 255) / 256);
    if (blocks > 65536) blocks = 65536;
    dgelu_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(da, f, df, n);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

// showo_dgelu_bf16 on a [T, C] matrix (row stride ld; C % 8 == 0, 16-byte aligned) fused with showo_colsum_bf16 of its result:
// colsum[c] = sum_t df[t][c] (the fc1 bias gradient).  colpart: fp32 scratch of (ceil(T / 64) + 8) * C floats.  df may alias da.
extern "C" int showo_dgelu_colsum_bf16(const uint16_t* da, const uint16_t* f, uint16_t* df, int ld, int T, int C, float* colpart,
                                       float* colsum, void* stream) {
    if (T <= 0 || C <= 0) return 0;
    if (!da || !f || !df || !colpart || !colsum) return set_error_msg(1, "dgelu_colsum: null argument");
    if ((C % 8) || (ld % 8) || ((((uintptr_t)da) | ((uintptr_t)f) | ((uintptr_t)df)) & 15))
        return set_error_msg(1, "dgelu_colsum: C, ld multiples of 8 and 16-byte aligned tensors required");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (T + 63) / 64;
    dgelu_colsum_kernel<<<dim3((C + 2047) / 2048, nblk), dim3(256), 0, s>>>(da, f, df, colpart, T, C, ld);
    colsum_reduce(colpart, colsum, nblk, C, 0, s);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int showo_gelu_bf16(const uint16_t* f, uint16_t* a, int64_t n, void* stream) {
    if (n <= 0) return 0;
    if (n % 8) return set_error_msg(1, "gelu: n % 8 == 0 required");
    int blocks = (int)((n / 8 + 255) / 256);
    if (blocks > 65536) blocks = 65536;
    gelu_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(f, a, n);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}


// g fp32 [n] (16-byte aligned): g *= min(max_norm / (||g||_2 + 1e-6), 1); ws = CLIP_PARTS doubles of scratch; out2 (device, 2 floats)
// receives the total norm and the coefficient.  No host synchronisation.
extern "C" int showo_grad_clip_norm(float* g, int64_t n, float max_norm, double* ws, float* out2, void* stream) {
    if (!g || !ws || !out2 || n <= 0 || (((uintptr_t)g) & 15)) return showo::set_error_msg(1, "grad_clip_norm: bad argument (g 16-byte aligned)");
    hipStream_t s = (hipStream_t)stream;
    sumsq_part_kernel<<<dim3(CLIP_PARTS), dim3(256), 0, s>>>(g, n, ws);
    clip_coef_kernel<<<dim3(1), dim3(256), 0, s>>>(ws, max_norm, out2);
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    scale_by_dev_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(g, n, out2 + 1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return showo::set_error_hip(e, "grad_clip_norm launch", __FILE__, __LINE__);
    return 0;
}
extern "C" int showo_grad_clip_ws_doubles(void) { return CLIP_PARTS; }
