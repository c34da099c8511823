// HBM-bound VQGAN kernels for gfx950 (channel-last activations):
//   GroupNorm(32, eps=1e-6) statistics + normalise(+swish) -> bf16   (reference models/common_modules.py:16-24)
//   thin 1x1/3x3 fp32 convolution for the 13-channel latent convs     (reference models/modeling_magvitv2.py:131-139, 360-362)
//   channel pad + cast fp32 -> bf16 (feeds the MFMA implicit-GEMM conv in gemm.hip)
#include "common.h"
#include "../../include/showo_hip.h"

using namespace showo;

namespace {

constexpr int GN_GROUPS = 32;
constexpr int GN_PIX_PER_BLOCK = 512;

// Deterministic GroupNorm statistics (no atomics: the same input gives the same bits on every run, which the
// sign-test quantizer downstream needs).  Pass 1: block (blk, b) reduces GN_PIX_PER_BLOCK pixels to 32 (sum, sumsq)
// pairs in double and writes them to part[b][blk][g][2].  Thread t owns channel quad f = t % (C/4) for all of its
// pixels, so its partial sums belong to exactly one group (C % 128 == 0  =>  channels-per-group % 4 == 0); every
// group is owned by exactly 8 threads of the block.  Pass 2 adds the per-block partials in block order.
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, double* __restrict__ part, int HW, int C) {
    __shared__ double ssum[256], ssq[256];
    const int tid = threadIdx.x, b = blockIdx.y, nblk = gridDim.x;
    const int quads = C >> 2, ppp = 256 / quads;  // pixels per pass
    const int f = tid % quads, pl = tid / quads;
    const int p0 = blockIdx.x * GN_PIX_PER_BLOCK;
    const int pend = min(p0 + GN_PIX_PER_BLOCK, HW);
    const float* xb = x + (int64_t)b * HW * C;
    double s = 0.0, q = 0.0;
    for (int p = p0 + pl; p < pend; p += ppp) {
        float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)p * C + f * 4);
        s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        q += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
    // slot layout: group-major, 8 owners per group in (pixel-lane, quad-in-group) order
    const int qpg = quads / GN_GROUPS;  // quads per group
    const int g = f / qpg;
    const int slot = g * 8 + pl * qpg + (f - g * qpg);
    ssum[slot] = s;
    ssq[slot] = q;
    __syncthreads();
    if (tid < GN_GROUPS) {
        double a = 0.0, c = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a += ssum[tid * 8 + i]; c += ssq[tid * 8 + i]; }
        double* o = part + (((int64_t)b * nblk + blockIdx.x) * GN_GROUPS + tid) * 2;
        o[0] = a;
        o[1] = c;
    }
}
// stats[b][t] = sum_k part[b][k][t], t = g * 2 + which.  256 threads: four quarters of the k range (k = q, q + 4, ..) are summed in
// parallel and combined in a fixed order ((q0 + q1) + (q2 + q3)): same bits on every run.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ part, double* __restrict__ stats, int nblk) {
    __shared__ double sq[4][GN_GROUPS * 2];
    const int b = blockIdx.x, t = threadIdx.x & 63, q = threadIdx.x >> 6;
    double a = 0.0;
    for (int k = q; k < nblk; k += 4) a += part[((int64_t)b * nblk + k) * (GN_GROUPS * 2) + t];
    sq[q][t] = a;
    __syncthreads();
    if (q == 0) stats[(int64_t)b * GN_GROUPS * 2 + t] = (sq[0][t] + sq[1][t]) + (sq[2][t] + sq[3][t]);
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       bf16_t* __restrict__ y, bf16_t* __restrict__ ylo, int HW, int C, float eps,
                                                       int do_swish) {
    __shared__ float smean[GN_GROUPS], srstd[GN_GROUPS];
    const int tid = threadIdx.x, b = blockIdx.y;
    if (tid < GN_GROUPS) {
        double n = (double)HW * (C / GN_GROUPS);
        double mean = stats[((int64_t)b * GN_GROUPS + tid) * 2] / n;
        double var = stats[((int64_t)b * GN_GROUPS + tid) * 2 + 1] / n - mean * mean;
        if (var < 0) var = 0;
        smean[tid] = (float)mean;
        srstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int quads = C >> 2, ppp = 256 / quads;
    const int f = tid % quads, pl = tid / quads;
    const int g = (f * 4) / (C / GN_GROUPS);
    const float mean = smean[g], rstd = srstd[g];
    const float4 ga = *reinterpret_cast<const float4*>(gamma + f * 4);
    const float4 be = *reinterpret_cast<const float4*>(beta + f * 4);
    const int p0 = blockIdx.x * GN_PIX_PER_BLOCK;
    const int pend = min(p0 + GN_PIX_PER_BLOCK, HW);
    const float* xb = x + (int64_t)b * HW * C;
    bf16_t* yb = y + (int64_t)b * HW * C;
    bf16_t* ylb = ylo ? ylo + (int64_t)b * HW * C : nullptr;
    for (int p = p0 + pl; p < pend; p += ppp) {
        float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)p * C + f * 4);
        float o[4] = {(v.x - mean) * rstd * ga.x + be.x, (v.y - mean) * rstd * ga.y + be.y,
                      (v.z - mean) * rstd * ga.z + be.z, (v.w - mean) * rstd * ga.w + be.w};
        if (do_swish) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = o[j] * __builtin_amdgcn_rcpf(1.0f + __expf(-o[j]));
        }
        uint2 pk;
        pk.x = pack_bf2(o[0], o[1]);
        pk.y = pack_bf2(o[2], o[3]);
        *reinterpret_cast<uint2*>(yb + (int64_t)p * C + f * 4) = pk;
        if (ylb) {  // split precision: lo = bf16(x - float(hi))
            uint2 pl2;
            pl2.x = pack_bf2(o[0] - bf2f((bf16_t)(pk.x & 0xffff)), o[1] - bf2f((bf16_t)(pk.x >> 16)));
            pl2.y = pack_bf2(o[2] - bf2f((bf16_t)(pk.y & 0xffff)), o[3] - bf2f((bf16_t)(pk.y >> 16)));
            *reinterpret_cast<uint2*>(ylb + (int64_t)p * C + f * 4) = pl2;
        }
    }
}

// thin direct conv (fp32): one thread per output element; used only for the 13->13 1x1 latent convs
__global__ void conv_small_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                  float* __restrict__ out, int H, int W, int Cin, int Cout, int ks, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int co = (int)(i % Cout);
    int64_t p = i / Cout;
    int ox = (int)(p % W);
    int64_t t = p / W;
    int oy = (int)(t % H);
    int64_t b = t / H;
    int pad = (ks - 1) / 2;
    float acc = bias ? bias[co] : 0.f;
    for (int ky = 0; ky < ks; ++ky) {
        int iy = oy + ky - pad;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < ks; ++kx) {
            int ix = ox + kx - pad;
            if (ix < 0 || ix >= W) continue;
            const float* xp = x + ((b * H + iy) * W + ix) * (int64_t)Cin;
            const float* wp = w + ((int64_t)(co * ks + ky) * ks + kx) * Cin;
            for (int ci = 0; ci < Cin; ++ci) acc += xp[ci] * wp[ci];
        }
    }
    out[i] = acc;
}

// fp32 [P, C] -> bf16 [P, Cpad] (zero padded channels)
__global__ void pad_cast_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, bf16_t* __restrict__ ylo, int C, int Cpad,
                                int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int c = (int)(i % Cpad);
    int64_t p = i / Cpad;
    float v = (c < C) ? x[p * C + c] : 0.f;
    bf16_t h = f2bf(v);
    y[i] = h;
    if (ylo) ylo[i] = f2bf(v - bf2f(h));
}

__global__ void lfq_unpack_nhwc_kernel(const int64_t* __restrict__ ids, float* __restrict__ zq, int C, int64_t total) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int64_t id = ids[t];
    float* zp = zq + t * C;
    for (int c = 0; c < C; ++c) zp[c] = ((id >> (C - 1 - c)) & 1) ? 1.0f : -1.0f;
}

}  // namespace

// [B, 32, 2] result + per-block partials: one per GN_PIX_PER_BLOCK pixels for showo_gn_stats, one per 256-pixel conv tile when the
// statistics come from the convolution epilogue (showo_conv3x3_bf16x3_gn) -- sized for the latter
extern "C" int showo_gn_stats_doubles(int B, int HW) {
    int nblk = (HW + 255) / 256;
    return B * GN_GROUPS * 2 * (1 + nblk);
}

// stats[b][g][2] = sum over k < nblk of part[b][k][g][2], in k order
extern "C" int showo_gn_finalize(const double* part, double* stats, int B, int nblk, void* stream) {
    if (B <= 0 || nblk <= 0) return 0;
    gn_finalize_kernel<<<dim3(B), dim3(256), 0, (hipStream_t)stream>>>(part, stats, nblk);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int showo_gn_stats(const float* x, double* stats, int B, int HW, int C, void* stream) {
    if (B <= 0 || HW <= 0) return 0;
    if ((C % 128) || C > 1024) return set_error_msg(1, "gn_stats: C must be a multiple of 128 and <= 1024");
    const int nblk = (HW + GN_PIX_PER_BLOCK - 1) / GN_PIX_PER_BLOCK;
    double* part = stats + (int64_t)B * GN_GROUPS * 2;  // per-block partials live behind the [B,32,2] result
    gn_partial_kernel<<<dim3(nblk, B), dim3(256), 0, (hipStream_t)stream>>>(x, part, HW, C);
    gn_finalize_kernel<<<dim3(B), dim3(256), 0, (hipStream_t)stream>>>(part, stats, nblk);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int showo_gn_apply(const float* x, const double* stats, const float* gamma, const float* beta, uint16_t* y,
                              uint16_t* ylo, int B, int HW, int C, float eps, int do_swish, void* stream) {
    if (B <= 0 || HW <= 0) return 0;
    if ((C % 128) || C > 1024) return set_error_msg(1, "gn_apply: C must be a multiple of 128 and <= 1024");
    gn_apply_kernel<<<dim3((HW + GN_PIX_PER_BLOCK - 1) / GN_PIX_PER_BLOCK, B), dim3(256), 0, (hipStream_t)stream>>>(
        x, stats, gamma, beta, y, ylo, HW, C, eps, do_swish);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int showo_conv_small_f32(const float* x, const float* w, const float* bias, float* out, int B, int H, int W,
                                    int Cin, int Cout, int ksize, void* stream) {
    int64_t total = (int64_t)B * H * W * Cout;
    if (total <= 0) return 0;
    if (ksize != 1 && ksize != 3) return set_error_msg(1, "conv_small: ksize must be 1 or 3");
    conv_small_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(x, w, bias, out, H, W, Cin,
                                                                                                  Cout, ksize, total);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int showo_pad_cast_bf16(const float* x, uint16_t* y, uint16_t* ylo, int64_t P, int C, int Cpad, void* stream) {
    int64_t total = P * Cpad;
    if (total <= 0) return 0;
    pad_cast_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(x, y, ylo, C, Cpad, total);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int showo_lfq_unpack_nhwc(const int64_t* ids, float* zq, int B, int C, int hw, void* stream) {
    int64_t total = (int64_t)B * hw;
    if (total <= 0) return 0;
    if (C < 1 || C > 62) return set_error_msg(1, "lfq: C must be in [1,62]");
    lfq_unpack_nhwc_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(ids, zq, C, total);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
