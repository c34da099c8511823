// HBM-bound elementwise / reduction kernels of the Show-o hot path (gfx950).
//   LFQ sign-pack / unpack        (reference models/modeling_magvitv2.py:201-221, 239-241)
//   LayerNorm fp32 -> bf16        (reference models/phi.py:744, 776, 1065)
//   embedding gather, casts, row softmax, NCHW<->NHWC
#include "common.h"
#include "../../include/showo_hip.h"
#include <cfloat>

using namespace showo;

// ------------------------------------------------------------------------------------------------
// LFQ: 13 sign tests per token -> one int64 id.  52 B read + 8 B written per token (algorithmic bytes).
// NCHW: one thread per token, channel loop strided by hw: lanes of a wave read consecutive tokens of the
// same channel -> fully coalesced 256 B segments.
// ------------------------------------------------------------------------------------------------
__global__ void lfq_pack_nchw_kernel(const float* __restrict__ z, int64_t* __restrict__ ids, int C, int hw, int64_t total) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int64_t b = t / hw;
    int p = (int)(t - b * hw);
    const float* zp = z + b * (int64_t)C * hw + p;
    int64_t id = 0;
    for (int c = 0; c < C; ++c) id = (id << 1) | (zp[(int64_t)c * hw] > 0.0f ? 1 : 0);
    ids[t] = id;
}

// NHWC ([tokens][ldz] floats, the layout the encoder's last convolution leaves): a token's channels are ldz consecutive floats, so "one
// thread per token" makes every load instruction of a wave touch 64 different lines (2.6 TB/s measured, profiles/pmc/r4y_vq_hbm_rocprof.txt).
// Here a block owns 256 tokens = one contiguous run of 256 ldz floats: the threads stream it with consecutive 16-byte loads (every
// wave-instruction reads 1 KiB of consecutive bytes), park the 0/1 sign tests as bytes in LDS, and thread t then packs the C bytes of
// token t (MSB = channel 0, models/modeling_magvitv2.py:201-206) and stores ids[t]: consecutive 8-byte stores.
__global__ __launch_bounds__(256) void lfq_pack_nhwc_kernel(const float* __restrict__ z, int64_t* __restrict__ ids, int C, int ldz, int64_t total) {
    __shared__ __attribute__((aligned(16))) unsigned char bits[256 * 64];
    const int tid = threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.x * 256;
    const int nt = (int)(total - t0 < 256 ? total - t0 : 256);
    const int nf = nt * ldz;
    const float* zb = z + t0 * ldz;  // 256 ldz floats per block: 16-byte aligned whenever z is
    int done = 0;
    if ((((uintptr_t)zb) & 15) == 0) {
        const int nv = nf >> 2;
        for (int i = tid; i < nv; i += 256) {
            const float4 v = reinterpret_cast<const float4*>(zb)[i];
            reinterpret_cast<uint32_t*>(bits)[i] = (v.x > 0.0f ? 1u : 0u) | (v.y > 0.0f ? 0x100u : 0u) | (v.z > 0.0f ? 0x10000u : 0u) | (v.w > 0.0f ? 0x1000000u : 0u);
        }
        done = nv << 2;
    }
    for (int i = done + tid; i < nf; i += 256) bits[i] = zb[i] > 0.0f ? 1 : 0;
    __syncthreads();
    if (tid < nt) {
        const unsigned char* bp = bits + tid * ldz;
        int64_t id = 0;
        for (int c = 0; c < C; ++c) id = (id << 1) | bp[c];
        ids[t0 + tid] = id;
    }
}

__global__ void lfq_unpack_nchw_kernel(const int64_t* __restrict__ ids, float* __restrict__ zq, int C, int hw, int64_t total) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    int64_t b = t / hw;
    int p = (int)(t - b * hw);
    int64_t id = ids[t];
    float* zp = zq + b * (int64_t)C * hw + p;
    for (int c = 0; c < C; ++c) zp[(int64_t)c * hw] = ((id >> (C - 1 - c)) & 1) ? 1.0f : -1.0f;
}

extern "C" int showo_lfq_pack_nchw(const float* z, int64_t* ids, int B, int C, int hw, void* stream) {
    int64_t total = (int64_t)B * hw;
    if (total == 0) return 0;
    if (C < 1 || C > 62) return set_error_msg(1, "lfq: C must be in [1,62]");
    lfq_pack_nchw_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(z, ids, C, hw, total);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
extern "C" int showo_lfq_pack_nhwc(const float* z, int64_t* ids, int B, int C, int hw, int ldz, void* stream) {
    int64_t total = (int64_t)B * hw;
    if (total == 0) return 0;
    if (C < 1 || C > 62 || ldz < C || ldz > 64) return set_error_msg(1, "lfq: bad C/ldz (1 <= C <= 62, C <= ldz <= 64)");
    lfq_pack_nhwc_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(z, ids, C, ldz, total);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
extern "C" int showo_lfq_unpack_nchw(const int64_t* ids, float* zq, int B, int C, int hw, void* stream) {
    int64_t total = (int64_t)B * hw;
    if (total == 0) return 0;
    if (C < 1 || C > 62) return set_error_msg(1, "lfq: C must be in [1,62]");
    lfq_unpack_nchw_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(ids, zq, C, hw, total);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, float4 loads, two-pass (mean, then centred variance) in fp32.
// Row (8 KB at H=2048) stays in L1/L2 between passes; output bf16x4 stores.
// ------------------------------------------------------------------------------------------------
template <bool F16>  // F16: the output is IEEE half (common.h Op16)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, bf16_t* __restrict__ y,
                                                        const int32_t* __restrict__ row_index, int rows, int H, float eps) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    int64_t src = row_index ? (int64_t)row_index[r] : (int64_t)r;
    const float* xr = x + src * H;
    bf16_t* yr = y + (int64_t)r * H;
    const bool vec = (H & 3) == 0;
    if (vec && H <= 2048) {
        // the row lives in registers (8 float4 per lane at H = 2048): one pass over HBM instead of three trips through L1/L2;
        // same lane split and the same expressions as the general form below, hence the same bits
        float4 xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = lane * 4 + j * 256;
            xv[j] = i < H ? *reinterpret_cast<const float4*>(xr + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (lane * 4 + j * 256 < H) s += (xv[j].x + xv[j].y) + (xv[j].z + xv[j].w);
        const float mean = wave_sum(s) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (lane * 4 + j * 256 < H) {
                const float a = xv[j].x - mean, c = xv[j].y - mean, d = xv[j].z - mean, e = xv[j].w - mean;
                q += (a * a + c * c) + (d * d + e * e);
            }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = lane * 4 + j * 256;
            if (i < H) {
                const float4 v = xv[j];
                const float4 g = *reinterpret_cast<const float4*>(w + i);
                const float4 bb = *reinterpret_cast<const float4*>(b + i);
                uint2 o;
                o.x = Op16<F16>::pack2((v.x - mean) * rstd * g.x + bb.x, (v.y - mean) * rstd * g.y + bb.y);
                o.y = Op16<F16>::pack2((v.z - mean) * rstd * g.z + bb.z, (v.w - mean) * rstd * g.w + bb.w);
                *reinterpret_cast<uint2*>(yr + i) = o;
            }
        }
        return;
    }
    float s = 0.f;
    if (vec) {
        for (int i = lane * 4; i < H; i += 256) {
            float4 v = *reinterpret_cast<const float4*>(xr + i);
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int i = lane; i < H; i += 64) s += xr[i];
    }
    float mean = wave_sum(s) / (float)H;
    float q = 0.f;
    if (vec) {
        for (int i = lane * 4; i < H; i += 256) {
            float4 v = *reinterpret_cast<const float4*>(xr + i);
            float a = v.x - mean, c = v.y - mean, d = v.z - mean, e = v.w - mean;
            q += (a * a + c * c) + (d * d + e * e);
        }
    } else {
        for (int i = lane; i < H; i += 64) { float a = xr[i] - mean; q += a * a; }
    }
    float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
    if (vec) {
        for (int i = lane * 4; i < H; i += 256) {
            float4 v = *reinterpret_cast<const float4*>(xr + i);
            float4 g = *reinterpret_cast<const float4*>(w + i);
            float4 bb = *reinterpret_cast<const float4*>(b + i);
            uint2 o;
            o.x = Op16<F16>::pack2((v.x - mean) * rstd * g.x + bb.x, (v.y - mean) * rstd * g.y + bb.y);
            o.y = Op16<F16>::pack2((v.z - mean) * rstd * g.z + bb.z, (v.w - mean) * rstd * g.w + bb.w);
            *reinterpret_cast<uint2*>(yr + i) = o;
        }
    } else {
        for (int i = lane; i < H; i += 64) yr[i] = Op16<F16>::cvt((xr[i] - mean) * rstd * w[i] + b[i]);
    }
}

extern "C" int showo_layernorm_f32_op16(const float* x, const float* w, const float* b, uint16_t* y,
                                        const int32_t* row_index, int rows, int H, float eps, int op, void* stream) {
    if (rows <= 0) return 0;
    if (op == SHOWO_OP_F16) layernorm_kernel<true><<<dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>(x, w, b, y, row_index, rows, H, eps);
    else if (op == SHOWO_OP_BF16) layernorm_kernel<false><<<dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>(x, w, b, y, row_index, rows, H, eps);
    else return set_error_msg(1, "layernorm: op must be SHOWO_OP_BF16 or SHOWO_OP_F16");
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
extern "C" int showo_layernorm_f32_bf16(const float* x, const float* w, const float* b, uint16_t* y,
                                        const int32_t* row_index, int rows, int H, float eps, void* stream) {
    return showo_layernorm_f32_op16(x, w, b, y, row_index, rows, H, eps, SHOWO_OP_BF16, stream);
}

// ------------------------------------------------------------------------------------------------
template <bool F16>
__global__ void cast_f32_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        float4 v = *reinterpret_cast<const float4*>(s + i);
        uint2 o;
        o.x = Op16<F16>::pack2(v.x, v.y);
        o.y = Op16<F16>::pack2(v.z, v.w);
        *reinterpret_cast<uint2*>(d + i) = o;
    }
    // tail (n % 4) handled by the first threads of block 0
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        int64_t j = (n & ~(int64_t)3) + threadIdx.x;
        d[j] = Op16<F16>::cvt(s[j]);
    }
}
// range check of precision 2: elements that read as |x| >= 65504 (a saturated convert leaves exactly 65504) or NaN / inf in IEEE half
__global__ void count_f16_saturated_kernel(const bf16_t* __restrict__ x, int64_t n, unsigned long long* __restrict__ count) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned c = 0;
    for (; i < n; i += stride) c += (x[i] & 0x7fffu) >= 0x7bffu;  // 0x7bff = 65504; 0x7c00.. = inf / NaN
    c = (unsigned)wave_sum((float)c);  // < 2^24 per wave: exact in fp32
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, (unsigned long long)c);
}
extern "C" int showo_count_f16_saturated(const uint16_t* x, int64_t n, int64_t* count, void* stream) {
    if (n <= 0) return 0;
    if (!x || !count) return set_error_msg(1, "count_f16_saturated: null argument");
    int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
    if (blocks > 4096) blocks = 4096;
    count_f16_saturated_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(x, n, reinterpret_cast<unsigned long long*>(count));
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
// split: x = hi + lo with hi = bf16(x), lo = bf16(x - hi)
__global__ void split_f32_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float v = s[i];
        bf16_t h = f2bf(v);
        hi[i] = h;
        lo[i] = f2bf(v - bf2f(h));
    }
}
extern "C" int showo_split_f32_bf16(const float* src, uint16_t* hi, uint16_t* lo, int64_t n, void* stream) {
    if (n <= 0) return 0;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    split_f32_bf16_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(src, hi, lo, n);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int showo_cast_f32_op16(const float* src, uint16_t* dst, int64_t n, int op, void* stream) {
    if (n <= 0) return 0;
    if ((((uintptr_t)src) & 15) || (((uintptr_t)dst) & 7)) return set_error_msg(1, "cast: src must be 16B and dst 8B aligned");
    int64_t groups = (n + 3) / 4;
    int blocks = (int)((groups + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (op == SHOWO_OP_F16) cast_f32_bf16_kernel<true><<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(src, dst, n);
    else if (op == SHOWO_OP_BF16) cast_f32_bf16_kernel<false><<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(src, dst, n);
    else return set_error_msg(1, "cast: op must be SHOWO_OP_BF16 or SHOWO_OP_F16");
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
extern "C" int showo_cast_f32_bf16(const float* src, uint16_t* dst, int64_t n, void* stream) {
    return showo_cast_f32_op16(src, dst, n, SHOWO_OP_BF16, stream);
}

// embedding gather: one wave per token row, float4 copies.  Out-of-range ids poison the row with NaN so a
// bad id can never pass silently (the reference would raise an index error).
__global__ void embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table, float* __restrict__ x,
                             int T, int H, int V) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int t = blockIdx.x * 4 + wave;
    if (t >= T) return;
    int64_t id = ids[t];
    float* xr = x + (int64_t)t * H;
    if (id < 0 || id >= V) {
        for (int i = lane; i < H; i += 64) xr[i] = __builtin_nanf("");
        return;
    }
    const float* src = table + id * H;
    if ((H & 3) == 0) {
        for (int i = lane * 4; i < H; i += 256) *reinterpret_cast<float4*>(xr + i) = *reinterpret_cast<const float4*>(src + i);
    } else {
        for (int i = lane; i < H; i += 64) xr[i] = src[i];
    }
}
extern "C" int showo_embed_f32(const int64_t* ids, const float* table, float* x, int T, int H, int V, void* stream) {
    if (T <= 0) return 0;
    embed_kernel<<<dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>(ids, table, x, T, H, V);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

// row softmax (VQGAN AttnBlock): one wave per row; n <= a few thousand.
__global__ void softmax_rows_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, bf16_t* __restrict__ ylo, int rows, int n,
                                    int ldy, float scale) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const float* xr = x + (int64_t)r * n;
    float m = -FLT_MAX;
    for (int i = lane; i < n; i += 64) m = fmaxf(m, xr[i] * scale);
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += expf(xr[i] * scale - m);
    s = wave_sum(s);
    float inv = 1.0f / s;
    for (int i = lane; i < n; i += 64) {
        float p = expf(xr[i] * scale - m) * inv;
        bf16_t h = f2bf(p);
        y[(int64_t)r * ldy + i] = h;
        if (ylo) ylo[(int64_t)r * ldy + i] = f2bf(p - bf2f(h));
    }
    for (int i = n + lane; i < ldy; i += 64) {
        y[(int64_t)r * ldy + i] = 0;
        if (ylo) ylo[(int64_t)r * ldy + i] = 0;
    }
}
extern "C" int showo_softmax_rows_bf16(const float* x, uint16_t* y, uint16_t* ylo, int rows, int n, int ldy, float scale,
                                       void* stream) {
    if (rows <= 0) return 0;
    softmax_rows_kernel<<<dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>(x, y, ylo, rows, n, ldy, scale);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

// NCHW <-> NHWC fp32 (only at the image / latent boundary; C is 3 or 13 there, so a simple kernel suffices)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // index into NHWC output
    if (i >= total) return;
    int c = (int)(i % C);
    int64_t bp = i / C;
    int64_t b = bp / HW;
    int p = (int)(bp - b * HW);
    y[i] = x[(b * C + c) * (int64_t)HW + p];
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // index into NCHW output
    if (i >= total) return;
    int p = (int)(i % HW);
    int64_t bc = i / HW;
    int64_t b = bc / C;
    int c = (int)(bc - b * C);
    y[i] = x[(b * HW + p) * (int64_t)C + c];
}
extern "C" int showo_nchw_to_nhwc_f32(const float* x, float* y, int B, int C, int HW, void* stream) {
    int64_t total = (int64_t)B * C * HW;
    if (total == 0) return 0;
    nchw_to_nhwc_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(x, y, C, HW, total);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
extern "C" int showo_nhwc_to_nchw_f32(const float* x, float* y, int B, int C, int HW, void* stream) {
    int64_t total = (int64_t)B * C * HW;
    if (total == 0) return 0;
    nhwc_to_nchw_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(x, y, C, HW, total);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

// argmax (first maximal index) of a single fp32 vector — top_k=1 decode (modeling_showo.py:220-228)
__global__ void argmax_kernel(const float* __restrict__ x, int n, int64_t* __restrict__ out) {
    __shared__ float sv[16];
    __shared__ int si[16];
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float v = x[i];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int nw = blockDim.x >> 6;
        for (int w = 1; w < nw; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        out[0] = bi;
    }
}
// two-stage form for long rows (the 58 498 lm_head logits of a decode step): 64 blocks of 1024 threads find per-chunk
// winners, one wave merges them with the same "greater value, else smaller index" rule -- the same index as the single-block
// kernel, for every input.  One 1024-thread block took 23.8 us for 234 KB (a latency chain of 58 dependent loads per thread).
namespace {
constexpr int ARGMAX_PARTS = 64;
__device__ float g_argmax_val[ARGMAX_PARTS];
__device__ int g_argmax_idx[ARGMAX_PARTS];

__global__ __launch_bounds__(1024) void argmax_part_kernel(const float* __restrict__ x, int n) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = (n + ARGMAX_PARTS - 1) / ARGMAX_PARTS;
    const int lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lo + tid; i < hi; i += 1024) {
        float v = x[i];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        g_argmax_val[blockIdx.x] = best;
        g_argmax_idx[blockIdx.x] = bi;
    }
}
__global__ __launch_bounds__(64) void argmax_merge_kernel(int64_t* __restrict__ out) {
    const int lane = threadIdx.x;
    float best = g_argmax_val[lane];
    int bi = g_argmax_idx[lane];
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) *out = bi;
}
}  // namespace

// Token boundary of the greedy AR loop in ONE launch after argmax_part_kernel: merge the 64 partial arg-maxima -> token, append it
// to the output, advance the device-side position, write the next step's input embedding row and its mask row.  Replaces
// argmax_merge + store_token + step_inc + embed + decode_iv (five ~4.5 us launches per token in the per-token graph).
namespace {
__global__ __launch_bounds__(256) void greedy_token_seam_kernel(int64_t* __restrict__ tok, int64_t* __restrict__ out_tokens, int* __restrict__ pos,
                                                               int base, const float* __restrict__ table, float* __restrict__ x, int H, int V,
                                                               const int32_t* __restrict__ last_iv, int L0, int32_t* __restrict__ iv) {
    __shared__ int s_tok;
    const int tid = threadIdx.x;
    if (tid < 64) {  // argmax_merge_kernel
        float best = g_argmax_val[tid];
        int bi = g_argmax_idx[tid];
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o, 64);
            int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (tid == 0) {
            s_tok = bi;
            *tok = bi;
            const int P = *pos;
            out_tokens[P - base] = bi;  // store_token_kernel
            *pos = P + 1;               // step_inc_kernel
            // decode_iv_kernel for the next position: the last prompt row extended by the columns [L0, P + 1]
            int a = last_iv[0], b = last_iv[1], c = last_iv[2], d = last_iv[3];
            const int Pn = P + 1;
            if (b == L0 && a < b) b = Pn + 1;
            else if (d == L0 && c < d) d = Pn + 1;
            else if (!(c < d)) { c = L0; d = Pn + 1; }
            else if (!(a < b)) { a = L0; b = Pn + 1; }
            iv[0] = a; iv[1] = b; iv[2] = c; iv[3] = d;
        }
    }
    __syncthreads();
    const int id = s_tok;  // embed_kernel, one row
    if (id < 0 || id >= V) {
        for (int i = tid; i < H; i += 256) x[i] = __builtin_nanf("");
        return;
    }
    const float* src = table + (int64_t)id * H;
    if ((H & 3) == 0) {
        for (int i = tid * 4; i < H; i += 1024) *reinterpret_cast<float4*>(x + i) = *reinterpret_cast<const float4*>(src + i);
    } else {
        for (int i = tid; i < H; i += 256) x[i] = src[i];
    }
}
}  // namespace

namespace showo {
// returns -1 when the two-stage arg-max does not apply (n < 16384): the caller keeps the separate launches
int greedy_token_seam(const float* logits, int n, int64_t* tok, int64_t* out_tokens, int* pos, int base, const float* table, float* x, int H,
                      int V, const int32_t* last_iv, int L0, int32_t* iv, hipStream_t s) {
    if (n < 16384) return -1;
    argmax_part_kernel<<<dim3(ARGMAX_PARTS), dim3(1024), 0, s>>>(logits, n);
    greedy_token_seam_kernel<<<dim3(1), dim3(256), 0, s>>>(tok, out_tokens, pos, base, table, x, H, V, last_iv, L0, iv);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error_hip(e, "greedy_token_seam launch", __FILE__, __LINE__);
    return 0;
}
}  // namespace showo

extern "C" int showo_argmax_f32(const float* x, int n, int64_t* out, void* stream) {
    if (n <= 0) return set_error_msg(1, "argmax: n must be > 0");
    if (n >= 16384) {
        argmax_part_kernel<<<dim3(ARGMAX_PARTS), dim3(1024), 0, (hipStream_t)stream>>>(x, n);
        argmax_merge_kernel<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(out);
    } else {
        argmax_kernel<<<dim3(1), dim3(1024), 0, (hipStream_t)stream>>>(x, n, out);
    }
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- HBM ceiling anchor (tools/ceiling.py): float4 grid-stride copy, the measured counterpart of the 8 TB/s HBM3E spec
namespace {
__global__ __launch_bounds__(256) void copy_b128_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x * 4 + threadIdx.x; i < n16; i += stride) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i + u * 256 < n16) ? src[i + u * 256] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * 256 < n16) dst[i + u * 256] = v[u];
    }
}
}  // namespace

extern "C" int showo_copy_b128(const void* src, void* dst, int64_t nbytes, void* stream) {
    if (nbytes <= 0) return 0;
    if ((nbytes & 15) || (((uintptr_t)src) & 15) || (((uintptr_t)dst) & 15)) return showo::set_error_msg(1, "copy_b128: 16-byte alignment required");
    const int64_t n16 = nbytes >> 4;
    int64_t blocks = (n16 + 1023) / 1024;
    if (blocks > 256 * 16) blocks = 256 * 16;
    copy_b128_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>((const uint4*)src, (uint4*)dst, n16);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return showo::set_error_hip(e, "copy_b128 launch", __FILE__, __LINE__);
    return 0;
}



// ---- CU census (tests of showo_stream_create_cu_mask): where do the blocks of a launch on this stream run?
namespace {
__global__ __launch_bounds__(64) void cu_census_kernel(int32_t* __restrict__ ids, int spin) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the block alive for a while so that the launch spreads over every CU it is allowed on
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) ids[blockIdx.x] = (int32_t)(((xcc & 0xf) << 16) | (hw & 0xff00u));  // HW_ID: cu [11:8], sh [12], se [15:13]
}
}  // namespace
extern "C" int showo_cu_census(int32_t* ids, int blocks, int spin, void* stream) {
    if (!ids || blocks < 1) return showo::set_error_msg(1, "cu_census: bad argument");
    cu_census_kernel<<<dim3(blocks), dim3(64), 0, (hipStream_t)stream>>>(ids, spin);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return showo::set_error_hip(e, "cu_census launch", __FILE__, __LINE__);
    return 0;
}

// Probe of the wave reductions (tests): wave w of the launch reduces in[64 w .. 64 w + 63]; out[4 w + {0, 1, 2, 3}] = wave_sum,
// wave_sum_swap (permlane swaps + DPP), wave_max, wave_max_swap.  The VALU-only forms must give wave_sum's / wave_max's bits.
namespace {
__global__ __launch_bounds__(256) void wave_reduce_probe_kernel(const float* __restrict__ in, float* __restrict__ out, int n_waves) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n_waves) return;
    const float v = in[(int64_t)w * 64 + lane];
    const float a = showo::wave_sum(v), b = showo::wave_sum_swap(v), c = showo::wave_max(v), d = showo::wave_max_swap(v);
    // every lane must hold the result: report a lane that varies with w
    if (lane == (w & 63)) { out[4 * w] = a; out[4 * w + 1] = b; out[4 * w + 2] = c; out[4 * w + 3] = d; }
}
}  // namespace
extern "C" int showo_wave_reduce_probe(const float* in, float* out, int n_waves, void* stream) {
    if (!in || !out || n_waves < 1) return showo::set_error_msg(1, "wave_reduce_probe: bad argument");
    wave_reduce_probe_kernel<<<dim3((n_waves + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>(in, out, n_waves);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return showo::set_error_hip(e, "wave_reduce_probe launch", __FILE__, __LINE__);
    return 0;
}
