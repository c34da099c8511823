// Fused AR-decode layer for one new token (B = 1, L = 1): three launches per transformer layer instead of seven.
//
//   reference path per layer (models/phi.py:806-835 parallel block, 1 token against the KV cache, modeling_showo.py:190-240):
//     LN -> qkv projection -> q/k LayerNorm + RoPE + cache append -> attention -> dense(+x) ; LN -> fc1+gelu -> fc2(+x)
//   here:
//     ln_gemv2_kernel   x -> LayerNorm (in LDS, every block) -> qkv row (bf16) and gelu(fc1) row (bf16)       56 MB of weights
//     attn_decode_kernel<FUSED> (attention.hip): prep of the new token + cache append + single-query attention
//     out_gemv2_kernel  x += dense(attn) + bd ; x += fc2(ffn) + b2                                             40 MB of weights
//
// All three are HBM-streaming kernels: every weight byte is read exactly once per token (2.4 GB over 24 layers + 240 MB
// lm_head), a wave owns whole output columns and splits K over its lanes exactly like gemv_kernel (gemm.hip) -- same
// lane split, same accumulation order, same epilogue expressions -- so the fused step is bit-identical to the unfused one.
// The next 4 x 16-B weight loads of a wave are always in flight while the current ones are multiplied.
#include "common.h"
#include "engine.h"
#include "decode_common.h"
#include "../../include/showo_hip.h"

using namespace showo;

namespace {

struct LnGemvArgs {
    const float* x;     // [H] fp32 residual stream row
    const float *lnw, *lnb;
    float eps;
    int H;
    const bf16_t* W0;   // [N0, H]  -> out0 = bf16(W0 h + b0)   or, when outf != nullptr, outf = fp32(W0 h + b0)
    const float* b0;
    bf16_t* out0;
    float* outf;
    int N0;
    const bf16_t* W1;   // [N1, H]  -> out1 = bf16(gelu_new(W1 h + b1))
    const float* b1;
    bf16_t* out1;
    int N1;
};

// R = weight rows in flight per wave (2; see decode_ln_gemv2 for the measurement of 4); larger matrices (lm_head) loop and refill.
template <int R, bool F16 = false>
__global__ __launch_bounds__(256) void ln_gemv2_kernel(LnGemvArgs g) {  // H <= 2048: one 4-load group covers a weight row
    extern __shared__ bf16_t sh[];  // normalised row, bf16 like showo_layernorm_f32_bf16's output
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: weight-row bases in SGPRs
    const int H = g.H, Ntot = g.N0 + g.N1;
    const int stride = gridDim.x * 4;
    int n = blockIdx.x * 4 + wave;
    auto rowp = [&](int c) { return c < g.N0 ? g.W0 + (int64_t)c * H : g.W1 + (int64_t)(c - g.N0) * H; };
    // the first R weight rows of the wave are in flight before anything else: they do not depend on the LayerNorm
    uint4 br[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (n + r * stride < Ntot) load4(rowp(n + r * stride), lane * 8, H, br[r]);
    {   // every wave repeats layernorm_kernel's statistics (same lane split, same expressions => same bits) on a register
        // copy of the row (one global round trip instead of three), then writes a quarter of the normalised row to LDS.
        // (Measured: a 1024-thread block sharing one wave's statistics is slower, 23 us vs 19 us -- two barriers and a second
        // pass over x cost more than the redundant reductions.)
        float4 xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = lane * 4 + j * 256;
            xv[j] = i < H ? *reinterpret_cast<const float4*>(g.x + i) : make_float4(0.f, 0.f, 0.f, 0.f);
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
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + g.eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = lane * 4 + j * 256;
            if ((j & 3) == wave && i < H) {
                const float4 v = xv[j];
                const float4 w = *reinterpret_cast<const float4*>(g.lnw + i);
                const float4 bb = *reinterpret_cast<const float4*>(g.lnb + i);
                uint2 o;
                o.x = Op16<F16>::pack2((v.x - mean) * rstd * w.x + bb.x, (v.y - mean) * rstd * w.y + bb.y);
                o.y = Op16<F16>::pack2((v.z - mean) * rstd * w.z + bb.z, (v.w - mean) * rstd * w.w + bb.w);
                *reinterpret_cast<uint2*>(sh + i) = o;
            }
        }
    }
    __syncthreads();
    auto finish = [&](int c, float acc) {
        acc = wave_sum_swap(acc);
        if (lane == 0) {
            if (c < g.N0) {
                const float v = acc + g.b0[c];
                if (g.outf) g.outf[c] = v;
                else g.out0[c] = Op16<F16>::cvt(v);
            } else {
                g.out1[c - g.N0] = Op16<F16>::cvt(gelu_new_fast(acc + g.b1[c - g.N0]));
            }
        }
    };
    while (n < Ntot) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int c = n + r * stride;
            if (c < Ntot) {
                const float acc = fma4<F16>(br[r], sh, lane * 8, H, 0.f);
                if (c + R * stride < Ntot) load4(rowp(c + R * stride), lane * 8, H, br[r]);
                finish(c, acc);
            }
        }
        n += R * stride;
    }
}

// C = 2048-element chunks per output column (dense chunks first, then fc2 chunks), ALL of them in flight per wave: a wave
// streams 20 KB per column at the real shape and only 8 waves per CU exist, so depth is what hides the HBM latency.
// MODE 0: x[n] = (x[n] + (dense + bd)) + (fc2 + b2) in one launch.  The forked decode layer (engine.hip) splits it so that fc2 --
// which does not depend on the attention -- streams its 33.5 MB of weights WHILE the attention kernel runs:
//   MODE 1 (K0 = 0): y2[n] = fc2 + b2            (side stream, next to the attention kernel)
//   MODE 2 (K1 = 0): x[n] = (x[n] + (dense + bd)) + y2[n]   (after the join)
// Same lane split, accumulation order and parenthesisation in every mode: the three forms agree bit for bit.
template <int C, int MODE, bool F16 = false>
__global__ __launch_bounds__(512) void out_gemv2_kernel(OutGemvArgs g) {
    extern __shared__ bf16_t sa[];  // [K0] attention row, [K1] gelu(fc1) row: read once per block instead of once per wave
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: weight-row bases in SGPRs
    const int stride = gridDim.x * 8;
    const int c0 = (g.K0 + 2047) / 2048;
    int n = blockIdx.x * 8 + wave;
    auto issue = [&](int col, int t, uint4 (&wv)[4]) {
        if (t < c0) load4(g.W0 + (int64_t)col * g.K0, t * 2048 + lane * 8, g.K0, wv);
        else load4(g.W1 + (int64_t)col * g.K1, (t - c0) * 2048 + lane * 8, g.K1, wv);
    };
    uint4 buf[C][4];
    if (n < g.N) {
#pragma unroll
        for (int t = 0; t < C; ++t) issue(n, t, buf[t]);
    }
    for (int i = threadIdx.x * 8; i < g.K0 + g.K1; i += 512 * 8)
        *reinterpret_cast<uint4*>(sa + i) = i < g.K0 ? *reinterpret_cast<const uint4*>(g.a0 + i) : *reinterpret_cast<const uint4*>(g.a1 + (i - g.K0));
    __syncthreads();
    const bf16_t* a0 = sa;
    const bf16_t* a1 = sa + g.K0;
    while (n < g.N) {
        const int nn = n + stride;
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int t = 0; t < C; ++t) {
            if (t < c0) acc0 = fma4<F16>(buf[t], a0, t * 2048 + lane * 8, g.K0, acc0);
            else acc1 = fma4<F16>(buf[t], a1, (t - c0) * 2048 + lane * 8, g.K1, acc1);
            if (nn < g.N) issue(nn, t, buf[t]);
        }
        acc0 = wave_sum_swap(acc0);
        acc1 = wave_sum_swap(acc1);
        if (lane == 0) {
            if (MODE == 1) {
                g.y2[n] = acc1 + g.b1[n];
            } else {
                float v = acc0 + g.b0[n];   // x1 = x + (dense + bd)         (RESID epilogue order of gemv_kernel)
                v += g.x[n];
                float v2 = MODE == 2 ? g.y2[n] : acc1 + g.b1[n];  // x2 = x1 + (fc2 + b2)
                v2 += v;
                g.x[n] = v2;
            }
        }
        n = nn;
    }
}

int pick_blocks(int cols, int waves_per_block, int max_blocks) {  // strided column assignment: any grid covers the matrix
    const int b = (cols + waves_per_block - 1) / waves_per_block;
    return b < max_blocks ? b : max_blocks;
}

}  // namespace

namespace showo {

bool decode_fused_shapes_ok(int H, int F) { return (H % 8) == 0 && (F % 8) == 0 && H <= 2048 && F <= 8192; }

// x -> LN(lnw, lnb) -> { out0 = bf16(W0 h + b0) | outf = fp32(W0 h + b0) } and out1 = bf16(gelu(W1 h + b1))  (N1 may be 0)
int decode_ln_gemv2(const float* x, const float* lnw, const float* lnb, float eps, int H, const bf16_t* W0, const float* b0,
                    bf16_t* out0, float* outf, int N0, const bf16_t* W1, const float* b1, bf16_t* out1, int N1, hipStream_t s, int op) {
    LnGemvArgs g{x, lnw, lnb, eps, H, W0, b0, out0, outf, N0, W1, b1, out1, N1};
    // two weight rows in flight per wave (12 rows per block; a wave's third row is requested behind its first FMA).  Four rows -- the whole
    // layer matrix requested at kernel start -- measured SLOWER on cfg4 (0.983-0.992 vs 0.958-0.969 ms per token, round 2: 123 VGPRs halve
    // the resident waves and the single burst queues behind itself); that instance left the library in round 6.
    const dim3 grid(pick_blocks(N0 + N1, 12, showo::decode_tuning().ln_blocks));
    if (op) ln_gemv2_kernel<2, true><<<grid, dim3(256), (size_t)H * sizeof(bf16_t), s>>>(g);
    else ln_gemv2_kernel<2><<<grid, dim3(256), (size_t)H * sizeof(bf16_t), s>>>(g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error_hip(e, "ln_gemv2 launch", __FILE__, __LINE__);
    return 0;
}

// x[n] += (W0[n,:] a0 + b0[n]);  x[n] += (W1[n,:] a1 + b1[n])
int decode_out_gemv2(float* x, const bf16_t* W0, const bf16_t* a0, const float* b0, int K0, const bf16_t* W1, const bf16_t* a1,
                     const float* b1, int K1, int N, hipStream_t s, int mode, float* y2, int op) {
    OutGemvArgs g{x, W0, a0, b0, K0, W1, a1, b1, K1, N, y2};
    if (mode == 1) g.K0 = 0;
    if (mode == 2) g.K1 = 0;
    if (mode && !y2) return set_error_msg(1, "decode_out_gemv2: y2 required");
    const int C = (g.K0 + 2047) / 2048 + (g.K1 + 2047) / 2048;
    const dim3 grid(pick_blocks(N, 8, showo::decode_tuning().out_blocks));
    const size_t smem = (size_t)(g.K0 + g.K1) * sizeof(bf16_t);
    if (op) {  // IEEE-half operands (precision 2): the shapes of the fused layer at Phi-1.5's size
        if (mode == 1 && C == 4) out_gemv2_kernel<4, 1, true><<<grid, dim3(512), smem, s>>>(g);
        else if (mode == 1 && C >= 1 && C <= 3) out_gemv2_kernel<3, 1, true><<<grid, dim3(512), smem, s>>>(g);
        else if (mode == 2 && C == 1) out_gemv2_kernel<1, 2, true><<<grid, dim3(512), smem, s>>>(g);
        else if (mode != 0) return set_error_msg(1, "decode_out_gemv2: unsupported K0/K1 for the forked layer");
        else switch (C) {
            case 2: out_gemv2_kernel<2, 0, true><<<grid, dim3(512), smem, s>>>(g); break;
            case 3: out_gemv2_kernel<3, 0, true><<<grid, dim3(512), smem, s>>>(g); break;
            case 4: out_gemv2_kernel<4, 0, true><<<grid, dim3(512), smem, s>>>(g); break;
            case 5: out_gemv2_kernel<5, 0, true><<<grid, dim3(512), smem, s>>>(g); break;
            default: return set_error_msg(1, "decode_out_gemv2: unsupported K0/K1 (decode_fused_shapes_ok)");
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return set_error_hip(e, "out_gemv2 launch", __FILE__, __LINE__);
        return 0;
    }
    if (mode == 1 && C == 4) out_gemv2_kernel<4, 1><<<grid, dim3(512), smem, s>>>(g);
    else if (mode == 1 && C >= 1 && C <= 3) out_gemv2_kernel<3, 1><<<grid, dim3(512), smem, s>>>(g);
    else if (mode == 2 && C == 1) out_gemv2_kernel<1, 2><<<grid, dim3(512), smem, s>>>(g);
    else if (mode != 0) return set_error_msg(1, "decode_out_gemv2: unsupported K0/K1 for the forked layer");
    else switch (C) {
        case 2: out_gemv2_kernel<2, 0><<<grid, dim3(512), smem, s>>>(g); break;
        case 3: out_gemv2_kernel<3, 0><<<grid, dim3(512), smem, s>>>(g); break;
        case 4: out_gemv2_kernel<4, 0><<<grid, dim3(512), smem, s>>>(g); break;
        case 5: out_gemv2_kernel<5, 0><<<grid, dim3(512), smem, s>>>(g); break;
        default: return set_error_msg(1, "decode_out_gemv2: unsupported K0/K1 (decode_fused_shapes_ok)");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error_hip(e, "out_gemv2 launch", __FILE__, __LINE__);
    return 0;
}

}  // namespace showo
