// Batched AR decode: NB independent sequences (BASELINE cfg4: "batch=4 images"), NB KV caches, ONE weight stream per token step.
//
// Reference: inference_mmu.py:87-177 walks the images one by one and Showo.mmu_generate (models/modeling_showo.py:183-240) is batch-1;
// a decode step is weight-streaming bound (2.66 GB of bf16 weights per token), so serving the NB sequences of a
// batch together reads every weight byte once per STEP instead of once per token.  The GEMVs become M = NB skinny GEMMs:
//
//   ln_gemvB_kernel<NB>   x [NB, H] -> LayerNorm (all NB rows in every block's LDS) -> qkv [NB, 3H] (bf16) and gelu(fc1) [NB, F] (bf16)
//   attn_decode_kernel<FUSED> (attention.hip), grid (heads, NB): prep + cache append + single-query attention, per-sequence position
//   out_gemvB_kernel<C, NB>  x[b] = (x[b] + (dense(attn[b]) + bd)) + (fc2(ffn[b]) + b2)
//   ln_gemvB_kernel<NB>   final LayerNorm + lm_head -> logits [NB, V] (fp32)
//   greedy_seam_rows_kernel: per sequence arg-max -> token, store, position + 1, next mask row, next embedding row
//
// A wave owns whole output columns and splits K over its lanes EXACTLY like the batch-1 kernels of decode.hip (same lane split, same
// accumulation order per sequence, same epilogue expressions), so every sequence gets the bits its batch-1 run gets: tokens AND
// logits of showo_engine_batch_decode_greedy equal NB separate showo_engine_decode_greedy runs (tests/test_decode_batch_gpu.py).
// The whole step replays as a hipGraph with the NB positions in device memory.
#include "engine.h"
#include "decode_common.h"
#include <cstdlib>

using namespace showo;

namespace showo {
int attn_decode_co_batch(const bf16_t* qkv, const float* qw, const float* qb, const float* kw, const float* kb, const float* cosT,
                         const float* sinT, bf16_t* K, bf16_t* Vt, const int32_t* iv, bf16_t* O, int B, int nH, int rot, float eps,
                         const int* pos_dev, int lk_max, int Lcap, int Lp, const OutGemvBArgs& fc2, int co_blocks, hipStream_t s,
                         const DecodePrefetch* pf, int op = 0);
int attn_decode_fused_batch(const bf16_t* qkv, const float* qw, const float* qb, const float* kw, const float* kb, const float* cosT,
                            const float* sinT, bf16_t* K, bf16_t* Vt, const int32_t* iv, bf16_t* O, int B, int nH, int rot, float eps,
                            const int* pos_dev, int lk_max, int Lcap, int Lp, hipStream_t s, int op = 0);
}

#define TRY(expr)            \
    do {                     \
        int _rc = (expr);    \
        if (_rc) return _rc; \
    } while (0)

namespace {

constexpr int MAXB = 8;

struct LnGemvBArgs {
    const float* x;  // [NB, H] fp32 residual rows
    const float *lnw, *lnb;
    float eps;
    int H;
    const bf16_t* W0;  // [N0, H] -> out0[b][n] = bf16(W0 h_b + b0) (ld0) or outf[b][n] = fp32(...) (ld0)
    const float* b0;
    bf16_t* out0;
    float* outf;
    int N0, ld0;
    const bf16_t* W1;  // [N1, H] -> out1[b][n] = bf16(gelu_new(W1 h_b + b1)) (ld1)
    const float* b1;
    bf16_t* out1;
    int N1, ld1;
    const bf16_t* W0lo = nullptr;  // HEAD3 instances (precision 2's lm_head): low halves of W0's rows, same layout
};

// REG (NB <= 4): after the LayerNorm every lane keeps ITS 32 activation values of each sequence in registers as 16 packed bf16 pairs
// (the k positions lane * 8 + u * 512 + j are the same for every weight row), so a weight row costs 16 NB v_dot2c_f32_bf16 per lane, no
// conversion and no LDS read -- with the activations re-read from LDS per row (the batch-1 form, !REG) four sequences made the step
// VALU / LDS-bound.  Fewer, longer-lived waves (2 blocks per CU) amortise the register fill (grid caps 256 / 1024 and 4 rows in flight
// measured slower: profiles/r5_decode_batch_sweep.txt).  Same products and the same accumulation order per sequence either way.
// F16: weights and 16-bit activations are IEEE half (precision 2; common.h Op16).  HEAD3 (precision 2's lm_head, fp32 logits only): the
// normalised row is kept as a (hi, lo) bf16 pair and the weights come as (hi, lo) bf16 rows (W0, W0lo): acc = hi.hi + lo.hi + hi.lo in
// ONE chain per sequence -- the split-bf16 product of engine.hip head_rows_precise_fast without its 3 H image (2 x 240 MB instead of
// 719 MB per token step) and without a separate LayerNorm launch.
template <int NB, bool REG, bool F16 = false, bool HEAD3 = false>
__global__ __launch_bounds__(256, 2) void ln_gemvB_kernel(LnGemvBArgs g) {  // H <= 2048; 2 weight rows in flight per wave (4: measured slower)
    static_assert(!(HEAD3 && (REG || F16)), "the split head keeps its activations in LDS and multiplies bf16 halves");
    constexpr int R = 2;
    extern __shared__ bf16_t sh[];  // [NB][H] normalised rows (16-bit, like showo_layernorm_f32_op16's output); HEAD3: [NB][2 H] = hi | lo
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: row bases live in SGPRs
    const int H = g.H, Ntot = g.N0 + g.N1;
    const int stride = gridDim.x * 4;
    int n = blockIdx.x * 4 + wave;
    auto rowp = [&](int c) { return c < g.N0 ? g.W0 + (int64_t)c * H : g.W1 + (int64_t)(c - g.N0) * H; };
    uint4 br[R][4], bl[HEAD3 ? R : 1][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (n + r * stride < Ntot) {
            load4(rowp(n + r * stride), lane * 8, H, br[r]);
            if constexpr (HEAD3) load4(g.W0lo + (int64_t)(n + r * stride) * H, lane * 8, H, bl[r]);
        }
    constexpr int LDH = HEAD3 ? 2 : 1;  // activation row stride in LDS, in units of H
    // LayerNorm: wave w normalises rows w, w + 4, ... with ln_gemv2_kernel's lane split and expressions (same bits per row)
    for (int b = wave; b < NB; b += 4) {
        const float* xr = g.x + (int64_t)b * H;
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
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + g.eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = lane * 4 + j * 256;
            if (i < H) {
                const float4 v = xv[j];
                const float4 w = *reinterpret_cast<const float4*>(g.lnw + i);
                const float4 bb = *reinterpret_cast<const float4*>(g.lnb + i);
                const float y0 = (v.x - mean) * rstd * w.x + bb.x, y1 = (v.y - mean) * rstd * w.y + bb.y;
                const float y2 = (v.z - mean) * rstd * w.z + bb.z, y3 = (v.w - mean) * rstd * w.w + bb.w;
                uint2 o;
                o.x = Op16<F16>::pack2(y0, y1);
                o.y = Op16<F16>::pack2(y2, y3);
                *reinterpret_cast<uint2*>(sh + b * LDH * H + i) = o;
                if constexpr (HEAD3) {  // low halves: y - bf16(y), rounded to bf16 (precise.hip ln_split3_kernel's expressions)
                    uint2 lo;
                    lo.x = pack_bf2(y0 - bf2f((bf16_t)(o.x & 0xffffu)), y1 - bf2f((bf16_t)(o.x >> 16)));
                    lo.y = pack_bf2(y2 - bf2f((bf16_t)(o.y & 0xffffu)), y3 - bf2f((bf16_t)(o.y >> 16)));
                    *reinterpret_cast<uint2*>(sh + b * LDH * H + H + i) = lo;
                }
            }
        }
    }
    __syncthreads();
    uint32_t act[REG ? NB : 1][16];
    if constexpr (REG) {
#pragma unroll
        for (int b = 0; b < NB; ++b) load_act_pairs(sh + b * H, lane * 8, H, act[b]);
    }
    while (n < Ntot) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int c = n + r * stride;
            if (c < Ntot) {
                float acc[NB];
                if constexpr (REG) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[b] = 0.f;
                    fma4_regs<NB, F16>(br[r], act, lane * 8, H, acc);
                } else if constexpr (HEAD3) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) {  // K' = [hi.hi | lo.hi | hi.lo], one chain (the order of the 3 H GEMV it replaces)
                        float t = fma4(br[r], sh + b * 2 * H, lane * 8, H, 0.f);
                        t = fma4(br[r], sh + b * 2 * H + H, lane * 8, H, t);
                        acc[b] = fma4(bl[r], sh + b * 2 * H, lane * 8, H, t);
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[b] = fma4<F16>(br[r], sh + b * H, lane * 8, H, 0.f);
                }
                if (c + R * stride < Ntot) {
                    load4(rowp(c + R * stride), lane * 8, H, br[r]);
                    if constexpr (HEAD3) load4(g.W0lo + (int64_t)(c + R * stride) * H, lane * 8, H, bl[r]);
                }
                if constexpr (NB <= 4) {  // lane 16 b finishes sequence b (wave_sum's order of additions per sequence: decode_common.h)
                    const float tot = wave_sum_groups<NB>(acc);
                    const int b = lane >> 4;
                    if ((lane & 15) == 0 && b < NB) {
                        if (c < g.N0) {
                            const float v = tot + g.b0[c];
                            if (g.outf) g.outf[(int64_t)b * g.ld0 + c] = v;
                            else g.out0[(int64_t)b * g.ld0 + c] = Op16<F16>::cvt(v);
                        } else {
                            g.out1[(int64_t)b * g.ld1 + (c - g.N0)] = Op16<F16>::cvt(gelu_new_fast(tot + g.b1[c - g.N0]));
                        }
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[b] = wave_sum(acc[b]);
                    if (lane == 0) {
                        if (c < g.N0) {
                            const float bias = g.b0[c];
#pragma unroll
                            for (int b = 0; b < NB; ++b) {
                                const float v = acc[b] + bias;
                                if (g.outf) g.outf[(int64_t)b * g.ld0 + c] = v;
                                else g.out0[(int64_t)b * g.ld0 + c] = Op16<F16>::cvt(v);
                            }
                        } else {
                            const float bias = g.b1[c - g.N0];
#pragma unroll
                            for (int b = 0; b < NB; ++b) g.out1[(int64_t)b * g.ld1 + (c - g.N0)] = Op16<F16>::cvt(gelu_new_fast(acc[b] + bias));
                        }
                    }
                }
            }
        }
        n += R * stride;
    }
}

// C = 2048-element chunks per output column (dense chunks first, then fc2 chunks), all in flight per wave (out_gemv2_kernel<C, 0>)
template <int C, int NB, bool F16 = false>
__global__ __launch_bounds__(512) void out_gemvB_kernel(OutGemvBArgs g) {
    extern __shared__ bf16_t sa[];  // [NB][K0 + K1]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: row bases live in SGPRs
    const int stride = gridDim.x * 8;
    const int c0 = (g.K0 + 2047) / 2048;
    const int KK = g.K0 + g.K1;
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
    for (int b = 0; b < NB; ++b)
        for (int i = threadIdx.x * 8; i < KK; i += 512 * 8)
            *reinterpret_cast<uint4*>(sa + b * KK + i) = i < g.K0 ? *reinterpret_cast<const uint4*>(g.a0 + (int64_t)b * g.lda0 + i)
                                                                  : *reinterpret_cast<const uint4*>(g.a1 + (int64_t)b * g.lda1 + (i - g.K0));
    __syncthreads();
    while (n < g.N) {
        const int nn = n + stride;
        float acc0[NB], acc1[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) { acc0[b] = 0.f; acc1[b] = 0.f; }
#pragma unroll
        for (int t = 0; t < C; ++t) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (t < c0) acc0[b] = fma4<F16>(buf[t], sa + b * KK, t * 2048 + lane * 8, g.K0, acc0[b]);
                else acc1[b] = fma4<F16>(buf[t], sa + b * KK + g.K0, (t - c0) * 2048 + lane * 8, g.K1, acc1[b]);
            }
            if (nn < g.N) issue(nn, t, buf[t]);
        }
        if constexpr (NB <= 4) {
            const float t0 = wave_sum_groups<NB>(acc0), t1 = wave_sum_groups<NB>(acc1);
            if ((lane & 15) == 0 && (lane >> 4) < NB) {
                const int64_t i = (int64_t)(lane >> 4) * g.N + n;
                float v = t0 + g.b0[n];  // x1 = x + (dense + bd)       (out_gemv2_kernel's order and parenthesisation)
                v += g.x[i];
                float v2 = t1 + g.b1[n];  // x2 = x1 + (fc2 + b2)
                v2 += v;
                g.x[i] = v2;
            }
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b) { acc0[b] = wave_sum(acc0[b]); acc1[b] = wave_sum(acc1[b]); }
            if (lane == 0) {
                const float bd = g.b0[n], b2 = g.b1[n];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    float v = acc0[b] + bd;
                    v += g.x[(int64_t)b * g.N + n];
                    float v2 = acc1[b] + b2;
                    v2 += v;
                    g.x[(int64_t)b * g.N + n] = v2;
                }
            }
        }
        n = nn;
    }
}

// Third launch of the co-scheduled batched layer: x[b][n] = (x[b][n] + (dense(attn[b]) + bd)) + y2[b][n]  (out_gemv2_kernel<1, 2>'s
// expression; y2 = fc2 + b2 from the fc2 role of the attention launch).  K0 <= 2048: the lane's 32 attention values per sequence stay
// in registers.
template <int NB, bool F16 = false>
__global__ __launch_bounds__(512) void out_dense_y2B_kernel(OutGemvBArgs g) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: row bases live in SGPRs
    const int stride = gridDim.x * 8;
    int n = blockIdx.x * 8 + wave;
    uint4 buf[4];
    if (n < g.N) load4(g.W0 + (int64_t)n * g.K0, lane * 8, g.K0, buf);
    uint32_t act[NB][16];
#pragma unroll
    for (int b = 0; b < NB; ++b) load_act_pairs(g.a0 + (int64_t)b * g.lda0, lane * 8, g.K0, act[b]);
    while (n < g.N) {
        const int nn = n + stride;
        // epilogue operands of this column requested up front (they do not depend on the products): no dependent round trip at the tail
        const bool writer = (lane & 15) == 0 && (lane >> 4) < NB;
        const int64_t i = (int64_t)(lane >> 4) * g.N + n;
        float xv = 0.f, yv = 0.f, bd = 0.f;
        if (writer) { xv = g.x[i]; yv = g.y2[i]; bd = g.b0[n]; }
        float acc0[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc0[b] = 0.f;
        fma4_regs<NB, F16>(buf, act, lane * 8, g.K0, acc0);
        if (nn < g.N) load4(g.W0 + (int64_t)nn * g.K0, lane * 8, g.K0, buf);
        const float tot = wave_sum_groups<NB>(acc0);
        if (writer) {
            float v = tot + bd;
            v += xv;
            float v2 = yv;
            v2 += v;
            g.x[i] = v2;
        }
        n = nn;
    }
}

// mask row of the token at position P: the last prompt row extended by the columns [L0, P] (modeling_showo.py:203-217)
__device__ __forceinline__ void next_iv(const int32_t* last_iv, int L0, int P, int32_t* iv) {
    int a = last_iv[0], b = last_iv[1], c = last_iv[2], d = last_iv[3];
    if (b == L0 && a < b) b = P + 1;
    else if (d == L0 && c < d) d = P + 1;
    else if (!(c < d)) { c = L0; d = P + 1; }
    else if (!(a < b)) { a = L0; b = P + 1; }
    iv[0] = a; iv[1] = b; iv[2] = c; iv[3] = d;
}
__global__ void batch_iv_kernel(const int32_t* __restrict__ last_iv, const int* __restrict__ L0, const int* __restrict__ pos, int32_t* __restrict__ iv,
                                int nb) {
    const int b = threadIdx.x;
    if (b < nb) next_iv(last_iv + 4 * b, L0[b], pos[b], iv + 4 * b);
}

// Token boundary of the batched greedy loop, one block per sequence: arg-max of its logits row (first maximal index, like
// showo_argmax_f32) -> token, append to its output row, position + 1, next mask row, next embedding row.
__global__ __launch_bounds__(1024) void greedy_seam_rows_kernel(const float* __restrict__ logits, int V, int64_t* __restrict__ tok,
                                                                int64_t* __restrict__ out_tokens, int n_steps, int* __restrict__ pos,
                                                                const int* __restrict__ base, const float* __restrict__ table,
                                                                float* __restrict__ x, int H, const int32_t* __restrict__ last_iv,
                                                                const int* __restrict__ L0, int32_t* __restrict__ iv) {
    __shared__ float sv[16];
    __shared__ int si[16];
    __shared__ int s_tok;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = logits + (int64_t)b * V;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 1024) {
        const float v = row[i];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        s_tok = bi;
        tok[b] = bi;
        const int P = pos[b];
        out_tokens[(int64_t)b * n_steps + (P - base[b])] = bi;
        pos[b] = P + 1;
        next_iv(last_iv + 4 * b, L0[b], P + 1, iv + 4 * b);
    }
    __syncthreads();
    const int id = s_tok;
    float* xr = x + (int64_t)b * H;
    if (id < 0 || id >= V) {
        for (int i = tid; i < H; i += 1024) xr[i] = __builtin_nanf("");
        return;
    }
    const float* src = table + (int64_t)id * H;
    for (int i = tid; i < H; i += 1024) xr[i] = src[i];
}

template <int NB>
int launch_ln_gemvB(const LnGemvBArgs& g, hipStream_t s, int op, bool head3) {
    const int Ntot = g.N0 + g.N1;
    if (head3) {  // precision 2's lm_head: (hi, lo) activations in LDS, (hi, lo) bf16 weight rows
        if (!g.W0lo || !g.outf || g.N1) return set_error_msg(1, "ln_gemvB: the split head writes fp32 logits from (hi, lo) weight rows");
        int blocks = (Ntot + 11) / 12;
        if (blocks > 1280) blocks = 1280;
        ln_gemvB_kernel<NB, false, false, true><<<dim3(blocks), dim3(256), (size_t)NB * 2 * g.H * sizeof(bf16_t), s>>>(g);
    } else if (NB <= 4) {  // register-resident activations (16 NB packed registers per lane): up to 4 blocks per CU resident
        int blocks = (Ntot + 7) / 8;
        const int cap = showo::decode_tuning().batch_ln_blocks;
        if (blocks > cap) blocks = cap;
        if (op) ln_gemvB_kernel<NB, (NB <= 4), true><<<dim3(blocks), dim3(256), (size_t)NB * g.H * sizeof(bf16_t), s>>>(g);
        else ln_gemvB_kernel<NB, (NB <= 4)><<<dim3(blocks), dim3(256), (size_t)NB * g.H * sizeof(bf16_t), s>>>(g);
    } else {
        int blocks = (Ntot + 11) / 12;
        if (blocks > 1280) blocks = 1280;
        if (op) ln_gemvB_kernel<NB, false, true><<<dim3(blocks), dim3(256), (size_t)NB * g.H * sizeof(bf16_t), s>>>(g);
        else ln_gemvB_kernel<NB, false><<<dim3(blocks), dim3(256), (size_t)NB * g.H * sizeof(bf16_t), s>>>(g);
    }
    return hipGetLastError() == hipSuccess ? 0 : set_error_msg(7, "ln_gemvB launch failed");
}
int ln_gemvB(int nb, const LnGemvBArgs& g, hipStream_t s, int op = 0, bool head3 = false) {
    switch (nb) {
        case 1: return launch_ln_gemvB<1>(g, s, op, head3);
        case 2: return launch_ln_gemvB<2>(g, s, op, head3);
        case 3: return launch_ln_gemvB<3>(g, s, op, head3);
        case 4: return launch_ln_gemvB<4>(g, s, op, head3);
        case 5: return launch_ln_gemvB<5>(g, s, op, head3);
        case 6: return launch_ln_gemvB<6>(g, s, op, head3);
        case 7: return launch_ln_gemvB<7>(g, s, op, head3);
        case 8: return launch_ln_gemvB<8>(g, s, op, head3);
    }
    return set_error_msg(1, "batched decode: 1..8 sequences");
}

template <int C, int NB, bool F16>
int launch_out_gemvB(const OutGemvBArgs& g, hipStream_t s) {
    static bool attr_set = false;
    const size_t smem = (size_t)NB * (g.K0 + g.K1) * sizeof(bf16_t);
    auto kfn = out_gemvB_kernel<C, NB, F16>;
    if (!attr_set && smem > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return set_error_hip(e, "hipFuncSetAttribute(out_gemvB)", __FILE__, __LINE__);
        attr_set = true;
    }
    if (smem > 160 * 1024) return set_error_msg(5, "batched decode: activations exceed the LDS");
    int blocks = (g.N + 7) / 8;
    if (blocks > showo::decode_tuning().out_blocks) blocks = showo::decode_tuning().out_blocks;
    kfn<<<dim3(blocks), dim3(512), smem, s>>>(g);
    return hipGetLastError() == hipSuccess ? 0 : set_error_msg(7, "out_gemvB launch failed");
}
template <int NB>
int out_gemvB_c(const OutGemvBArgs& g, hipStream_t s, int op) {
    const int C = (g.K0 + 2047) / 2048 + (g.K1 + 2047) / 2048;
    switch (C) {
        case 2: return op ? launch_out_gemvB<2, NB, true>(g, s) : launch_out_gemvB<2, NB, false>(g, s);
        case 3: return op ? launch_out_gemvB<3, NB, true>(g, s) : launch_out_gemvB<3, NB, false>(g, s);
        case 4: return op ? launch_out_gemvB<4, NB, true>(g, s) : launch_out_gemvB<4, NB, false>(g, s);
        case 5: return op ? launch_out_gemvB<5, NB, true>(g, s) : launch_out_gemvB<5, NB, false>(g, s);
    }
    return set_error_msg(1, "batched decode: unsupported K0 / K1 (decode_fused_shapes_ok)");
}
int out_dense_y2B(int nb, const OutGemvBArgs& g, hipStream_t s, int op) {
    int blocks = (g.N + 7) / 8;
    if (blocks > showo::decode_tuning().out_blocks) blocks = showo::decode_tuning().out_blocks;
    switch (nb * 2 + (op ? 1 : 0)) {
        case 4: out_dense_y2B_kernel<2><<<dim3(blocks), dim3(512), 0, s>>>(g); break;
        case 6: out_dense_y2B_kernel<3><<<dim3(blocks), dim3(512), 0, s>>>(g); break;
        case 8: out_dense_y2B_kernel<4><<<dim3(blocks), dim3(512), 0, s>>>(g); break;
        case 5: out_dense_y2B_kernel<2, true><<<dim3(blocks), dim3(512), 0, s>>>(g); break;
        case 7: out_dense_y2B_kernel<3, true><<<dim3(blocks), dim3(512), 0, s>>>(g); break;
        case 9: out_dense_y2B_kernel<4, true><<<dim3(blocks), dim3(512), 0, s>>>(g); break;
        default: return set_error_msg(1, "batched decode: the co-scheduled layer serves 2..4 sequences");
    }
    return hipGetLastError() == hipSuccess ? 0 : set_error_msg(7, "out_dense_y2B launch failed");
}
int out_gemvB(int nb, const OutGemvBArgs& g, hipStream_t s, int op) {
    switch (nb) {
        case 1: return out_gemvB_c<1>(g, s, op);
        case 2: return out_gemvB_c<2>(g, s, op);
        case 3: return out_gemvB_c<3>(g, s, op);
        case 4: return out_gemvB_c<4>(g, s, op);
        case 5: return out_gemvB_c<5>(g, s, op);
        case 6: return out_gemvB_c<6>(g, s, op);
        case 7: return out_gemvB_c<7>(g, s, op);
        case 8: return out_gemvB_c<8>(g, s, op);
    }
    return set_error_msg(1, "batched decode: 1..8 sequences");
}

}  // namespace

namespace showo {
// LayerNorm + lm_head of `nb` residual rows as the split-bf16 product (precision 2's head on a decode step): logits[b][n] = fp32,
// row stride ld.  One launch, 2 x [N, H] bf16 of weights (ln_gemvB_kernel<.., HEAD3>); the batched loop and the batch-1 step share it.
int decode_split_head(const float* x, const float* lnw, const float* lnb, float eps, int H, const bf16_t* Whi, const bf16_t* Wlo,
                      const float* bias, float* logits, int N, int ld, int nb, hipStream_t s) {
    LnGemvBArgs h{x, lnw, lnb, eps, H, Whi, bias, nullptr, logits, N, ld, nullptr, nullptr, nullptr, 0, 0};
    h.W0lo = Wlo;
    return ln_gemvB(nb, h, s, 0, true);
}
}  // namespace showo

// ---- engine entry points ------------------------------------------------------------------------------------------------------
// State of a decode batch: per-layer caches [nb][heads][cap][64] (K) / [nb][heads][64][cap] (V^T), per-sequence prompt length,
// live length and last prompt mask row; device copies for the graph-replayed loop.
struct showo_engine::BatchDecode {
    int nb = 0, cap = 0;
    bf16_t *k = nullptr, *vt = nullptr;
    int64_t elems = 0;  // allocated elements per cache
    int prompt_len[MAXB] = {0}, cache_len[MAXB] = {0};
    int last_iv[MAXB][4] = {{0}};
    int *pos_dev = nullptr, *L0_dev = nullptr, *base_dev = nullptr;
    int32_t *last_iv_dev = nullptr, *iv_dev = nullptr;
    float* y2 = nullptr;  // [nb, H] fc2 + b2 of the current layer (co-scheduled form)
    int precision = 0;    // the precision the caches were prefilled under (element type of K / V^T)
};

namespace showo {
void engine_batch_free(showo_engine* e) {
    if (e && e->bd) { delete e->bd; e->bd = nullptr; }  // device buffers live in e->allocs
}
}  // namespace showo

extern "C" int showo_engine_batch_begin(showo_engine* e, int nb, int cap_tokens) {
    if (!e) return set_error_msg(1, "engine: null handle");
    if (nb < 1 || nb > MAXB) return set_error_msg(1, "batch_begin: 1..8 sequences");
    if (!decode_fused_shapes_ok(e->H, e->F)) return set_error_msg(1, "batch_begin: hidden <= 2048 and ffn <= 8192 (multiples of 8) required");
    if (nb > e->maxT) return set_error_msg(5, "batch_begin: more sequences than workspace rows");
    int cap = ((cap_tokens + 63) / 64) * 64;
    if (cap_tokens < 2 || cap > ((e->cfg.max_pos + 63) / 64) * 64) return set_error_msg(5, "batch_begin: capacity exceeds max_position_embeddings");
    if ((size_t)cap * 4 + 2048 > 60000) return set_error_msg(5, "batch_begin: capacity exceeds the single-block decode attention");
    if (!e->bd) e->bd = new showo_engine::BatchDecode();
    auto* d = e->bd;
    const int64_t need = (int64_t)e->nL * nb * e->nH * cap * 64;
    if (need > d->elems) {  // grow: the previous caches are freed first (nothing of an earlier batch survives a batch_begin)
        SHOWO_CHECK_HIP(hipDeviceSynchronize());
        e->release(&d->k);
        e->release(&d->vt);
        d->elems = 0;
        TRY(e->alloc(&d->k, need));
        TRY(e->alloc(&d->vt, need));
        d->elems = need;
        // zeroed ONCE, when the caches are (re)allocated (the two memsets -- 600 MB at cfg4 -- ran on every call): columns of
        // V^T beyond a sequence's live length only ever meet probabilities that are exactly 0, so they must be FINITE, not zero, and
        // whatever an earlier batch left there is (a cache row is written by prefill / decode steps of finite activations)
        SHOWO_CHECK_HIP(hipMemset(d->k, 0, (size_t)need * sizeof(bf16_t)));
        SHOWO_CHECK_HIP(hipMemset(d->vt, 0, (size_t)need * sizeof(bf16_t)));
    }
    if (!d->pos_dev) {
        TRY(e->alloc(&d->pos_dev, MAXB)); TRY(e->alloc(&d->L0_dev, MAXB)); TRY(e->alloc(&d->base_dev, MAXB));
        TRY(e->alloc(&d->last_iv_dev, 4 * MAXB)); TRY(e->alloc(&d->iv_dev, 4 * MAXB));
    }
    d->nb = nb; d->cap = cap;
    for (int b = 0; b < MAXB; ++b) { d->prompt_len[b] = 0; d->cache_len[b] = 0; }
    return 0;
}

namespace showo {
int engine_prefill_into(showo_engine* e, const int64_t* ids, const float* embeds, const float* mask, int L, bf16_t* k, bf16_t* vt,
                        int64_t k_lstride, int64_t v_lstride, int cap, int* last_iv_out, float* logits_last, hipStream_t s,
                        bf16_t* k_lo, bf16_t* vt_lo);
}

extern "C" int showo_engine_batch_prefill(showo_engine* e, int b, const int64_t* ids, const float* embeds, const float* mask, int L,
                                          float* logits_last, void* stream) {
    if (!e || !e->bd || e->bd->nb == 0) return set_error_msg(1, "batch_prefill: showo_engine_batch_begin first");
    auto* d = e->bd;
    if (b < 0 || b >= d->nb) return set_error_msg(1, "batch_prefill: bad sequence index");
    if (L + 1 > d->cap) return set_error_msg(5, "batch_prefill: prompt exceeds the batch's cache capacity");
    if (showo_engine_missing(e) != 0) return set_error_msg(4, "engine: weights missing (showo_engine_missing() != 0)");
    if (L < 1 || L > e->maxT || L > e->cfg.max_seq || L > e->cfg.max_pos) return set_error_msg(5, "engine: sequence exceeds the configured workspace");
    if (e->precision == 1) return set_error_msg(1, "batch_prefill: the batched decode runs at precision 0 (bf16 operands) or 2 (fp16 operands)");
    const int64_t per_seq = (int64_t)e->nH * d->cap * 64, lstride = (int64_t)d->nb * per_seq;
    TRY(showo::engine_prefill_into(e, ids, embeds, mask, L, d->k + b * per_seq, d->vt + b * per_seq, lstride, lstride, d->cap, d->last_iv[b],
                                   logits_last, (hipStream_t)stream, nullptr, nullptr));
    d->prompt_len[b] = L;
    d->cache_len[b] = L;
    d->precision = e->precision;
    return 0;
}

// tok int64 [nb] (device): in = the token each sequence feeds first (e.g. the arg-max of its prefill logits), out = the last tokens
// produced; out_tokens int64 [nb, n_steps] (device); logits_ws fp32 [nb, vocab] (device).  Every sequence advances n_steps tokens
// (the caller cuts each row at its <eot>, like the chunked batch-1 loop of Showo.mmu_generate).
extern "C" int showo_engine_batch_decode_greedy(showo_engine* e, int64_t* tok, int n_steps, int64_t* out_tokens, float* logits_ws,
                                                int use_graph, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!e || !e->bd || e->bd->nb == 0) return set_error_msg(1, "batch_decode_greedy: showo_engine_batch_begin + _batch_prefill first");
    auto* d = e->bd;
    const int nb = d->nb, H = e->H, F = e->F, nH = e->nH, V = e->V;
    if (!tok || !out_tokens || !logits_ws || n_steps < 1) return set_error_msg(1, "batch_decode_greedy: bad arguments");
    if (e->precision == 1) return set_error_msg(1, "batch_decode_greedy: precision 0 (bf16 operands) or 2 (fp16 operands)");
    const int op = e->precision == 2 ? SHOWO_OP_F16 : SHOWO_OP_BF16;
    if ((op == SHOWO_OP_F16) != e->img_f16) return set_error_msg(4, "batch_decode_greedy: the weight images hold the other 16-bit type: upload the weights again");
    if (d->precision != e->precision) return set_error_msg(1, "batch_decode_greedy: the caches were prefilled under another precision: prefill again");
    if (op && (!e->wlm_lo || !e->lo_loaded.count("showo.lm_head.weight"))) return set_error_msg(4, "batch_decode_greedy (precision 2): the lm_head's low half is missing: upload the weights again");
    int lk_max = 0;
    int P0[MAXB];
    for (int b = 0; b < nb; ++b) {
        if (d->cache_len[b] <= 0) return set_error_msg(1, "batch_decode_greedy: a sequence has no prefill");
        P0[b] = d->cache_len[b];
        if (P0[b] + n_steps > d->cap || P0[b] + n_steps > e->cfg.max_pos) return set_error_msg(5, "batch_decode_greedy: cache full");
        const int *v = d->last_iv[b], L0 = d->prompt_len[b];
        if (!((v[1] == L0 && v[0] < v[1]) || (v[3] == L0 && v[2] < v[3]) || !(v[2] < v[3]) || !(v[0] < v[1])))
            return set_error_msg(6, "batch_decode_greedy: mask row needs more than two intervals");
        lk_max = lk_max > P0[b] + n_steps ? lk_max : P0[b] + n_steps;
    }
    SHOWO_CHECK_HIP(hipMemcpyAsync(d->pos_dev, P0, sizeof(int) * nb, hipMemcpyHostToDevice, s));
    SHOWO_CHECK_HIP(hipMemcpyAsync(d->base_dev, P0, sizeof(int) * nb, hipMemcpyHostToDevice, s));
    SHOWO_CHECK_HIP(hipMemcpyAsync(d->L0_dev, d->prompt_len, sizeof(int) * nb, hipMemcpyHostToDevice, s));
    SHOWO_CHECK_HIP(hipMemcpyAsync(d->last_iv_dev, d->last_iv, sizeof(int32_t) * 4 * nb, hipMemcpyHostToDevice, s));
    SHOWO_CHECK_HIP(hipStreamSynchronize(s));  // P0 is a host temporary of this call
    TRY(showo_embed_f32(tok, e->embed, e->x, nb, H, V, s));
    batch_iv_kernel<<<1, 64, 0, s>>>(d->last_iv_dev, d->L0_dev, d->pos_dev, d->iv_dev, nb);
    SHOWO_CHECK_HIP(hipGetLastError());
    const int64_t per_seq = (int64_t)nH * d->cap * 64, lstride = (int64_t)nb * per_seq;
    // co-scheduled layer (fc2 streams next to the latency-bound attention blocks): Phi-1.5's shape, 2..4 sequences (the fc2 role
    // keeps nb x 8192 bf16 activations in LDS); SHOWO_DECODE_BATCH_CO=0 / other shapes: three plain launches per layer
    static int co_on = -1;
    if (co_on < 0) {
        const char* env = getenv("SHOWO_DECODE_BATCH_CO");
        co_on = env ? (atoi(env) != 0) : 1;
    }
    const int co_blocks = showo::decode_tuning().batch_co_blocks;  // 128 role blocks + 32 nb attention blocks = one block per CU at nb = 4
    const bool co = co_on && F == 8192 && H <= 2048 && nb >= 2 && nb <= 4 && (size_t)nb * F * 2 <= 128 * 1024;
    if (co && !d->y2) TRY(e->alloc(&d->y2, (int64_t)MAXB * H));
    auto one = [&]() -> int {
        for (int li = 0; li < e->nL; ++li) {
            showo::Layer& l = e->layers[li];
            LnGemvBArgs a{e->x, l.ln_w, l.ln_b, e->cfg.ln_eps, H, l.wqkv, l.bqkv, e->qkv, nullptr, 3 * H, 3 * H, l.w1, l.b1, e->ffn, F, F};
            TRY(ln_gemvB(nb, a, s, op));
            OutGemvBArgs o{e->x, l.wd, e->attn, l.bd, H, H, l.w2, e->ffn, l.b2, F, F, H, d->y2};
            if (co) {
                // [ attention of the nb x heads (sequence, head) pairs || fc2 of all nb sequences -> y2 ] -> dense + both residual adds
                // (Phi's block is parallel-residual, models/phi.py:806-835: fc2 does not depend on the attention; decode.hip's batch-1 layer
                // co-schedules the same way)
                showo::DecodePrefetch pf;
                showo::decode_prefetch_plan(e, li, &pf);
                TRY(showo::attn_decode_co_batch(e->qkv, l.qln_w, l.qln_b, l.kln_w, l.kln_b, e->cosT, e->sinT, d->k + li * lstride,
                                                d->vt + li * lstride, d->iv_dev, e->attn, nb, nH, e->cfg.rotary_dim, e->cfg.ln_eps,
                                                d->pos_dev, lk_max, d->cap, d->cap, o, co_blocks, s, &pf, op));
                TRY(out_dense_y2B(nb, o, s, op));
            } else {
                TRY(showo::attn_decode_fused_batch(e->qkv, l.qln_w, l.qln_b, l.kln_w, l.kln_b, e->cosT, e->sinT, d->k + li * lstride,
                                                   d->vt + li * lstride, d->iv_dev, e->attn, nb, nH, e->cfg.rotary_dim, e->cfg.ln_eps,
                                                   d->pos_dev, lk_max, d->cap, d->cap, s, op));
                TRY(out_gemvB(nb, o, s, op));
            }
        }
        if (op) {  // precision 2: the lm_head is the split-bf16 product (its images stay bf16 (hi, lo) at that precision)
            TRY(showo::decode_split_head(e->x, e->fln_w, e->fln_b, e->cfg.ln_eps, H, e->wlm, e->wlm_lo, e->blm, logits_ws, V, V, nb, s));
        } else {
            LnGemvBArgs h{e->x, e->fln_w, e->fln_b, e->cfg.ln_eps, H, e->wlm, e->blm, nullptr, logits_ws, V, V, nullptr, nullptr, nullptr, 0, 0};
            TRY(ln_gemvB(nb, h, s));
        }
        greedy_seam_rows_kernel<<<dim3(nb), dim3(1024), 0, s>>>(logits_ws, V, tok, out_tokens, n_steps, d->pos_dev, d->base_dev, e->embed, e->x,
                                                               H, d->last_iv_dev, d->L0_dev, d->iv_dev);
        SHOWO_CHECK_HIP(hipGetLastError());
        return 0;
    };
    int rc = one();  // eager first step (kernel attributes)
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    const bool graph = use_graph && n_steps > 1 && !showo::g_prof_on_query();
    if (!rc && graph) {
        hipError_t he = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        if (he == hipSuccess) {
            rc = one();
            hipError_t he2 = hipStreamEndCapture(s, &g);
            if (!rc && he2 != hipSuccess) rc = set_error_hip(he2, "hipStreamEndCapture", __FILE__, __LINE__);
        } else {
            rc = set_error_hip(he, "hipStreamBeginCapture", __FILE__, __LINE__);
        }
        if (!rc) {
            hipError_t he3 = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            if (he3 != hipSuccess) rc = set_error_hip(he3, "hipGraphInstantiate", __FILE__, __LINE__);
        }
        for (int i = 1; !rc && i < n_steps; ++i) {
            hipError_t he4 = hipGraphLaunch(ge, s);
            if (he4 != hipSuccess) rc = set_error_hip(he4, "hipGraphLaunch", __FILE__, __LINE__);
        }
    } else {
        for (int i = 1; !rc && i < n_steps; ++i) rc = one();
    }
    if (ge) { hipStreamSynchronize(s); hipGraphExecDestroy(ge); }
    if (g) hipGraphDestroy(g);
    if (rc) return rc;
    for (int b = 0; b < nb; ++b) d->cache_len[b] = P0[b] + n_steps;
    return 0;
}
