// Pieces of the AR-decode GEMV kernels shared by decode.hip and the co-scheduled attention launch (attention.hip).
#pragma once
#include "common.h"

namespace showo {

static __device__ __forceinline__ void load4(const bf16_t* row, int k0, int K, uint4 (&wv)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * 512;
        wv[u] = k < K ? ldg_nt16(row + k) : make_uint4(0, 0, 0, 0);
    }
}
// acc += sum over the 4 loaded 8-element groups, in gemv_kernel's order (u ascending, pairs ascending: dot8_bf16)
// F16: the operand type of weights and activations (common.h Op16; precision 2 = IEEE half on v_dot2_f32_f16)
template <bool F16 = false, class AP>
static __device__ __forceinline__ float fma4(const uint4 (&wv)[4], AP act, int k0, int K, float acc) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * 512;
        if (k >= K) break;
        acc = dot8_op<F16>(wv[u], *reinterpret_cast<const uint4*>(act + k), acc);
    }
    return acc;
}

struct OutGemvArgs {
    float* x;           // [N] fp32 residual stream row, updated in place
    const bf16_t* W0;   // [N, K0] dense weight,  a0 [K0] attention output
    const bf16_t* a0;
    const float* b0;
    int K0;
    const bf16_t* W1;   // [N, K1] fc2 weight,    a1 [K1] gelu(fc1) row
    const bf16_t* a1;
    const float* b1;
    int K1;
    int N;
    float* y2;          // [N] fp32: fc2 + b2 of the forked layer (MODE 1 writes it, MODE 2 adds it)
};


// fc2 role of the co-scheduled decode launch (attention.hip, attn_decode_co_kernel): y2[n] = W1[n, :] a1 + b1[n] for the columns of
// role-block `rb` of `nrb`, `nw` waves per block.  Same lane split, accumulation order and epilogue expression as
// out_gemv2_kernel<C, 1> (decode.hip), so the result does not depend on which launch computed it.  sa: >= K1 bf16 of LDS.
template <int C, bool F16 = false>
static __device__ __forceinline__ void fc2_columns_role(const OutGemvArgs& g, int rb, int nrb, int nw, bf16_t* sa) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int stride = nrb * nw;
    int n = wave * nrb + rb;
    uint4 buf[C][4];
    if (n < g.N) {
#pragma unroll
        for (int t = 0; t < C; ++t) load4(g.W1 + (int64_t)n * g.K1, t * 2048 + lane * 8, g.K1, buf[t]);
    }
    for (int i = threadIdx.x * 8; i < g.K1; i += nw * 64 * 8) *reinterpret_cast<uint4*>(sa + i) = *reinterpret_cast<const uint4*>(g.a1 + i);
    __syncthreads();
    while (n < g.N) {
        const int nn = n + stride;
        float acc1 = 0.f;
#pragma unroll
        for (int t = 0; t < C; ++t) {
            acc1 = fma4<F16>(buf[t], sa, t * 2048 + lane * 8, g.K1, acc1);
            if (nn < g.N) load4(g.W1 + (int64_t)nn * g.K1, t * 2048 + lane * 8, g.K1, buf[t]);
        }
        acc1 = wave_sum_swap(acc1);
        if (lane == 0) g.y2[n] = acc1 + g.b1[n];
        n = nn;
    }
}

// ---- Infinity-Cache prefetch role of the co-scheduled decode launches ---------------------------------------------------------------
// The attention launch of a decode layer is latency-bound (32 x NB single-query blocks) and its fc2 role needs 33.5 MB: HBM idles for
// part of it, and the CUs neither role occupies idle throughout.  Extra blocks of the SAME launch read weights the NEXT launches will
// stream (this layer's dense matrix, the next layer's [Wqkv ; W1]) so that those launches find them in the 256 MiB memory-side cache.
// The reads are LDS-DMA (global_load_lds, 16 B per lane, default cache policy, no VGPR round trip; the LDS image is scratch that every
// wave overwrites): nothing is computed from them, the product's results cannot depend on this role.
struct DecodePrefetch {
    const void* p[3];   // up to three segments, read in order
    int64_t bytes[3];   // multiples of 16
    int blocks;         // prefetch blocks appended to the grid (0: role absent)
};
// block rb of nrb: a contiguous slice of every segment; lds: >= nw KiB of LDS (1 KiB per wave)
static __device__ __forceinline__ void prefetch_role(const DecodePrefetch& pf, int rb, int nrb, void* lds) {
    const int wave = threadIdx.x >> 6, nthr = blockDim.x;
    bf16_t* dst = reinterpret_cast<bf16_t*>(lds) + wave * 512;
#pragma unroll 1
    for (int sgm = 0; sgm < 3; ++sgm) {
        const int64_t n16 = pf.bytes[sgm] >> 4;
        if (n16 <= 0) continue;
        const int64_t per = (((n16 + nrb - 1) / nrb) + nthr - 1) / nthr * nthr;
        const int64_t i0 = (int64_t)rb * per, i1 = i0 + per < n16 ? i0 + per : n16;
        const bf16_t* src = reinterpret_cast<const bf16_t*>(pf.p[sgm]);
        for (int64_t i = i0 + threadIdx.x; i < i1; i += nthr) glds16(src + i * 8, dst);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- batched decode (decode_batch.hip): NB sequences share one weight stream --------------------------------------------------
struct OutGemvBArgs {
    float* x;          // [NB, N] fp32 residual rows (ld = N), updated in place
    const bf16_t* W0;  // [N, K0] dense;  a0 [NB, K0] (lda0)
    const bf16_t* a0;
    const float* b0;
    int K0, lda0;
    const bf16_t* W1;  // [N, K1] fc2;    a1 [NB, K1] (lda1)
    const bf16_t* a1;
    const float* b1;
    int K1, lda1;
    int N;
    float* y2;         // [NB, N] fp32: fc2 + b2 per sequence (written by the co-scheduled role, added by out_gemvB_kernel<.., 2>)
};

// wave_sum of up to four accumulators at once, each with wave_sum's own order of additions (partners 32, 16, 8, 4, 2, 1 away: the same
// bits).  The 32 / 16 steps run on every accumulator with gfx950's row-swap instructions (VALU only); after them a lane holds, for
// each b, the sum over its 4-lane group {l, l^16, l^32, l^48}, so lane group g = lane >> 4 continues with accumulator g alone: ONE
// register takes the remaining four steps (DPP: row_sum_dpp).  No LDS-crossbar permute at all (6 per sequence in wave_sum).  Returns, in the lanes
// 16 b .. 16 b + 15, the wave-wide sum of acc[b] (b < NB <= 4).
template <int NB>
static __device__ __forceinline__ float wave_sum_groups(const float (&acc)[NB]) {
    static_assert(NB >= 1 && NB <= 4, "one 16-lane group per accumulator");
    const int g = (threadIdx.x >> 4) & 3;
    float r = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[b]), __float_as_uint(acc[b]), false, false);
        const float v = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
        const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        const float w = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
        r = g == b ? w : r;
    }
    return row_sum_dpp(r);
}

// acc[b] += sum over the 4 loaded 8-element groups of w * act[b], in fma4's order (u ascending, pairs ascending) per sequence.
// act: this lane's 32 activation values per sequence as 16 packed bf16 pairs (registers).
template <int NB, bool F16 = false>
static __device__ __forceinline__ void fma4_regs(const uint4 (&wv)[4], const uint32_t (&act)[NB][16], int k0, int K, float (&acc)[NB]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (k0 + u * 512 >= K) break;
        const uint32_t w[4] = {wv[u].x, wv[u].y, wv[u].z, wv[u].w};
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = Op16<F16>::dot2(w[p], act[b][u * 4 + p], acc[b]);
    }
}
// this lane's 16 pairs of one bf16 activation row (global or LDS), zero beyond K
template <class AP>
static __device__ __forceinline__ void load_act_pairs(AP row, int k0, int K, uint32_t (&act)[16]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * 512;
        uint4 av = make_uint4(0, 0, 0, 0);
        if (k < K) av = *reinterpret_cast<const uint4*>(row + k);
        act[u * 4 + 0] = av.x; act[u * 4 + 1] = av.y; act[u * 4 + 2] = av.z; act[u * 4 + 3] = av.w;
    }
}
// the same with the activations in LDS (sa: row b at sa + b * ld, bf16): one 16-byte read per 8 products per sequence
template <int NB, bool F16 = false>
static __device__ __forceinline__ void fma4_lds(const uint4 (&wv)[4], const bf16_t* sa, int ld, int k0, int K, float (&acc)[NB]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * 512;
        if (k >= K) break;
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = dot8_op<F16>(wv[u], *reinterpret_cast<const uint4*>(sa + (size_t)b * ld + k), acc[b]);
    }
}

// fc2 role of the co-scheduled BATCHED decode launch (attention.hip, attn_decode_coB_kernel): y2[b][n] = W1[n, :] a1[b] + b1[n] for
// the columns of role-block rb of nrb, nw waves per block.  fc2_columns_role's lane split and accumulation order per sequence (chunks
// t ascending into ONE accumulator chain, then the wave reduction): the bits of the batch-1 launch.  The NB activation rows live in
// LDS as bf16 (NB * K1 * 2 bytes: 64 KiB at NB = 4).
template <int C, int NB, bool F16 = false>
static __device__ __forceinline__ void fc2_columns_roleB(const OutGemvBArgs& g, int rb, int nrb, int nw, bf16_t* sa) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int stride = nrb * nw;
    int n = wave * nrb + rb;
    uint4 buf[C][4];
    if (n < g.N) {
#pragma unroll
        for (int t = 0; t < C; ++t) load4(g.W1 + (int64_t)n * g.K1, t * 2048 + lane * 8, g.K1, buf[t]);
    }
    for (int b = 0; b < NB; ++b)
        for (int i = threadIdx.x * 8; i < g.K1; i += nw * 64 * 8)
            *reinterpret_cast<uint4*>(sa + (size_t)b * g.K1 + i) = *reinterpret_cast<const uint4*>(g.a1 + (int64_t)b * g.lda1 + i);
    __syncthreads();
    while (n < g.N) {
        const int nn = n + stride;
        float acc1[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc1[b] = 0.f;
#pragma unroll
        for (int t = 0; t < C; ++t) {
            fma4_lds<NB, F16>(buf[t], sa, g.K1, t * 2048 + lane * 8, g.K1, acc1);
            if (nn < g.N) load4(g.W1 + (int64_t)nn * g.K1, t * 2048 + lane * 8, g.K1, buf[t]);
        }
        const float tot = wave_sum_groups<NB>(acc1);
        if ((lane & 15) == 0 && (lane >> 4) < NB) g.y2[(int64_t)(lane >> 4) * g.N + n] = tot + g.b1[n];
        n = nn;
    }
}
}  // namespace showo
