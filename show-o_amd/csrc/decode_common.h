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
// acc += sum over the 4 loaded 8-element groups, in gemv_kernel's order (u ascending, j ascending)
template <class AP>
static __device__ __forceinline__ float fma4(const uint4 (&wv)[4], AP act, int k0, int K, float acc) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * 512;
        if (k >= K) break;
        const uint4 av = *reinterpret_cast<const uint4*>(act + k);
        const bf16_t* ea = reinterpret_cast<const bf16_t*>(&av);
        const bf16_t* ew = reinterpret_cast<const bf16_t*>(&wv[u]);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = fmaf(bf2f(ea[j]), bf2f(ew[j]), acc);
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
template <int C>
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
            acc1 = fma4(buf[t], sa, t * 2048 + lane * 8, g.K1, acc1);
            if (nn < g.N) load4(g.W1 + (int64_t)nn * g.K1, t * 2048 + lane * 8, g.K1, buf[t]);
        }
        acc1 = wave_sum(acc1);
        if (lane == 0) g.y2[n] = acc1 + g.b1[n];
        n = nn;
    }
}

}  // namespace showo
