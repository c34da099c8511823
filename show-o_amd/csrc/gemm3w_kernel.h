// Kernel template and launchers of the 3-deep-weight-ring GEMM (documentation: gemm3w.hip).  Included by gemm3w.hip (bfloat16
// operands) and gemm3w_f16.hip (IEEE-half operands, precision 2).
#pragma once
#include "gemm_common.h"

namespace showo {
int gemm3w_variant_f16(const GemmArgs& g, int epilogue, int rows, hipStream_t s);  // gemm3w_f16.hip
namespace g3w {

// BL: the DMAs are `buffer_load_dwordx4 ... offen lds` through wave-uniform buffer descriptors (operand base in SGPRs, the
// per-lane 32-bit offset as voffset, the k advance as soffset): no 64-bit address VALU per DMA (three v_lshl_add_u64 each with
// global_load_lds) -- the load segment is the critical path of the phase program.
__device__ __forceinline__ void bufl16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, bf16_t* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

template <int EPI, int MF0, int MF1, bool NS, bool BL, bool F16 = false>
__global__ __launch_bounds__(512) void gemm3w_kernel(GemmArgs g) {
    static_assert(NS ? (MF0 >= MF1 && MF1 >= 3 && MF0 <= 6) : (MF0 >= 5 && MF0 <= 8 && MF1 >= 4 && MF1 <= 8),
                  "m-split: each group needs 4 lo fragments, group 0 at least one hi fragment; n-split: at most 6 fragments per group");
    constexpr int BK = GEMM_BK;
    constexpr int BMT = 16 * (MF0 + MF1);
    constexpr int NA = 2 * (MF0 + MF1);
    constexpr int NPW = (NA + 7) / 8;
    constexpr int NAO = NS ? NPW : 4;
    constexpr int NHI = NS ? 1 : 2 * (MF0 - 4) + 2 * (MF1 - 4);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesM = (g.M + BMT - 1) / BMT, tilesN = (g.N + B2 - 1) / B2;
    int nwg = tilesM * tilesN, bid = blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int tn, tm;
    {
        const int per = g.gn * tilesM;
        const int grp = bid / per, rem = bid - grp * per;
        const int first = grp * g.gn;
        const int gsz = min(tilesN - first, g.gn);
        tm = rem / gsz;
        tn = first + (rem - tm * gsz);
    }
    const int m0 = tm * BMT, n0 = tn * B2;
    const int nk = g.K / BK;
    const bool g_stage = !(g.flags & 8);  // bf16 epilogues store through LDS (full 128-B lines); flags bit 3 = direct stores (A/B)
    const int wn = wave & 3, wm = wave >> 2;
    const int gbase = wm * 16 * MF0;

    const int srow = lane >> 3;
    const int coff = ((lane & 7) ^ srow) << 3;
    constexpr bool KC = (EPI == SHOWO_EPI_RESID_F32);
    const int Ks = KC ? g.Ksplit : (1 << 30);
    const int wks = g.wtiled ? 8 : 0;
    const char* wbase = reinterpret_cast<const char*>(g.W) + (g.wtiled ? (size_t)tn * (size_t)(g.K / BK) * 32768 : (size_t)0);
    const char* abase0 = reinterpret_cast<const char*>(g.A);
    const char* abase1 = (KC && g.A2) ? reinterpret_cast<const char*>(g.A2) - (int64_t)Ks * 2 : abase0;
    const int lda1 = (KC && g.A2) ? g.lda2 : g.lda;
    constexpr int WBUF = 256 * 64;          // elements per W buffer (32 KiB)
    constexpr int AOFF = 3 * WBUF;          // A buffers behind the W ring
    uint32_t woff[2][2], aoff[KC ? 2 : 1][NAO];
    int arowl[NAO], wrowl[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int row;
            if (NS) { const int q = wave * 2 + i; row = (q >> 2) * 64 + h * 32 + (q & 3) * 8; }  // half h = fragments 2h, 2h+1 of every wave column
            else row = h * 128 + i * 64 + wave * 8;
            wrowl[h][i] = row;
            int n = n0 + row + srow;
            n = n < g.N ? n : g.N - 1;
            woff[h][i] = g.wtiled ? (uint32_t)(row * 128 + lane * 16) : (uint32_t)(((int64_t)n * g.ldw + coff) * 2);
        }
#pragma unroll
    for (int j = 0; j < NAO; ++j) {
        int row;
        if (NS) { int p = wave + 8 * j; p = p < NA ? p : p % NA; row = 8 * p; }
        else if (j < 2) row = j * 16 * MF0 + wave * 8;
        else {
            int p = wave + 8 * (j - 2);
            p = p < NHI ? p : p % NHI;
            row = p < 2 * (MF0 - 4) ? 64 + 8 * p : 16 * MF0 + 64 + 8 * (p - 2 * (MF0 - 4));
        }
        arowl[j] = row;
        int m = m0 + row + srow;
        m = m < g.M ? m : g.M - 1;
        aoff[0][j] = (uint32_t)(((int64_t)m * g.lda + coff) * 2);
        if (KC) aoff[KC ? 1 : 0][j] = (uint32_t)(((int64_t)m * lda1 + coff) * 2);
    }
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbase), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(abase0), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(abase1), 0, -1, 0x00020000);
    // W piece (H, i) of k-tile at element offset K0 into ring slot WB
#define R_DMA_W(WB, H, K0)                                                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                            \
        if constexpr (BL) bufl16(rsW, woff[H][i_], (uint32_t)(((K0) * 2) << wks), smem + (WB) * WBUF + wrowl[H][i_] * 64); \
        else glds16(reinterpret_cast<const bf16_t*>(wbase + (((size_t)(K0) * 2) << wks) + (size_t)woff[H][i_]),  \
                    smem + (WB) * WBUF + wrowl[H][i_] * 64);                                                      \
    }
#define R_DMA_A(AB, J0, J1, K0)                                                                                   \
    do {                                                                                                          \
        const bool s1_ = KC && (K0) >= Ks;                                                                        \
        const char* ab_ = (s1_ ? abase1 : abase0) + (size_t)(K0) * 2;                                             \
        _Pragma("unroll") for (int j_ = (J0); j_ < (J1); ++j_) {                                                  \
            if constexpr (BL) bufl16(s1_ ? rsA1 : rsA0, s1_ ? aoff[KC ? 1 : 0][j_] : aoff[0][j_], (uint32_t)((K0) * 2), \
                                     smem + AOFF + (AB) * WBUF + arowl[j_] * 64);                                 \
            else glds16(reinterpret_cast<const bf16_t*>(ab_ + (size_t)(s1_ ? aoff[KC ? 1 : 0][j_] : aoff[0][j_])), \
                        smem + AOFF + (AB) * WBUF + arowl[j_] * 64);                                              \
        }                                                                                                         \
    } while (0)

    const int fr = lane & 15, fg = lane >> 4;
    const int lsw0 = fr * 64 + ((fg ^ (fr & 7)) << 3);
    const int lsw1 = fr * 64 + (((fg + 4) ^ (fr & 7)) << 3);
    const bf16_t* ldsW = smem + (wn * 64) * 64;
    const bf16_t* ldsA = smem + AOFF + gbase * 64;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NAF = NS ? MF0 : 4;
    bf16x8 wf[2][4], af[2][NAF];

#define R_READ_W(WB)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                               \
        wf[0][i] = *reinterpret_cast<const bf16x8*>(ldsW + (WB) * WBUF + i * 16 * 64 + lsw0);                     \
        wf[1][i] = *reinterpret_cast<const bf16x8*>(ldsW + (WB) * WBUF + i * 16 * 64 + lsw1);                     \
    }
#define R_READ_A(AB, MB, CNT)                                                                                     \
    _Pragma("unroll") for (int j = 0; j < (CNT); ++j) {                                                           \
        af[0][j] = *reinterpret_cast<const bf16x8*>(ldsA + (AB) * WBUF + ((MB) + j) * 16 * 64 + lsw0);            \
        af[1][j] = *reinterpret_cast<const bf16x8*>(ldsA + (AB) * WBUF + ((MB) + j) * 16 * 64 + lsw1);            \
    }
#define R_MFMA(MB, CNT)                                                                                           \
    do {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                          \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
                _Pragma("unroll") for (int j = 0; j < (CNT); ++j)                                                 \
                    acc[i][(MB) + j] = Op16<F16>::mfma16(wf[kk][i], af[kk][j], acc[i][(MB) + j]);                 \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)
#define R_READ_W2(WB, I0)                                                                                         \
    _Pragma("unroll") for (int i = (I0); i < (I0) + 2; ++i) {                                                     \
        wf[0][i] = *reinterpret_cast<const bf16x8*>(ldsW + (WB) * WBUF + i * 16 * 64 + lsw0);                     \
        wf[1][i] = *reinterpret_cast<const bf16x8*>(ldsW + (WB) * WBUF + i * 16 * 64 + lsw1);                     \
    }
#define R_MFMA_N(I0, CNT)                                                                                         \
    do {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                          \
            _Pragma("unroll") for (int i = (I0); i < (I0) + 2; ++i)                                               \
                _Pragma("unroll") for (int j = 0; j < (CNT); ++j)                                                 \
                    acc[i][j] = Op16<F16>::mfma16(wf[kk][i], af[kk][j], acc[i][j]);                               \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)
#define R_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
    // n-split k-tile on the ring
#define RN_TILE(WB, AB, T, MFG)                                                                                   \
    do {                                                                                                          \
        const int kA = ((T) + 1) * BK, kW = ((T) + 2) * BK;                                                       \
        const bool hasA = (T) + 1 < nk, hasW = (T) + 2 < nk;                                                      \
        /* ph0 */                                                                                                 \
        R_READ_W2(WB, 0)                                                                                          \
        R_READ_A(AB, 0, MFG)                                                                                      \
        if (hasA) R_DMA_A((AB) ^ 1, 0, NPW, kA);                                                                  \
        bar_raw_fn();                                                                                             \
        R_MFMA_N(0, MFG);                                                                                         \
        bar_raw_fn();                                                                                             \
        /* ph1 */                                                                                                 \
        R_READ_W2(WB, 2)                                                                                          \
        if (hasW) { R_DMA_W(((WB) + 2) % 3, 0, kW); R_DMA_W(((WB) + 2) % 3, 1, kW); R_WAIT(4); } else { R_WAIT(0); } \
        bar_raw_fn();                                                                                             \
        R_MFMA_N(2, MFG);                                                                                         \
        bar_raw_fn();                                                                                             \
    } while (0)
    // one k-tile: ring slot WB = T % 3, A buffer AB = T % 2 (literals)
#define R_TILE(WB, AB, T, MFG)                                                                                    \
    do {                                                                                                          \
        const int kA = ((T) + 1) * BK, kW = ((T) + 2) * BK;                                                       \
        const bool hasA = (T) + 1 < nk, hasW = (T) + 2 < nk;                                                      \
        /* ph0 */                                                                                                 \
        R_READ_W(WB)                                                                                              \
        R_READ_A(AB, 0, 4)                                                                                        \
        if (hasA) R_DMA_A((AB) ^ 1, 0, 2, kA);                                                                    \
        if (hasW) { R_DMA_W(((WB) + 2) % 3, 0, kW); R_WAIT(6); } else { R_WAIT(0); }                              \
        bar_raw_fn();                                                                                             \
        R_MFMA(0, 4);                                                                                             \
        bar_raw_fn();                                                                                             \
        /* ph1 */                                                                                                 \
        R_READ_A(AB, 4, NS ? 0 : (MFG) - 4)                                                                       \
        if (hasA) R_DMA_A((AB) ^ 1, 2, 4, kA);                                                                    \
        if (hasW) { R_DMA_W(((WB) + 2) % 3, 1, kW); R_WAIT(6); } else { R_WAIT(0); }                              \
        bar_raw_fn();                                                                                             \
        R_MFMA(4, NS ? 0 : (MFG) - 4);                                                                            \
        bar_raw_fn();                                                                                             \
    } while (0)

    // ---- prologue: all of tile 0 and the weights of tile 1 (the latter may still be in flight: retired by ph1(0)'s wait)
    R_DMA_W(0, 0, 0);
    R_DMA_W(0, 1, 0);
    R_DMA_A(0, 0, NAO, 0);
    if (nk > 1) {
        R_DMA_W(1, 0, BK);
        R_DMA_W(1, 1, BK);
        R_WAIT(4);
    } else {
        R_WAIT(0);
    }
    bar_raw_fn();
    if (wm == 1) bar_raw_fn();  // group 1 runs one barrier behind group 0

#define R_ANY(WB, AB, T, MFG) do { if constexpr (NS) { RN_TILE(WB, AB, T, MFG); } else { R_TILE(WB, AB, T, MFG); } } while (0)
#define R_RUN(MFG)                                                                                                \
    do {                                                                                                          \
        int t = 0;                                                                                                \
        for (; t + 5 < nk; t += 6) {                                                                              \
            R_ANY(0, 0, t, MFG);                                                                                  \
            R_ANY(1, 1, t + 1, MFG);                                                                              \
            R_ANY(2, 0, t + 2, MFG);                                                                              \
            R_ANY(0, 1, t + 3, MFG);                                                                              \
            R_ANY(1, 0, t + 4, MFG);                                                                              \
            R_ANY(2, 1, t + 5, MFG);                                                                              \
        }                                                                                                         \
        if (t < nk) R_ANY(0, 0, t, MFG);                                                                          \
        if (t + 1 < nk) R_ANY(1, 1, t + 1, MFG);                                                                  \
        if (t + 2 < nk) R_ANY(2, 0, t + 2, MFG);                                                                  \
        if (t + 3 < nk) R_ANY(0, 1, t + 3, MFG);                                                                  \
        if (t + 4 < nk) R_ANY(1, 0, t + 4, MFG);                                                                  \
    } while (0)
    if (MF0 == MF1 || wm == 0) {
        R_RUN(MF0);
        if (wm == 0) bar_raw_fn();  // re-align the barrier counts of the two groups
        if constexpr (EPI == EPI_QKV_SPLIT) epilogue_qkv_split<MF0>(g, acc, n0, wn, m0 + gbase, fr, fg, g_stage ? smem + wave * 8192 : nullptr);
        else epilogue8p<EPI, MF0, F16>(g, acc, n0, wn, m0 + gbase, fr, fg, g_stage ? smem + wave * 8192 : nullptr);
    } else {
        R_RUN(MF1);
        if constexpr (EPI == EPI_QKV_SPLIT) epilogue_qkv_split<MF1>(g, acc, n0, wn, m0 + gbase, fr, fg, g_stage ? smem + wave * 8192 : nullptr);
        else epilogue8p<EPI, MF1, F16>(g, acc, n0, wn, m0 + gbase, fr, fg, g_stage ? smem + wave * 8192 : nullptr);
    }
#undef R_RUN
#undef R_ANY
#undef RN_TILE
#undef R_MFMA_N
#undef R_READ_W2
#undef R_TILE
#undef R_WAIT
#undef R_MFMA
#undef R_READ_A
#undef R_READ_W
#undef R_DMA_A
#undef R_DMA_W
}

constexpr int SMEM3W_BYTES = 5 * 256 * 64 * 2;  // 160 KiB

template <int EPI, bool F16, int MF0, int MF1, bool NS, bool BL = false>
int launch3w(const GemmArgs& g, hipStream_t s) {
    static bool attr_set = false;
    auto kfn = gemm3w_kernel<EPI, MF0, MF1, NS, BL, F16>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM3W_BYTES);
        if (e != hipSuccess) return set_error_hip(e, "hipFuncSetAttribute(gemm3w)", __FILE__, __LINE__);
        attr_set = true;
    }
    constexpr int BMT = 16 * (MF0 + MF1);
    int tilesM = (g.M + BMT - 1) / BMT, tilesN = (g.N + B2 - 1) / B2;
    kfn<<<dim3(tilesM * tilesN), dim3(512), SMEM3W_BYTES, s>>>(g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error_hip(e, "gemm3w launch", __FILE__, __LINE__);
    return 0;
}

template <int EPI, bool F16 = false>
int launch3w_h(const GemmArgs& g, int rows, hipStream_t s) {
    switch (rows) {
        case 240: return launch3w<EPI, F16, 8, 7, false>(g, s);
        case 224: return launch3w<EPI, F16, 7, 7, false>(g, s);
        case 208: return launch3w<EPI, F16, 7, 6, false>(g, s);
        case 1192: return launch3w<EPI, F16, 6, 6, true>(g, s);
        case 1176: return launch3w<EPI, F16, 6, 5, true>(g, s);
        case 1160: return launch3w<EPI, F16, 5, 5, true>(g, s);
        case 1144: return launch3w<EPI, F16, 5, 4, true>(g, s);
        case 2192: return launch3w<EPI, F16, 6, 6, true, true>(g, s);   // 4xxx: n-split on the ring with buffer-descriptor DMAs
        case 2176: return launch3w<EPI, F16, 6, 5, true, true>(g, s);
        case 2160: return launch3w<EPI, F16, 5, 5, true, true>(g, s);
        case 2144: return launch3w<EPI, F16, 5, 4, true, true>(g, s);
    }
    return launch3w<EPI, F16, 8, 8, false>(g, s);
}

}  // namespace g3w
}  // namespace showo
