// Kernel template and launchers of the production GEMM (documentation: gemm2p.hip).  Included by gemm2p.hip (bfloat16 operands) and
// gemm2p_f16.hip (IEEE-half operands, precision 2): one translation unit per operand type so that the ~150 kernel instances of each
// compile side by side.  Host state (split-K workspaces, cooperative-reduction ownership, counters) lives in gemm2p.hip.
#pragma once
#include "gemm_common.h"

namespace showo {

// host state of the family (gemm2p.hip); the caller of launch2p holds g_gemm_mu
constexpr int SPLITK_TICKS = 6144;  // arrivals [0, 2048) | departures [2048, 4096) | cooperative-reduction mode [4096, 6144)
bool splitk_ws(hipStream_t s, size_t need, float4** ws, unsigned** tick);
int splitk_count(int M, int N, int K, int cus);
int splitk_coop_mode();
bool splitk_coop_ok(int blocks, hipStream_t s);
void splitk_coop_launched(hipStream_t s);
extern int64_t g_cnt_gemm2p, g_cnt_qkv_save, g_cnt_splitk;
// one tile variant of one epilogue, by run-time codes: gemm2p.hip (bf16) / gemm2p_f16.hip (fp16)
int gemm2p_variant_bf16(const GemmArgs& g, int epilogue, int h, hipStream_t s);
int gemm2p_variant_f16(const GemmArgs& g, int epilogue, int h, hipStream_t s);

namespace g2p {

template <int EPI, int MF0, int MF1, bool NS, bool F16 = false>  // F16: IEEE-half operands (common.h Op16), same tiles / phase programs
__global__ __launch_bounds__(512) void gemm2p_kernel(GemmArgs g) {
    static_assert(NS ? (MF0 >= MF1 && MF1 >= 3 && MF0 <= 6) : (MF0 >= 5 && MF0 <= 8 && MF1 >= 4 && MF1 <= 8),
                  "m-split: each group needs 4 lo fragments, group 0 at least one hi fragment; n-split: at most 6 fragments per group");
    constexpr int BK = GEMM_BK;
    constexpr int BMT = 16 * (MF0 + MF1);
    constexpr int NA = 2 * (MF0 + MF1);                 // A pieces (8 rows x 128 B) per k-tile
    constexpr int NPW = (NA + 7) / 8;                   // n-split: A pieces per wave
    constexpr int NHI = NS ? 1 : 2 * (MF0 - 4) + 2 * (MF1 - 4);  // m-split: hi pieces
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesM = (g.M + BMT - 1) / BMT, tilesN = (g.N + B2 - 1) / B2;
    int nwg = tilesM * tilesN, bid = blockIdx.x;
    int split = 0;
    if (g.splits > 1) { split = bid / nwg; bid -= split * nwg; }  // split-K: grid = tiles x splits (gemm_common.h)
    {   // XCD-aware bijective remap: blocks with equal (id % 8) share an L2; give each XCD a contiguous id range
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int tn, tm;
    {   // grouped order: gn n-panels wide, m fastest inside a group
        const int per = g.gn * tilesM;
        const int grp = bid / per, rem = bid - grp * per;
        const int first = grp * g.gn;
        const int gsz = min(tilesN - first, g.gn);
        tm = rem / gsz;
        tn = first + (rem - tm * gsz);
    }
    const int m0 = tm * BMT, n0 = tn * B2;
    int nk = g.K / BK;
    int kb = 0;  // first k column of this block's range (split-K)
    if (g.splits > 1) {
        const int per = (nk + g.splits - 1) / g.splits;
        kb = split * per * BK;
        nk = min(nk - split * per, per);
    }
    const bool g_stage = !(g.flags & 8);  // bf16 epilogues store through LDS (full 128-B lines); flags bit 3 = direct stores (A/B)
    const int wn = wave & 3, wm = wave >> 2;
    const int gbase = wm * 16 * MF0;  // first tile row of this wave's group

    // ---- DMA roles (32-bit byte offsets from the operand base; LDS destinations are wave-uniform).  A piece = 8 rows x 128 B =
    // one wave-instruction; its LDS image is lane-linear, so the (row & 7) chunk swizzle is applied to the SOURCE address.
    const int srow = lane >> 3;
    const int coff = ((lane & 7) ^ srow) << 3;
    constexpr bool KC = (EPI == SHOWO_EPI_RESID_F32);  // K-concatenated A operand: only the residual epilogue carries the second offset set
    const int Ks = KC ? g.Ksplit - kb : (1 << 30);  // relative to this block's first column
    // row-major weights: base + k * 2 + row offset.  Tiled weights ([N/256][K/64][256][64] bf16, showo_gemm_tile_weight): panel base +
    // (k / 64) * 32 KiB + offset inside the block -- every k-tile of a panel is ONE contiguous 32 KiB read (DRAM-page and TLB friendly;
    // a wave-instruction reads 1 KiB contiguous instead of 8 lines 2 ldw bytes apart)
    const int wks = g.wtiled ? 8 : 0;  // tiled: (k * 2) << 8 = (k / 64) * 32768 for k a multiple of 64
    const char* wbase = reinterpret_cast<const char*>(g.W) + (g.wtiled ? (size_t)tn * (size_t)(g.K / BK) * 32768 : (size_t)0) +
                        (((size_t)kb * 2) << wks);
    const char* abase0 = reinterpret_cast<const char*>(g.A) + (size_t)kb * 2;
    // segment 1 base is biased by -Ksplit so that base + k * 2 addresses column k - Ksplit
    const char* abase1 = (KC && g.A2) ? reinterpret_cast<const char*>(g.A2) - ((int64_t)g.Ksplit - kb) * 2 : abase0;
    const int lda1 = (KC && g.A2) ? g.lda2 : g.lda;
    constexpr int AOFF = 2 * 256 * 64;  // LDS (elements): W[buf][256][64] at 0, A[buf][256][64] behind it
    constexpr int NAO = NS ? NPW : 4;
    uint32_t woff[2][2], aoff[KC ? 2 : 1][NAO];   // aoff[segment][piece]; m-split pieces: 0,1 = lo of group 0,1; 2,3 = hi
    int wrowl[2][2], arowl[NAO];         // LDS row of each piece (wave-uniform)
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
            // tiled weights: the (panel, k-tile) block is the LDS image itself (256 rows x 128 B, chunks pre-swizzled): lane-linear source
            woff[h][i] = g.wtiled ? (uint32_t)(row * 128 + lane * 16) : (uint32_t)(((int64_t)n * g.ldw + coff) * 2);
        }
#pragma unroll
    for (int j = 0; j < NAO; ++j) {
        int row;
        if (NS) { int p = wave + 8 * j; p = p < NA ? p : p % NA; row = 8 * p; }
        else if (j < 2) row = j * 16 * MF0 + wave * 8;  // lo piece `wave` of group j
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
#define Q2_DMA_W(BUF, H, K0)                                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                              \
        glds16(reinterpret_cast<const bf16_t*>(wbase + (((size_t)(K0) * 2) << wks) + (size_t)woff[H][i_]),       \
               smem + (BUF) * 256 * 64 + wrowl[H][i_] * 64)
#define Q2_DMA_A(BUF, J0, J1, K0)                                                                                 \
    do {                                                                                                          \
        const bool s1_ = KC && (K0) >= Ks;                                                                        \
        const char* ab_ = (s1_ ? abase1 : abase0) + (size_t)(K0) * 2;                                             \
        _Pragma("unroll") for (int j_ = (J0); j_ < (J1); ++j_)                                                    \
            glds16(reinterpret_cast<const bf16_t*>(ab_ + (size_t)(s1_ ? aoff[KC ? 1 : 0][j_] : aoff[0][j_])),     \
                   smem + AOFF + (BUF) * 256 * 64 + arowl[j_] * 64);                                              \
    } while (0)

    // ---- L2 prefetch of the weight panel (flags bit 1).  In the 24-layer stack every launch streams its weights from HBM for the
    // first time; all the tiles that share a weight panel run in lockstep, so an HBM miss of the one-phase-ahead DMA stalls every one
    // of them.  Every second k-tile each thread touches one dword of one 128-B weight line of the k-tiles PFD and PFD + 1 ahead
    // (256 rows x 2 k-tiles = 512 lines = 512 threads): the DMA issued two k-tiles later then hits L2.  The load is issued LAST in
    // its phase, so that the counted vmcnt waits that follow leave it in flight for one whole k-tile (VMEM returns in order).
    constexpr int PFD = 3;
    const bool pf_on = (g.flags & 2) != 0 && nk > PFD && !g.wtiled;
    const char* pfptr;
    {
        int n = n0 + (tid & 255);
        n = n < g.N ? n : g.N - 1;
        pfptr = wbase + (size_t)n * g.ldw * 2 + (size_t)(tid >> 8) * (BK * 2);
    }
    uint32_t pfreg = 0;
#define Q2_PF(T)                                                                                                  \
    do {                                                                                                          \
        int kt_ = (T) + PFD;                                                                                      \
        kt_ = kt_ + 1 < nk ? kt_ : nk - 2; /* the pair (kt_, kt_ + 1) stays inside the row */                     \
        asm volatile("global_load_dword %0, %1, off" : "+v"(pfreg) : "v"(pfptr + (size_t)kt_ * (BK * 2)) : "memory"); \
    } while (0)

    // ---- fragment read addresses (elements).  row & 7 == fr & 7 for every fragment row of this lane.
    const int fr = lane & 15, fg = lane >> 4;
    const int lsw0 = fr * 64 + ((fg ^ (fr & 7)) << 3);        // k-step 0 chunk
    const int lsw1 = fr * 64 + (((fg + 4) ^ (fr & 7)) << 3);  // k-step 1 chunk
    const bf16_t* ldsW = smem + (wn * 64) * 64;
    const bf16_t* ldsA = smem + AOFF + gbase * 64;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NAF = NS ? MF0 : 4;  // A fragments live at once
    bf16x8 wf[2][4], af[2][NAF];

#define Q2_READ_W(BUF, I0, I1)                                                                                    \
    _Pragma("unroll") for (int i = (I0); i < (I1); ++i) {                                                         \
        wf[0][i] = *reinterpret_cast<const bf16x8*>(ldsW + (BUF) * 256 * 64 + i * 16 * 64 + lsw0);                \
        wf[1][i] = *reinterpret_cast<const bf16x8*>(ldsW + (BUF) * 256 * 64 + i * 16 * 64 + lsw1);                \
    }
#define Q2_READ_A(BUF, MB, CNT)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < (CNT); ++j) {                                                           \
        af[0][j] = *reinterpret_cast<const bf16x8*>(ldsA + (BUF) * 256 * 64 + ((MB) + j) * 16 * 64 + lsw0);       \
        af[1][j] = *reinterpret_cast<const bf16x8*>(ldsA + (BUF) * 256 * 64 + ((MB) + j) * 16 * 64 + lsw1);       \
    }
    // W fragments [I0, I1) x A fragments af[0 .. CNT) -> acc[i][MB + j]
#define Q2_MFMA(I0, I1, MB, CNT)                                                                                  \
    do {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                          \
            _Pragma("unroll") for (int i = (I0); i < (I1); ++i)                                                   \
                _Pragma("unroll") for (int j = 0; j < (CNT); ++j)                                                 \
                    acc[i][(MB) + j] = Op16<F16>::mfma16(wf[kk][i], af[kk][j], acc[i][(MB) + j]);                 \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)
#define Q2_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
    // m-split k-tile
#define Q2_TILE(BUF, T, MFG)                                                                                         \
    do {                                                                                                          \
        const int kN = ((T) + 1) * BK;                                                                            \
        const bool has1 = (T) + 1 < nk;                                                                           \
        /* ph0: all W fragments + the 4 lo A fragments */                                                         \
        Q2_READ_W(BUF, 0, 4)                                                                                      \
        Q2_READ_A(BUF, 0, 4)                                                                                      \
        if (has1) {                                                                                               \
            Q2_DMA_W((BUF) ^ 1, 0, kN);                                                                           \
            Q2_DMA_W((BUF) ^ 1, 1, kN);                                                                           \
            Q2_DMA_A((BUF) ^ 1, 0, 2, kN);                                                                        \
            /* retires the hi pieces of this tile; an odd tile leaves the prefetch of the previous tile in flight */ \
            if ((BUF) == 1 && pf_on && (T) - 1 + PFD < nk) Q2_WAIT(7); else Q2_WAIT(6);                            \
        } else {                                                                                                  \
            Q2_WAIT(0);                                                                                           \
        }                                                                                                         \
        bar_raw_fn();                                                                                             \
        Q2_MFMA(0, 4, 0, 4);                                                                                      \
        bar_raw_fn();                                                                                             \
        /* ph1: the hi A fragments of this group */                                                               \
        Q2_READ_A(BUF, 4, (MFG) - 4)                                                                              \
        if (has1) {                                                                                               \
            Q2_DMA_A((BUF) ^ 1, 2, 4, kN);                                                                        \
            /* retires W + A-lo of tile T+1 (and, in an odd tile, the prefetch issued one k-tile ago) */          \
            if ((BUF) == 0 && pf_on && (T) + PFD < nk) { Q2_PF(T); Q2_WAIT(3); } else Q2_WAIT(2);                  \
        } else {                                                                                                  \
            Q2_WAIT(0);                                                                                           \
        }                                                                                                         \
        bar_raw_fn();                                                                                             \
        Q2_MFMA(0, 4, 4, (MFG) - 4);                                                                              \
        bar_raw_fn();                                                                                             \
    } while (0)
    // n-split k-tile
#define N2_TILE(BUF, T, MFG)                                                                                         \
    do {                                                                                                          \
        const int kN = ((T) + 1) * BK;                                                                            \
        const bool has1 = (T) + 1 < nk;                                                                           \
        /* ph0: W fragments 0,1 + every A fragment of this group */                                               \
        Q2_READ_W(BUF, 0, 2)                                                                                      \
        Q2_READ_A(BUF, 0, MFG)                                                                                    \
        if (has1) {                                                                                               \
            Q2_DMA_W((BUF) ^ 1, 0, kN);                                                                           \
            Q2_DMA_A((BUF) ^ 1, 0, NPW, kN);                                                                      \
            /* retires the W rows of fragments 2,3 of this tile */                                                \
            if ((BUF) == 1 && pf_on && (T) - 1 + PFD < nk) Q2_WAIT(3 + NPW); else Q2_WAIT(2 + NPW);                \
        } else {                                                                                                  \
            Q2_WAIT(0);                                                                                           \
        }                                                                                                         \
        bar_raw_fn();                                                                                             \
        Q2_MFMA(0, 2, 0, MFG);                                                                                    \
        bar_raw_fn();                                                                                             \
        /* ph1: W fragments 2,3 */                                                                                \
        Q2_READ_W(BUF, 2, 4)                                                                                      \
        if (has1) {                                                                                               \
            Q2_DMA_W((BUF) ^ 1, 1, kN);                                                                           \
            /* retires W 0,1 + A of tile T+1 */                                                                   \
            if ((BUF) == 0 && pf_on && (T) + PFD < nk) { Q2_PF(T); Q2_WAIT(3); } else Q2_WAIT(2);                  \
        } else {                                                                                                  \
            Q2_WAIT(0);                                                                                           \
        }                                                                                                         \
        bar_raw_fn();                                                                                             \
        Q2_MFMA(2, 4, 0, MFG);                                                                                    \
        bar_raw_fn();                                                                                             \
    } while (0)

    // ---- prologue: all of tile 0
    Q2_DMA_W(0, 0, 0);
    Q2_DMA_W(0, 1, 0);
    Q2_DMA_A(0, 0, NAO, 0);
    Q2_WAIT(0);
    bar_raw_fn();
    if (wm == 1) bar_raw_fn();  // group 1 runs one barrier behind group 0

    // the two groups run separate copies of the loop when their fragment counts differ (same barrier count in both)
#define Q2_RUN(MFG)                                                                                               \
    do {                                                                                                          \
        int t = 0;                                                                                                \
        if (NS) {                                                                                                 \
            for (; t + 1 < nk; t += 2) {                                                                          \
                N2_TILE(0, t, MFG);                                                                               \
                N2_TILE(1, t + 1, MFG);                                                                           \
            }                                                                                                     \
            if (t < nk) N2_TILE(0, t, MFG);                                                                       \
        } else {                                                                                                  \
            for (; t + 1 < nk; t += 2) {                                                                          \
                Q2_TILE(0, t, MFG);                                                                               \
                Q2_TILE(1, t + 1, MFG);                                                                           \
            }                                                                                                     \
            if (t < nk) Q2_TILE(0, t, MFG);                                                                       \
        }                                                                                                         \
    } while (0)
    __shared__ int s_last;
    __shared__ int s_ml[2];  // splitk_coop_finish's mode / last-arriver flags: shared by BOTH wave groups
    constexpr int NFS = 4 * (MF0 > MF1 ? MF0 : MF1);
    if (MF0 == MF1 || wm == 0) {
        Q2_RUN(MF0);
        if (wm == 0) bar_raw_fn();  // re-align the barrier counts of the two groups
        if constexpr (EPI != EPI_QKV && EPI != EPI_QKV_SPLIT) {
            if (g.splits > 1 && g.coop) { splitk_coop_finish<EPI, MF0, NFS, F16>(g, acc, tm * tilesN + tn, split, n0, wn, m0 + gbase, fr, fg, s_ml); return; }
        }
        if (g.splits > 1 && !splitk_exchange<MF0, NFS>(g, acc, tm * tilesN + tn, split, &s_last)) return;
        if constexpr (EPI == EPI_QKV_SPLIT) epilogue_qkv_split<MF0>(g, acc, n0, wn, m0 + gbase, fr, fg, g_stage ? smem + wave * 8192 : nullptr);
        else epilogue8p<EPI, MF0, F16>(g, acc, n0, wn, m0 + gbase, fr, fg, g_stage ? smem + wave * 8192 : nullptr);
    } else {
        Q2_RUN(MF1);
        if constexpr (EPI != EPI_QKV && EPI != EPI_QKV_SPLIT) {
            if (g.splits > 1 && g.coop) { splitk_coop_finish<EPI, MF1, NFS, F16>(g, acc, tm * tilesN + tn, split, n0, wn, m0 + gbase, fr, fg, s_ml); return; }
        }
        if (g.splits > 1 && !splitk_exchange<MF1, NFS>(g, acc, tm * tilesN + tn, split, &s_last)) return;
        if constexpr (EPI == EPI_QKV_SPLIT) epilogue_qkv_split<MF1>(g, acc, n0, wn, m0 + gbase, fr, fg, g_stage ? smem + wave * 8192 : nullptr);
        else epilogue8p<EPI, MF1, F16>(g, acc, n0, wn, m0 + gbase, fr, fg, g_stage ? smem + wave * 8192 : nullptr);
    }
#undef Q2_RUN
    asm volatile("" ::"v"(pfreg));  // keeps the prefetch destination register reserved for the whole loop
#undef N2_TILE
#undef Q2_TILE
#undef Q2_WAIT
#undef Q2_MFMA
#undef Q2_READ_A
#undef Q2_READ_W
#undef Q2_DMA_A
#undef Q2_PF
#undef Q2_DMA_W
}

template <int EPI, int MF0, int MF1, bool NS, bool F16>
int launch2p(const GemmArgs& g0, hipStream_t s) {
    GemmArgs g = g0;
    static bool attr_set = false;
    auto kfn = gemm2p_kernel<EPI, MF0, MF1, NS, F16>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM3_BYTES);
        if (e != hipSuccess) return set_error_hip(e, "hipFuncSetAttribute(gemm2p)", __FILE__, __LINE__);
        attr_set = true;
    }
    constexpr int BMT = 16 * (MF0 + MF1);
    int tilesM = (g.M + BMT - 1) / BMT, tilesN = (g.N + B2 - 1) / B2;
    const int tiles = tilesM * tilesN, nk = g.K / GEMM_BK;
    // split-K: the split count S (hence the k-partition and the fp32 summation order) is a function of (M, N, K) ALONE
    // (splitk_count below): every split-capable tile variant produces the same bits, whichever one the tuner picks.
    g.splits = 1;
    const int S = splitk_count(g.M, g.N, g.K, showo_cu_usable((void*)s));
    if (S >= 2) {
        constexpr int NFS = 4 * (MF0 > MF1 ? MF0 : MF1);
        if (tiles * S > SPLITK_TICKS || !splitk_ws(s, (size_t)tiles * S * NFS * 512 * sizeof(float4), &g.ws, &g.tick))
            return set_error_msg(7, "gemm2p: split-K workspace unavailable (first use of a split shape inside a stream capture, or more than 8 "
                                    "streams): run the shape once eagerly, or set SHOWO_GEMM_SPLITK=0");
        g.splits = S;
        g.coop = (EPI != EPI_QKV && EPI != EPI_QKV_SPLIT && tiles <= 2048 && splitk_coop_ok(tiles * S, s)) ? splitk_coop_mode() : 0;
        g_cnt_splitk++;
    }
    kfn<<<dim3(tiles * g.splits), dim3(512), SMEM3_BYTES, s>>>(g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error_hip(e, "gemm2p launch", __FILE__, __LINE__);
    if (g.coop) splitk_coop_launched(s);
    return 0;
}

template <int EPI, bool F16>
int launch2p_h(const GemmArgs& g, int h, hipStream_t s) {
    if (h >= 2000) return gemm3w_launch(g, EPI, h - 2000, s);  // (reads g.op)
    switch (h) {
        case 240: return launch2p<EPI, 8, 7, false, F16>(g, s);
        case 224: return launch2p<EPI, 7, 7, false, F16>(g, s);
        case 208: return launch2p<EPI, 7, 6, false, F16>(g, s);
        case 176: return launch2p<EPI, 6, 5, false, F16>(g, s);
        case 160: return launch2p<EPI, 5, 5, false, F16>(g, s);
        case 144: return launch2p<EPI, 5, 4, false, F16>(g, s);
        case 1192: return launch2p<EPI, 6, 6, true, F16>(g, s);
        case 1176: return launch2p<EPI, 6, 5, true, F16>(g, s);
        case 1160: return launch2p<EPI, 5, 5, true, F16>(g, s);
        case 1144: return launch2p<EPI, 5, 4, true, F16>(g, s);
        case 1128: return launch2p<EPI, 4, 4, true, F16>(g, s);
    }
    return launch2p<EPI, 8, 8, false, F16>(g, s);
}


}  // namespace g2p
}  // namespace showo
