// gemm4h: the production bf16 GEMM (C = A W^T, gemm_common.h epilogues) as a FOUR-wave kernel with 128 x 128 wave tiles -- the structural
// experiment of round 4.  Selectable (variant 5256: SHOWO_GEMM_BM=5256, or SHOWO_GEMM_4H=1 for the tuner), NOT the default.
//
// Hypothesis (profiles/r3n_lds_probe.txt): the 8-wave kernels (gemm2p / gemm3w: wave tile 64 x 96, two waves per SIMD alternating MFMA
// and load segments) are bound by LDS read INSTRUCTIONS per MFMA (0.42; the LDS retires about one wave-instruction per 7 cycles per
// CU).  A 128 x 128 wave tile needs 0.25: per 64-deep k-tile a wave issues 128 MFMAs (2 048 issue cycles on its SIMD), 32 ds_read_b128
// and 16 LDS-DMA pieces.  The price is ONE wave per SIMD (256 accumulator registers per lane): nothing hides a stall but the wave's
// own instruction stream, so the stream is laid out explicitly -- every ds_read and every DMA sits between MFMAs, the fragments of
// the next k-slab are read while the current slab multiplies (two register sets), one raw s_barrier per k-tile.
//
// How the stream is pinned.  With MFMA builtins the register allocator moved accumulator quads between VGPRs and AGPRs inside the loop
// (1 120 v_accvgpr moves per 6 k-tiles) and __builtin_amdgcn_sched_group_barrier pipelines were only partly honoured.  Here the MFMAs
// are `asm volatile` statements with the accumulator as a "+a" operand: the accumulators live in AGPRs for the whole kernel, volatile
// statements keep their order and memory operations (ds_read, the DMA builtin) are not moved across them, so the SOURCE order is the
// instruction stream; the compiler still allocates registers and inserts the counted lgkmcnt waits.  tools/check_gemm4h_isa.py audits
// the .s: 768 MFMA / 192 ds_read / 96 DMA per 6 k-tiles in the pattern MMMR MMMR D, 0 spills, 0 v_accvgpr, only the hand-placed
// vmcnt(8).
//
// Result (profiles/r4b_gemm4h_harness.txt, one MI355X, cold weights): correct -- bit-identical to the 8-wave family, 178 GEMM tests
// green with the variant forced -- and NOT faster: 1 285 TF/s at 4096^3 (8-wave 256^2: 1 313), 1 035-1 101 on (4128, 14336, 2048) with
// a plain epilogue (3192: 1 146), 655-670 with the fused QKV epilogue (998: 226 spilled VGPRs in an epilogue nothing overlaps), half
// the chip idle on N = 2 048 launches (136 tiles).  A k-tile takes ~3 200 cycles against 2 048 of MFMA issue: the 16 LDS-DMA pieces
// a wave issues per k-tile cost it ~70 cycles each (MI355X_MICROARCH.md "LDS-DMA piece issue cost"), and with one wave per SIMD that
// time comes straight out of the MFMA stream -- the two-waves-per-SIMD kernels hide it behind the partner's MFMA segment.  LDS read
// instructions were not the limiter; the HBM -> LDS path is.  (Round 3's `gemm4w`: same tile, VGPR staging, compiler order: 630-908.)
//
// Geometry: 256 threads = 4 waves as 2 (m) x 2 (n); block tile 256 (n) x 32 MJ (m), MJ = 16-row fragments per wave (8: 256 rows);
// v_mfma_f32_16x16x32_bf16 with the weight fragment as the A operand, k order as in gemm2p (slab 0, slab 1 of every k-tile in turn):
// every output element is the same fp32 chain -> bit-identical results to every other variant of the production family.
// LDS (160 KiB): W ring 3 x 32 KiB at 0 / 32 / 64 KiB, A ring 2 x 32 KiB at 96 / 128 KiB; images [rows][64] bf16 with the 16-byte
// chunk c of row r at position c ^ (r & 7) (swizzle on the DMA source side or in the pre-tiled weight layout; conflict-free
// ds_read_b128).  Tile T reads W[T % 3], A[T % 2].
//
// Schedule of k-tile T (F0 / F1 = the two fragment register sets; part = 64 MFMAs = one 32-deep k-slab of the 128 x 128 tile):
//   part 1: MFMA slab 0 (F0)  ||  ds_read slab 1 of tile T -> F1 (16 reads)  ||  DMA W(T+2) -> W[(T+2) % 3] (8 pieces per wave)
//           s_waitcnt lgkmcnt(0) ; s_waitcnt vmcnt(8) ; s_barrier                                                  [B_T]
//   part 2: MFMA slab 1 (F1)  ||  ds_read slab 0 of tile T+1 -> F0           ||  DMA A(T+2) -> A[T % 2]      (MJ pieces per wave)
// Hazards.  RAW: VMEM returns in order, so vmcnt(8) at B_T leaves only this part's W(T+2) pieces in flight: A(T+1) (issued in part 2
// of T-1, a full part earlier) and W(T+1) (part 1 of T-1) have landed in every wave's view once B_T is passed, and tile T+1 is first
// read in part 2 of T.  WAR: W[(T+2) % 3] = W[(T-1) % 3] was last read in part 1 of T-1 and those reads retired before B_(T-1);
// A[T % 2] is last read in part 1 of T (retired by the lgkmcnt(0) in front of B_T) and re-staged after B_T.
// The last two tiles have nothing to prefetch: their DMAs re-fetch the last k-tile into ring slots nobody reads again (no branch in
// the stream; 2 / nk extra L2 reads), and a vmcnt(0) + barrier in front of the epilogue keeps them out of the epilogue's staging.
#include "gemm_common.h"

namespace showo {
namespace {

__device__ __forceinline__ void h_bufl16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, bf16_t* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// One part of a k-tile: the 8 MJ MFMAs of a 32-deep k-slab on the fragment set (WC, AC), with the 8 + MJ ds_reads of the NEXT slab
// (into WN, AN) and ND DMA pieces placed between them IN SOURCE ORDER.  The MFMAs are `asm volatile` with the accumulator as a "+a"
// operand: (1) the accumulators live in AGPRs for the whole kernel -- with the builtin the register allocator moved accumulator quads
// between VGPRs and AGPRs inside the loop (1 120 v_accvgpr moves per 6 k-tiles at MJ = 8) --, (2) volatile statements keep their
// order, and memory operations (ds_read, the DMA builtin) are not moved across them, so the stream below IS the instruction stream;
// the compiler still allocates registers and inserts the counted lgkmcnt waits in front of the first MFMA that uses a fragment.
// Read order = use order of the next part (i-major MFMAs: W fragment 0 and every A fragment first).
template <int MJ, int ND, class RdW, class RdA, class Dma>
__device__ __forceinline__ void h_part(f32x4 (&acc)[8][8], const bf16x8 (&WC)[8], const bf16x8 (&AC)[MJ], bf16x8 (&WN)[8], bf16x8 (&AN)[MJ],
                                       RdW rdw, RdA rda, Dma dma) {
    constexpr int NM = 8 * MJ, NR = 8 + MJ;
    constexpr int MPR = (NM * 3 / 4) / NR > 0 ? (NM * 3 / 4) / NR : 1;  // MFMAs between two reads; the last quarter of the part is MFMA only
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m / MJ][m % MJ]) : "v"(WC[m / MJ]), "v"(AC[m % MJ]));
        if ((m + 1) % MPR == 0 && (m + 1) / MPR <= NR) {
            const int r = (m + 1) / MPR - 1;  // 0 .. NR-1
            if (r == 0) WN[0] = rdw(0);
            else if (r <= MJ) AN[r - 1] = rda(r - 1);
            else WN[r - MJ] = rdw(r - MJ);
            if ((r * ND) / NR != ((r + 1) * ND) / NR) dma((r * ND) / NR);
        }
    }
}

template <int EPI, int MJ>
__device__ __forceinline__ void gemm4h_body(const GemmArgs& g, bf16_t* smem, int tn, int m0) {
    constexpr int BK = GEMM_BK;
    constexpr int WBUF = 256 * 64;      // elements per ring slot (32 KiB), both operands
    constexpr int AOFF = 3 * WBUF;
    constexpr bool KC = (EPI == SHOWO_EPI_RESID_F32);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn2 = wave & 1, wm = wave >> 1;
    const int n0 = tn * B2;
    const int nk = g.K / BK;
    const int srow = lane >> 3;
    const int coff = ((lane & 7) ^ srow) << 3;
    const int Ks = KC ? g.Ksplit : (1 << 30);
    const int wks = g.wtiled ? 8 : 0;
    const char* wbase = reinterpret_cast<const char*>(g.W) + (g.wtiled ? (size_t)tn * (size_t)(g.K / BK) * 32768 : (size_t)0);
    const char* abase0 = reinterpret_cast<const char*>(g.A);
    const char* abase1 = (KC && g.A2) ? reinterpret_cast<const char*>(g.A2) - (int64_t)Ks * 2 : abase0;
    const int lda1 = (KC && g.A2) ? g.lda2 : g.lda;
    // DMA pieces (8 rows x 128 B each): wave w stages W rows [64 w, 64 w + 64) and A rows [8 MJ w, 8 MJ w + 8 MJ) of every k-tile
    uint32_t woff[8], aoff[KC ? 2 : 1][MJ];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int row = wave * 64 + p * 8;
        int n = n0 + row + srow;
        n = n < g.N ? n : g.N - 1;
        woff[p] = g.wtiled ? (uint32_t)(row * 128 + lane * 16) : (uint32_t)(((int64_t)n * g.ldw + coff) * 2);
    }
#pragma unroll
    for (int p = 0; p < MJ; ++p) {
        int m = m0 + wave * 8 * MJ + p * 8 + srow;
        m = m < g.M ? m : g.M - 1;
        aoff[0][p] = (uint32_t)(((int64_t)m * g.lda + coff) * 2);
        if (KC) aoff[KC ? 1 : 0][p] = (uint32_t)(((int64_t)m * lda1 + coff) * 2);
    }
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbase), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(abase0), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(abase1), 0, -1, 0x00020000);
    bf16_t* const dmaW = smem + wave * 64 * 64;
    bf16_t* const dmaA = smem + AOFF + wave * 8 * MJ * 64;

    const int fr = lane & 15, fg = lane >> 4;
    const int lsw0 = fr * 64 + ((fg ^ (fr & 7)) << 3);
    const int lsw1 = fr * 64 + (((fg + 4) ^ (fr & 7)) << 3);
    const bf16_t* ldsW = smem + (wn2 * 128) * 64;
    const bf16_t* ldsA = smem + AOFF + (wm * 16 * MJ) * 64;

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 wf0[8], af0[MJ], wf1[8], af1[MJ];

    // one k-tile; WB = T % 3, AB = T % 2 are literals (the compiler can tell the ring slots apart: no conservative waits)
#define H_TILE(WB, AB, T)                                                                                         \
    do {                                                                                                          \
        const int t2_ = (T) + 2 < nk ? (T) + 2 : nk - 1;                                                          \
        const uint32_t sow_ = (uint32_t)((t2_ * BK * 2) << wks);                                                  \
        const int ka_ = t2_ * BK;                                                                                 \
        const bool s1_ = KC && ka_ >= Ks;                                                                         \
        const __amdgpu_buffer_rsrc_t rsA_ = s1_ ? rsA1 : rsA0;                                                    \
        /* part 1: slab 0 multiplies, slab 1 of this tile is read, W(T+2) is requested */                         \
        h_part<MJ, 8>(acc, wf0, af0, wf1, af1,                                                                    \
            [&](int i) { return *reinterpret_cast<const bf16x8*>(ldsW + (WB) * WBUF + i * 16 * 64 + lsw1); },     \
            [&](int j) { return *reinterpret_cast<const bf16x8*>(ldsA + (AB) * WBUF + j * 16 * 64 + lsw1); },     \
            [&](int p) { h_bufl16(rsW, woff[p], sow_, dmaW + (((WB) + 2) % 3) * WBUF + p * 8 * 64); });           \
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                               \
        bar_raw_fn();                                                                                             \
        /* part 2: slab 1 multiplies, slab 0 of tile T+1 is read, A(T+2) is requested */                          \
        h_part<MJ, MJ>(acc, wf1, af1, wf0, af0,                                                                   \
            [&](int i) { return *reinterpret_cast<const bf16x8*>(ldsW + (((WB) + 1) % 3) * WBUF + i * 16 * 64 + lsw0); }, \
            [&](int j) { return *reinterpret_cast<const bf16x8*>(ldsA + ((AB) ^ 1) * WBUF + j * 16 * 64 + lsw0); }, \
            [&](int p) { h_bufl16(rsA_, s1_ ? aoff[KC ? 1 : 0][p] : aoff[0][p], (uint32_t)(ka_ * 2), dmaA + (AB) * WBUF + p * 8 * 64); }); \
    } while (0)

    // ---- prologue: tile 0 (W, A), then W(1), then A(1): vmcnt(MJ + 8) retires tile 0; barrier; slab 0 of tile 0 -> F0
    {
        const int t1 = nk > 1 ? 1 : 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) h_bufl16(rsW, woff[p], 0u, dmaW + p * 8 * 64);
#pragma unroll
        for (int p = 0; p < MJ; ++p) h_bufl16(rsA0, aoff[0][p], 0u, dmaA + p * 8 * 64);
#pragma unroll
        for (int p = 0; p < 8; ++p) h_bufl16(rsW, woff[p], (uint32_t)((t1 * BK * 2) << wks), dmaW + WBUF + p * 8 * 64);
        const bool s1 = KC && t1 * BK >= Ks;
#pragma unroll
        for (int p = 0; p < MJ; ++p)
            h_bufl16(s1 ? rsA1 : rsA0, s1 ? aoff[KC ? 1 : 0][p] : aoff[0][p], (uint32_t)(t1 * BK * 2), dmaA + WBUF + p * 8 * 64);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + MJ) : "memory");
    bar_raw_fn();
#pragma unroll
    for (int i = 0; i < 8; ++i) wf0[i] = *reinterpret_cast<const bf16x8*>(ldsW + i * 16 * 64 + lsw0);
#pragma unroll
    for (int j = 0; j < MJ; ++j) af0[j] = *reinterpret_cast<const bf16x8*>(ldsA + j * 16 * 64 + lsw0);
    // B_0 waits with vmcnt(8): the 8 newest pieces are W(2); A(1) and W(1) are older -> landed.  From tile 1 on the steady state.
    {
        int t = 0;
        for (; t + 5 < nk; t += 6) {
            H_TILE(0, 0, t);
            H_TILE(1, 1, t + 1);
            H_TILE(2, 0, t + 2);
            H_TILE(0, 1, t + 3);
            H_TILE(1, 0, t + 4);
            H_TILE(2, 1, t + 5);
        }
        if (t < nk) H_TILE(0, 0, t);
        if (t + 1 < nk) H_TILE(1, 1, t + 1);
        if (t + 2 < nk) H_TILE(2, 0, t + 2);
        if (t + 3 < nk) H_TILE(0, 1, t + 3);
        if (t + 4 < nk) H_TILE(1, 0, t + 4);
    }
    // the tail's re-fetches must not land in the epilogue's staging slices; the s_nops cover the MFMA -> VALU read distance the
    // compiler cannot see through the asm statements (16 passes of the last MFMA)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");
    bar_raw_fn();
    const bool g_stage = !(g.flags & 8);
    bf16_t* stg = g_stage ? smem + wave * 8192 : nullptr;
    const int mrow0 = m0 + wm * 16 * MJ;
    epilogue8p<EPI, MJ>(g, reinterpret_cast<f32x4(&)[4][8]>(acc[0]), n0, wn2 * 2, mrow0, fr, fg, stg);
    epilogue8p<EPI, MJ>(g, reinterpret_cast<f32x4(&)[4][8]>(acc[4]), n0, wn2 * 2 + 1, mrow0, fr, fg, stg);
#undef H_TILE
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm4h_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    constexpr int BMT = 256;
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
    const int m0 = tm * BMT;
    // the last tile row of a ragged M runs with as many 16-row fragments per wave as it has rows for (M = 4 128: 32 rows -> MJ = 1 of
    // the 2-fragment body): the short body costs a quarter of a full tile instead of a whole one
    const int rows = g.M - m0;
    if (rows > 128) gemm4h_body<EPI, 8>(g, smem, tn, m0);
    else if (rows > 64) gemm4h_body<EPI, 4>(g, smem, tn, m0);
    else gemm4h_body<EPI, 2>(g, smem, tn, m0);
}

constexpr int SMEM4H_BYTES = 5 * 256 * 64 * 2;  // 160 KiB

template <int EPI>
int launch4h(const GemmArgs& g, hipStream_t s) {
    static bool attr_set = false;
    auto kfn = gemm4h_kernel<EPI>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM4H_BYTES);
        if (e != hipSuccess) return set_error_hip(e, "hipFuncSetAttribute(gemm4h)", __FILE__, __LINE__);
        attr_set = true;
    }
    const int tilesM = (g.M + 255) / 256, tilesN = (g.N + B2 - 1) / B2;
    kfn<<<dim3(tilesM * tilesN), dim3(256), SMEM4H_BYTES, s>>>(g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error_hip(e, "gemm4h launch", __FILE__, __LINE__);
    return 0;
}

}  // namespace

// variant code 5256 of gemm2p's tile table
int gemm4h_launch(const GemmArgs& g, int epilogue, hipStream_t s) {
    if (g.K % GEMM_BK) return set_error_msg(1, "gemm4h: K must be a multiple of 64");
    switch (epilogue) {
        case SHOWO_EPI_BF16: return launch4h<SHOWO_EPI_BF16>(g, s);
        case SHOWO_EPI_GELU_BF16: return launch4h<SHOWO_EPI_GELU_BF16>(g, s);
        case SHOWO_EPI_F32: return launch4h<SHOWO_EPI_F32>(g, s);
        case SHOWO_EPI_RESID_F32: return launch4h<SHOWO_EPI_RESID_F32>(g, s);
        case EPI_QKV: return launch4h<EPI_QKV>(g, s);
    }
    return set_error_msg(1, "gemm4h: unknown epilogue");
}

}  // namespace showo
