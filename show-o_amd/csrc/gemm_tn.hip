// Weight-gradient GEMM on token-major operands ("TN"): out[M, N] (fp32) = A^T B with A = dY bf16 [T, M] (lda) and B = X bf16 [T, N]
// (ldw), the contraction running over the TOKEN rows.  dW = dY^T X of every Linear of the training step
// (training/train.py:612 loss.backward(): autograd of F.linear, models/phi.py:657-659, 727, 208-212, 1182).
//
// The production kernel (gemm2p.hip) wants both operands k-contiguous, so the trainer used to transpose dY and X first (243 launches,
// 8.7 ms of pure HBM traffic per step).  Here the 64-token k-tile is staged the way it lies in memory -- LDS image [64 t][256 columns],
// rows of consecutive COLUMNS -- and the MFMA fragments (a lane needs 8 consecutive t of ONE column) are gathered by
// ds_read_b64_tr_b16, gfx950's transposing LDS read: each 16-lane group reads a [4 t][16 columns] block (lane i supplies the 8-byte
// address of row i >> 2, columns 4 (i & 3) ..) and receives it transposed (lane c gets the 4 t of column c); two reads make the
// 8-deep k slice of a v_mfma_f32_16x16x32_bf16 operand.
//
// Everything else is gemm2p's m-split program at its tallest tile: 256 x 256 x 64, 8 waves = 4 along n x 2 groups along m, the groups
// one barrier apart, operands HBM -> LDS by global_load_lds with counted vmcnt, double-buffered; same accumulator layout, so the
// split-K exchange and the epilogue are the shared ones (gemm_common.h).
//
// LDS image of one operand tile (32 KiB): 32 DMA pieces of 1 KiB; piece (tq, cq) = token rows 8 tq .. 8 tq + 7 x columns 64 cq ..
// 64 cq + 63, row tl at byte 128 tl, 16-byte unit cu (8 columns) at position cu ^ swz(tl, tq), swz = ((tl >> 1) & 1) << 1 | (tq & 1) << 2:
// the 32 lanes serviced together by a transposing read (rows 4 h .. 4 h + 3 of pieces tq and tq + 1, one 32-byte block each) then cover
// eight distinct 32-byte bank slots.  The swizzle is applied to the DMA SOURCE column (the LDS side of the DMA is lane-linear).
// Wave w stages the token rows 8 w .. 8 w + 7 of both tiles: 128-byte global segments, 8 rows per wave-instruction, like gemm2p.
// Token rows >= Klim (the zero padding of T up to a multiple of 64) and columns >= M / N are fetched from a zero page.
#include "gemm_common.h"
#include "prof.h"
#include <cstdlib>

namespace showo {

__device__ __attribute__((aligned(16))) unsigned int g_tn_zero_page[128];  // 512 B: a lane of the zero-row path reads at offsets < 512

namespace {

typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 llvm_bf16x4;

// 8 consecutive t of one column = two transposing reads 4 rows (512 B) apart
static __device__ __forceinline__ bf16x8 tr_read8(const bf16_t* p) {
    const llvm_bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) llvm_bf16x4*)(p));
    const llvm_bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) llvm_bf16x4*)(p + 4 * 64));
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}

constexpr int TN_TILE = 64 * 256;  // elements of one operand tile (32 KiB)

// The unchecked kernels issue their DMAs from inline assembly (uniform 64-bit base in SGPRs + 32-bit per-lane byte offset, LDS base in
// M0): with __builtin_amdgcn_global_load_lds the compiler's wait-count pass sees LDS writes in flight and puts an s_waitcnt vmcnt(0) in
// front of every ds_read_b64_tr_b16 (it cannot prove that the transposing read does not alias them), which drains the DMA queue once
// per phase -- the opposite of the counted-vmcnt schedule.  Ordering is ours anyway: counted waits + barriers, as in gemm2p.
static __device__ __forceinline__ void glds16_sa(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

// EDGE = false (the shapes of the training step): every 256-column tile of both operands lies inside a row (lda >= 256 tilesM, ldb >=
// 256 tilesN) -- columns beyond M / N are real memory whose products land in rows / columns that are never stored -- and the operands
// are addressed as base + 32-bit byte offset; only the ONE k-tile that contains token rows >= Klim takes the checked DMA form (rows
// beyond Klim from a zero page).  EDGE = true: every DMA checks its row and its 8-column unit (any lda / ldb that covers M / N rounded
// up to 8).
// RING = true (launches that do not split K, EDGE = false): gemm3w's schedule -- a THREE-deep LDS ring for the B ("W") tile, so that the
// pieces of tile T + 2 are issued half in ph0 and half in ph1 of tile T (4 DMA pieces per load segment instead of 6 + 2) and the
// operand stream has 1.5 k-tiles of lead.  LDS: W ring 3 x 32 KiB, A double buffer 2 x 32 KiB = 160 KiB.
template <int EPI, bool EDGE, bool RING>
__global__ __launch_bounds__(512) void gemm_tn_kernel(GemmArgs g) {
    static_assert(!(RING && EDGE), "the ring form is built for the unchecked shapes only");
    constexpr int BK = GEMM_BK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesM = (g.M + 255) / 256, tilesN = (g.N + B2 - 1) / B2;
    int nwg = tilesM * tilesN, bid = blockIdx.x;
    int split = 0;
    if (g.splits > 1) { split = bid / nwg; bid -= split * nwg; }
    {   // XCD-aware bijective remap (blocks with equal id % 8 share an L2), as in gemm2p
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
    const int m0 = tm * 256, n0 = tn * B2;
    int nk = g.K / BK;
    int kb = 0;  // first token row of this block's range (split-K)
    if (g.splits > 1) {
        const int per = (nk + g.splits - 1) / g.splits;
        kb = split * per * BK;
        nk = min(nk - split * per, per);
    }
    const int wn = wave & 3, wm = wave >> 2;
    const int gbase = wm * 128;

    // ---- DMA roles: this wave stages token rows 8 wave .. 8 wave + 7 of every k-tile; lane -> row tl, unit position lane & 7
    const int tl = lane >> 3;
    const int cu = (lane & 7) ^ ((((tl >> 1) & 1) << 1) | ((wave & 1) << 2));  // logical unit fetched into this lane's position
    const int trow0 = kb + wave * 8 + tl;
    const char* wbase = reinterpret_cast<const char*>(g.W);
    const char* abase = reinterpret_cast<const char*>(g.A);
    const uint32_t woff0 = (uint32_t)(((int64_t)trow0 * g.ldw + n0 + 8 * cu) * 2), aoff0 = (uint32_t)(((int64_t)trow0 * g.lda + m0 + 8 * cu) * 2);
    const uint32_t wstep = (uint32_t)(BK * g.ldw * 2), astep = (uint32_t)(BK * g.lda * 2);
    unsigned wok = 0xf, aok = 0xf;  // bit cq: the 8 columns of block cq this lane fetches exist (EDGE only)
    if (EDGE) {
        wok = aok = 0;
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
            wok |= (n0 + 64 * cq + 8 * cu < g.N) ? (1u << cq) : 0u;
            aok |= (m0 + 64 * cq + 8 * cu < g.M) ? (1u << cq) : 0u;
        }
    }
    // the k-tile (block-relative index) that holds token rows >= Klim, or -1: K = Klim rounded up to 64, so it can only be the last
    // one.  EDGE = false reads those rows from memory (the caller guarantees they are readable) and zeroes them in REGISTERS
    // (TN_MASK below): no per-lane pointer select in any DMA of the loop.
    const int tpart = (!EDGE && (g.Klim % BK) != 0 && kb + nk * BK == g.K) ? nk - 1 : -1;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_tn_zero_page);
    constexpr int AOFF = (RING ? 3 : 2) * TN_TILE;  // LDS (elements): W tiles [2 | ring of 3] at 0, A tiles [2] behind them
    // T_: k-tile index whose rows are fetched; CQ_: column block
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem_raw);  // LDS byte address of smem
#define TN_DMA_W(BUF, T_, CQ_)                                                                                    \
    do {                                                                                                          \
        const uint32_t o_ = woff0 + (uint32_t)(T_) * wstep + (CQ_) * 128;                                         \
        if (EDGE) {                                                                                               \
            const bool ok_ = (trow0 + (T_) * BK < g.Klim) && ((wok >> (CQ_)) & 1u);                               \
            glds16(ok_ ? reinterpret_cast<const bf16_t*>(wbase + (size_t)o_) : zero, smem + (BUF) * TN_TILE + (wave * 4 + (CQ_)) * 512); \
        } else {                                                                                                  \
            glds16_sa(wbase, o_, lds0 + (uint32_t)(((BUF) * TN_TILE + (wave * 4 + (CQ_)) * 512) * 2));            \
        }                                                                                                         \
    } while (0)
#define TN_DMA_A(BUF, T_, CQ_)                                                                                    \
    do {                                                                                                          \
        const uint32_t o_ = aoff0 + (uint32_t)(T_) * astep + (CQ_) * 128;                                         \
        if (EDGE) {                                                                                               \
            const bool ok_ = (trow0 + (T_) * BK < g.Klim) && ((aok >> (CQ_)) & 1u);                               \
            glds16(ok_ ? reinterpret_cast<const bf16_t*>(abase + (size_t)o_) : zero, smem + AOFF + (BUF) * TN_TILE + (wave * 4 + (CQ_)) * 512); \
        } else {                                                                                                  \
            glds16_sa(abase, o_, lds0 + (uint32_t)((AOFF + (BUF) * TN_TILE + (wave * 4 + (CQ_)) * 512) * 2));     \
        }                                                                                                         \
    } while (0)

    // ---- fragment read addresses (elements).  lane -> 16-lane group fg (token rows 8 fg .. of a 32-token k-step), li = lane & 15:
    // row (li >> 2) of the 4-row block, 4-column segment li & 3 of the fragment's 16 columns.
    const int fr = lane & 15, fg = lane >> 4;
    const int swz_r = (((fr >> 3) & 1) << 1) | ((fg & 1) << 2);
    const int lbase = fg * 4 * 512 + (fr >> 2) * 64 + (fr & 1) * 4;
    int uo[4];  // unit offset of fragment b = 0..3 inside its 64-column block
#pragma unroll
    for (int b = 0; b < 4; ++b) uo[b] = ((2 * b + ((fr >> 1) & 1)) ^ swz_r) * 8;
    const bf16_t* ldsW = smem + wn * 512 + lbase;
    const bf16_t* ldsA = smem + AOFF + (2 * wm) * 512 + lbase;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[2][4], af[2][4];

    // k-step kk covers pieces tq = 4 kk + fg: + 16 kk pieces
#define TN_READ_W(BUF)                                                                                            \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) wf[kk][i] = tr_read8(ldsW + (BUF) * TN_TILE + kk * 16 * 512 + uo[i]);
#define TN_READ_A(BUF, HI)                                                                                        \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                              \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) af[kk][j] = tr_read8(ldsA + (BUF) * TN_TILE + kk * 16 * 512 + (HI) * 512 + uo[j]);
    // partial last k-tile: this lane's fragment of k-step kk holds token rows 32 kk + 8 fg + 0..7 of the tile; rows >= rem are zeroed
    // (both operands: the memory behind them is arbitrary).  Dword d of a fragment = rows 2 d, 2 d + 1.
    const int rem = g.Klim - (kb + (nk - 1) * BK);  // valid rows of the last tile of this block (meaningful when tpart >= 0)
    auto frag_mask = [&](int kk, int d) -> uint32_t {
        const int nv = rem - 32 * kk - 8 * fg - 2 * d;  // valid rows from the first row of this dword
        return nv >= 2 ? 0xffffffffu : (nv == 1 ? 0x0000ffffu : 0u);
    };
    auto mask_frag = [&](bf16x8& f, int kk) {
        u32x4 v = __builtin_bit_cast(u32x4, f);
        v[0] &= frag_mask(kk, 0); v[1] &= frag_mask(kk, 1); v[2] &= frag_mask(kk, 2); v[3] &= frag_mask(kk, 3);
        f = __builtin_bit_cast(bf16x8, v);
    };
#define TN_MASK_W()                                                                                               \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) mask_frag(wf[kk][i], kk);
#define TN_MASK_A()                                                                                               \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                              \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) mask_frag(af[kk][j], kk);
#define TN_MFMA(MB)                                                                                               \
    do {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                          \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                     \
                    acc[i][(MB) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][i], af[kk][j], acc[i][(MB) + j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)
#define TN_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
    // m-split k-tile (gemm2p's Q2_TILE): ph0 = all W fragments x the group's first 4 column fragments, ph1 = x its last 4.
    // DMA of tile T + 1: ph0 W (4 pieces) + the "lo" column blocks 0 and 2 of A (2), ph1 the "hi" blocks 1 and 3 (2).
#define TN_TILE_STEP(BUF, T)                                                                                      \
    do {                                                                                                          \
        const bool has1 = (T) + 1 < nk;                                                                           \
        const bool part = !EDGE && (T) == tpart;  /* block-uniform; only ever true in the last step */            \
        TN_READ_W(BUF)                                                                                            \
        TN_READ_A(BUF, 0)                                                                                         \
        if (has1) {                                                                                               \
            TN_DMA_W((BUF) ^ 1, (T) + 1, 0); TN_DMA_W((BUF) ^ 1, (T) + 1, 1);                                     \
            TN_DMA_W((BUF) ^ 1, (T) + 1, 2); TN_DMA_W((BUF) ^ 1, (T) + 1, 3);                                     \
            TN_DMA_A((BUF) ^ 1, (T) + 1, 0); TN_DMA_A((BUF) ^ 1, (T) + 1, 2);                                     \
            TN_WAIT(6);                                                                                           \
        } else {                                                                                                  \
            TN_WAIT(0);                                                                                           \
        }                                                                                                         \
        if (part) { TN_MASK_W() TN_MASK_A() }                                                                     \
        bar_raw_fn();                                                                                             \
        TN_MFMA(0);                                                                                               \
        bar_raw_fn();                                                                                             \
        TN_READ_A(BUF, 1)                                                                                         \
        if (has1) {                                                                                               \
            TN_DMA_A((BUF) ^ 1, (T) + 1, 1); TN_DMA_A((BUF) ^ 1, (T) + 1, 3);                                     \
            TN_WAIT(2);                                                                                           \
        } else {                                                                                                  \
            TN_WAIT(0);                                                                                           \
        }                                                                                                         \
        if (part) { TN_MASK_A() }                                                                                 \
        bar_raw_fn();                                                                                             \
        TN_MFMA(4);                                                                                               \
        bar_raw_fn();                                                                                             \
    } while (0)

    // ring k-tile (gemm3w's R_TILE): W slot WB = T % 3 (W2 = (T + 2) % 3: the slot tile T + 2 goes to), A buffer AB = T % 2
#define TN_RING_STEP(WB, W2, AB, T)                                                                                   \
    do {                                                                                                          \
        const bool hasA = (T) + 1 < nk, hasW = (T) + 2 < nk;                                                      \
        const bool part = (T) == tpart;                                                                           \
        TN_READ_W(WB)                                                                                             \
        TN_READ_A(AB, 0)                                                                                          \
        if (hasA) { TN_DMA_A((AB) ^ 1, (T) + 1, 0); TN_DMA_A((AB) ^ 1, (T) + 1, 2); }                             \
        if (hasW) { TN_DMA_W(W2, (T) + 2, 0); TN_DMA_W(W2, (T) + 2, 1); TN_WAIT(6); }     \
        else { TN_WAIT(0); }                                                                                      \
        if (part) { TN_MASK_W() TN_MASK_A() }                                                                     \
        bar_raw_fn();                                                                                             \
        TN_MFMA(0);                                                                                               \
        bar_raw_fn();                                                                                             \
        TN_READ_A(AB, 1)                                                                                          \
        if (hasA) { TN_DMA_A((AB) ^ 1, (T) + 1, 1); TN_DMA_A((AB) ^ 1, (T) + 1, 3); }                             \
        if (hasW) { TN_DMA_W(W2, (T) + 2, 2); TN_DMA_W(W2, (T) + 2, 3); TN_WAIT(6); }     \
        else { TN_WAIT(0); }                                                                                      \
        if (part) { TN_MASK_A() }                                                                                 \
        bar_raw_fn();                                                                                             \
        TN_MFMA(4);                                                                                               \
        bar_raw_fn();                                                                                             \
    } while (0)

    if constexpr (RING) {
        // prologue: all of tile 0 and the W tile of tile 1 (the latter may still be in flight: retired by ph1(0)'s wait)
        TN_DMA_W(0, 0, 0); TN_DMA_W(0, 0, 1); TN_DMA_W(0, 0, 2); TN_DMA_W(0, 0, 3);
        TN_DMA_A(0, 0, 0); TN_DMA_A(0, 0, 1); TN_DMA_A(0, 0, 2); TN_DMA_A(0, 0, 3);
        if (nk > 1) {
            TN_DMA_W(1, 1, 0); TN_DMA_W(1, 1, 1); TN_DMA_W(1, 1, 2); TN_DMA_W(1, 1, 3);
            TN_WAIT(4);
        } else {
            TN_WAIT(0);
        }
        bar_raw_fn();
        if (wm == 1) bar_raw_fn();  // group 1 runs one barrier behind group 0
        // ring slot / A buffer as RUN-TIME scalars (one copy of the step; the slot offsets live in SGPRs and are added to the four
        // per-lane fragment addresses at their use: literal slots unrolled six-fold made the compiler keep ~20 precomputed LDS
        // addresses alive and spill)
        int wslot = 0, abuf = 0;
        for (int t = 0; t < nk; ++t) {
            const int w2 = wslot == 0 ? 2 : wslot - 1;  // (wslot + 2) % 3
            TN_RING_STEP(wslot, w2, abuf, t);
            wslot = wslot == 2 ? 0 : wslot + 1;
            abuf ^= 1;
        }
        if (wm == 0) bar_raw_fn();  // re-align the barrier counts of the two groups
        epilogue8p<EPI, 8>(g, acc, n0, wn, m0 + gbase, fr, fg, nullptr);
        return;
    }
    // ---- prologue: all of tile 0
    TN_DMA_W(0, 0, 0); TN_DMA_W(0, 0, 1); TN_DMA_W(0, 0, 2); TN_DMA_W(0, 0, 3);
    TN_DMA_A(0, 0, 0); TN_DMA_A(0, 0, 1); TN_DMA_A(0, 0, 2); TN_DMA_A(0, 0, 3);
    TN_WAIT(0);
    bar_raw_fn();
    if (wm == 1) bar_raw_fn();  // group 1 runs one barrier behind group 0
    {
        int t = 0;
        for (; t + 1 < nk; t += 2) {
            TN_TILE_STEP(0, t);
            TN_TILE_STEP(1, t + 1);
        }
        if (t < nk) TN_TILE_STEP(0, t);
    }
    if (wm == 0) bar_raw_fn();  // re-align the barrier counts of the two groups
    if constexpr (!RING) {
        __shared__ int s_last;
        if (g.splits > 1 && !splitk_exchange<8, 32>(g, acc, tm * tilesN + tn, split, &s_last)) return;
    }
    epilogue8p<EPI, 8>(g, acc, n0, wn, m0 + gbase, fr, fg, nullptr);
#undef TN_RING_STEP
#undef TN_TILE_STEP
#undef TN_WAIT
#undef TN_MFMA
#undef TN_MASK_A
#undef TN_MASK_W
#undef TN_READ_A
#undef TN_READ_W
#undef TN_DMA_A
#undef TN_DMA_W
}

constexpr int SMEM_TN_RING = 5 * TN_TILE * 2;  // 160 KiB

template <int EPI, bool EDGE, bool RING>
int launch_tn_k(const GemmArgs& g, hipStream_t s) {
    static bool attr_set = false;
    auto kfn = gemm_tn_kernel<EPI, EDGE, RING>;
    constexpr int smem = RING ? SMEM_TN_RING : SMEM3_BYTES;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return set_error_hip(e, "hipFuncSetAttribute(gemm_tn)", __FILE__, __LINE__);
        attr_set = true;
    }
    const int tiles = ((g.M + 255) / 256) * ((g.N + B2 - 1) / B2);
    gemm_count_launch(g.splits > 1);
    kfn<<<dim3(tiles * g.splits), dim3(512), smem, s>>>(g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error_hip(e, "gemm_tn launch", __FILE__, __LINE__);
    return 0;
}
template <int EPI>
int launch_tn(GemmArgs g, hipStream_t s) {
    const int tiles = ((g.M + 255) / 256) * ((g.N + B2 - 1) / B2);
    // split-K: the same rule as the production kernel (a function of (M, N, K) alone -> run-to-run identical summation order)
    g.splits = 1;
    const int S = gemm_splitk_count(g.M, g.N, g.K, showo_cu_usable((void*)s));
    if (S >= 2) {
        if (tiles * S > gemm_splitk_ticks() || !gemm_splitk_ws(s, (size_t)tiles * S * 32 * 512 * sizeof(float4), &g.ws, &g.tick))
            return set_error_msg(7, "gemm_tn: split-K workspace unavailable (first use of a split shape inside a stream capture): run the "
                                    "shape once eagerly, or set SHOWO_GEMM_SPLITK=0");
        g.splits = S;
    }
    // unchecked fetches need every 256-column tile inside a row of its operand, and token rows readable up to K (g.flags bit 0:
    // the caller vouches for the rows between Klim and K)
    const bool edge = ((g.M + 255) / 256) * 256 > g.lda || ((g.N + B2 - 1) / B2) * B2 > g.ldw || ((g.Klim % GEMM_BK) != 0 && !(g.flags & 1));
    static int ring_env = -1;  // SHOWO_GEMM_TN_RING=0: double-buffered form everywhere (A/B)
    if (ring_env < 0) { const char* e = getenv("SHOWO_GEMM_TN_RING"); ring_env = e ? atoi(e) : 1; }
    if (edge) return launch_tn_k<EPI, true, false>(g, s);
    if (g.splits == 1 && ring_env) return launch_tn_k<EPI, false, true>(g, s);
    return launch_tn_k<EPI, false, false>(g, s);
}

}  // namespace
}  // namespace showo

using namespace showo;

// out fp32 [M, N] (ldo) = A^T B (+ out when accumulate): A bf16 [T, M] (lda), B bf16 [T, N] (ldb), token rows t < T.
// rows_padded != 0: both buffers are READABLE up to row roundup(T, 64) - 1 (contents arbitrary: those rows are zeroed in registers);
// the fast kernels then issue unchecked DMAs.  rows_padded == 0 with T % 64 != 0 takes the checked (slower) form.
extern "C" int showo_gemm_tn_bf16(const uint16_t* A, int lda, const uint16_t* B, int ldb, float* out, int ldo, int M, int N, int T,
                                  int accumulate, int rows_padded, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (T <= 0) return set_error_msg(1, "gemm_tn: T must be positive");
    if (!A || !B || !out) return set_error_msg(1, "gemm_tn: null operand");
    // columns are fetched 8 at a time: a row must hold the 8-column unit that contains its last column (the extra columns may hold
    // anything finite or not -- their products land in output rows / columns >= M / N, which are never stored)
    if ((lda % 8) || (ldb % 8) || lda < ((M + 7) & ~7) || ldb < ((N + 7) & ~7) || (((uintptr_t)A) & 15) || (((uintptr_t)B) & 15))
        return set_error_msg(1, "gemm_tn: lda, ldb must be multiples of 8 covering M, N rounded up to 8, operands 16-byte aligned");
    if ((int64_t)(T + 64) * lda * 2 >= ((int64_t)1 << 32) || (int64_t)(T + 64) * ldb * 2 >= ((int64_t)1 << 32))
        return set_error_msg(1, "gemm_tn: operand larger than 4 GiB (32-bit byte offsets)");
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = B; g.ldw = ldb; g.Wlo = nullptr; g.bias = nullptr; g.bias_per_row = 0;
    g.out = out; g.ldo = ldo; g.resid = accumulate ? out : nullptr; g.ldr = ldo;
    g.M = M; g.N = N; g.K = ((T + GEMM_BK - 1) / GEMM_BK) * GEMM_BK; g.Klim = T;
    // tile-group width (n-panels per XCD group): 8 measured best on the weight-gradient shapes (back-to-back launches, same box:
    // gn 4 / 8 / 32 = 1 157 / 1 195 / 1 134 TF/s on dW2, 1 136 / 1 167 / 1 186 on dW1, 986 / 1 007 / 1 039 on dWqkv); SHOWO_GEMM_TN_GN overrides
    static int tn_gn = 0;
    if (!tn_gn) { const char* e = getenv("SHOWO_GEMM_TN_GN"); tn_gn = (e && atoi(e) > 0) ? atoi(e) : 8; }
    g.gn = tn_gn; g.flags = rows_padded ? 1 : 0; g.dbg = nullptr;
    g.vec_out = ((ldo % 4) == 0) && ((((uintptr_t)out) & 15) == 0);
    ProfScope prof(PROF_GEMM, 2.0 * M * N * T, (hipStream_t)stream);
    if (accumulate) return launch_tn<SHOWO_EPI_RESID_F32>(g, (hipStream_t)stream);
    return launch_tn<SHOWO_EPI_F32>(g, (hipStream_t)stream);
}
