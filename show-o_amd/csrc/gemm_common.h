// Shared pieces of the bf16 MFMA GEMM kernels (gemm.hip, gemm2p.hip): launch arguments, epilogues, raw barrier.
#pragma once
#include "common.h"
#include "../../include/showo_hip.h"

namespace showo {

struct GemmArgs {
    const bf16_t* A; int lda;
    const bf16_t* W; int ldw;
    const bf16_t* Wlo;  // split-precision mode: low halves of the weights (same layout as W)
    const float* bias; int bias_per_row;
    void* out; int ldo;
    const float* resid; int ldr;
    int M, N, K;
    int vec_out;  // 1: out/resid rows allow 4-wide vector access
    int gn;       // v3: n-panels per tile group (L2 locality of the block -> tile map)
    int flags;    // v3 experiments: bit0 = no group stagger
    unsigned long long* dbg;  // v3 debug build: per-barrier timestamps of block 0
    // fused q/k LayerNorm + RoPE + head-major relayout epilogue (EPI_QKV): out is unused
    const float *qw, *qb, *kw, *kb, *cosT, *sinT;
    bf16_t *Q, *Kd, *Vt;
    int L, nH, pos0, Lcap, Lp;
    float eps;
    // gemm2p only.  K-concatenated activation operand: columns k < Ksplit come from A (lda), k >= Ksplit from A2 (lda2) at
    // k - Ksplit; the weight rows are [W_a | W_b] (K-contiguous, ldw).  Phi's block adds dense(attn) and fc2(ffn) into the same
    // residual row (models/phi.py:774-790), so both projections are ONE GEMM over K = H + F with one residual read-modify-write.
    const bf16_t* A2 = nullptr; int lda2 = 0; int Ksplit = 1 << 30;
    // EPI_QKV only: output columns n >= Nq are the fc1 rows of the [Wqkv ; W1] weight: bias + gelu_new -> out2[m][n - Nq] (bf16).
    // q/k/v and fc1 read the same LayerNorm output (models/phi.py:776-790).
    int Nq = 1 << 30; bf16_t* out2 = nullptr; int ldo2 = 0;
    // EPI_QKV, training forward (showo_gemm_qkv_fc1_save_bf16): what backward needs is saved by the same launch.  raw != nullptr:
    // raw[m][n] = bf16(A Wqkv^T + b) for n < 3 nH 64 (ldraw), and the LayerNorm / RoPE work on THESE rounded values -- the numbers
    // showo_qkln_rope_bwd recomputes its statistics from.  pre != nullptr: pre[m][n - Nq] = bf16(A W1^T + b1) (ldo2) and
    // out2 = gelu_new of that ROUNDED value -- the bits of the separate fc1 GEMM + showo_gelu_bf16 launches.
    bf16_t* raw = nullptr; int ldraw = 0; bf16_t* pre = nullptr;
    // gemm2p only: W is in the tiled layout of showo_gemm_tile_weight ([ceil(N/256)][K/64][256][64] bf16, 16-B chunks pre-swizzled)
    int wtiled = 0;
    // gemm2p only: split-K for launches with few tiles (M = 631 prefill, the CLIP tower).  The grid is tiles x splits; split s of a tile
    // accumulates k-tiles [s * per, (s + 1) * per), every split writes its fp32 fragments to ws, and the LAST block to arrive at the
    // tile's ticket sums the `splits` partials IN SPLIT ORDER (its own included, read back from ws) and runs the ordinary epilogue:
    // the result does not depend on which block arrived last.
    int splits = 1; float4* ws = nullptr; unsigned* tick = nullptr;
    // coop != 0 (host: tiles x splits <= the CUs this stream may use, one block per CU, so every split of a tile is resident at the same
    // time): the `splits` blocks of a tile reduce TOGETHER -- each sums and stores the fragments it owns (splitk_coop_finish) -- instead
    // of the last arriver reading every partial alone at the cross-XCD single-block rate.  Same split order per element: same bits.
    int coop = 0;
    int coop_polls = 1 << 15;  // cooperative form: polls (~64 ns each) before a block gives up waiting and the tile falls back to the last-arriver sum (showo_gemm_set_coop_polls; 0 = give up at once: tests)
    // gemm_tn only (token-major operands, contraction over rows): rows k >= Klim of both operands read as zero (K = Klim rounded up to 64)
    int Klim = 0;
    // EPI_QKV_SPLIT only (accuracy mode on the production kernel): every bf16 output also gets its LOW half -- value - bf16(value),
    // rounded to bf16 -- at the same index of a second image, so that hi + lo carries the fp32 accumulator to 2^-17
    bf16_t *Qlo = nullptr, *Klo = nullptr, *Vtlo = nullptr, *out2lo = nullptr;
    // 16-bit operand type of A, W and of every 16-bit output (common.h Op16): 0 = bfloat16, 1 = IEEE half (SHOWO_OP_F16).  Host-side
    // selector of the kernel instance; the device code carries it as the template parameter F16.
    int op = 0;
};

constexpr int EPI_QKV = 4;  // internal epilogue code of showo_gemm_qkv_bf16
constexpr int EPI_QKV_SPLIT = 5;  // the same projection with (hi, lo) bf16 pairs as outputs (showo_gemm_qkv_fc1_split)
constexpr int B2 = 256;   // tile width (n) of the 256-wide kernels
constexpr int GEMM_BK = 64;
constexpr int P3_BUF_ELEMS = 2 * 256 * 64;  // one k-tile: W[256][64] then A[256][64] (64 KiB)
constexpr int SMEM3_BYTES = 2 * P3_BUF_ELEMS * 2;

static __device__ inline void load_bias4(const GemmArgs& g, int n, float (&bn)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bn[r] = 0.f;
    if (g.bias && !g.bias_per_row) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bn[r] = (n + r < g.N) ? g.bias[n + r] : 0.f;
    }
}

// one MFMA C fragment: this lane holds out[m][n .. n+3]
template <int EPI, bool F16 = false>
static __device__ inline void store_frag(const GemmArgs& g, const f32x4& acc, int m, int n, const float (&bn)[4]) {
    if (m >= g.M || n >= g.N) return;
    float v[4];
    const float bm = (g.bias && g.bias_per_row) ? g.bias[m] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[r] + bn[r] + bm;
    const bool full = (n + 3 < g.N) && g.vec_out;
    if (EPI == SHOWO_EPI_BF16 || EPI == SHOWO_EPI_GELU_BF16) {
        if (EPI == SHOWO_EPI_GELU_BF16) {
            gelu4(v);
        }
        bf16_t* o = reinterpret_cast<bf16_t*>(g.out) + (int64_t)m * g.ldo + n;
        if (full) {
            uint2 pk;
            pk.x = Op16<F16>::pack2(v[0], v[1]);
            pk.y = Op16<F16>::pack2(v[2], v[3]);
            *reinterpret_cast<uint2*>(o) = pk;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < g.N) o[r] = Op16<F16>::cvt(v[r]);
        }
    } else {
        float* o = reinterpret_cast<float*>(g.out) + (int64_t)m * g.ldo + n;
        if (EPI == SHOWO_EPI_RESID_F32) {
            const float* rs = g.resid + (int64_t)m * g.ldr + n;
            if (full) {
                float4 rv = *reinterpret_cast<const float4*>(rs);
                v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < g.N) v[r] += rs[r];
            }
        }
        if (full) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < g.N) o[r] = v[r];
        }
    }
}

// x + (the values of the three lanes 16, 32 and 48 away): the reduction of a token's 64 head values in the QKV epilogue, where they
// sit in the 4 lanes fr + 16 {0..3}.  gfx950's row-swap instructions (v_permlane16_swap / v_permlane32_swap: 3 VALU instructions per
// step, no LDS round trip) in the SAME order as `x += shfl_xor(x, 16); x += shfl_xor(x, 32)` -- (own + 16-partner) + (the 32-partner's
// pair sum), floating-point addition being commutative -- so the result has the bits of the ds_bpermute form it replaces.
static __device__ __forceinline__ float sum4_lanes16(float x) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

static __device__ __forceinline__ void bar_raw_fn() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// ---- coalesced epilogue stores through LDS (bf16 outputs).  An MFMA C fragment leaves a lane with 4 consecutive columns of ONE row:
// stored directly, a wave instruction writes 16 rows x 32 B (8-byte stores, every 128-B line in four separate instructions) and the
// store tail of a tile is issue-bound (~7 us for the 24-32 dwordx2 per lane of a 192-256-row tile: MI355X_MICROARCH.md, "attention
// epilogue store tail"; measured here as ~12 us of fixed cost per 34-us K = 2048 tile, profiles/r2_gemm_harness.txt).  After the
// k-loop the operand buffers in LDS are dead, so each wave stages its own 64-column x (16 MF)-row tile in a private 16 KiB slice as
// [row][64] bf16 (16-B chunk c of row r at position c ^ (r & 7): two-way write conflicts at most), reads it back row-major and stores
// 16 B per lane: one instruction = 8 rows x 128 B = FULL lines.  No block barrier: only this wave touches its slice.
static __device__ __forceinline__ void stage_frag_bf16(bf16_t* sc, int row, int i, int fg, uint2 pk) {
    const int c = i * 2 + (fg >> 1);
    *reinterpret_cast<uint2*>(sc + row * 64 + ((c ^ (row & 7)) << 3) + ((fg & 1) << 2)) = pk;
}
static __device__ __forceinline__ uint4 unstage_row16(const bf16_t* sc, int row, int chunk) {
    return *reinterpret_cast<const uint4*>(sc + row * 64 + ((chunk ^ (row & 7)) << 3));
}

// split-K exchange (GemmArgs::splits): returns true in the block that has to run the epilogue, with acc = the sum over all splits.
// NFS = fragments per thread reserved in ws (4 x the larger group's fragment count); all 512 threads of the block call it.
template <int MF, int NFS>
static __device__ __forceinline__ bool splitk_exchange(const GemmArgs& g, f32x4 (&acc)[4][8], int tile, int split, int* s_last) {
    const int tid = threadIdx.x;
    float4* base = g.ws + (size_t)tile * g.splits * NFS * 512;
    float4* mine = base + (size_t)split * NFS * 512 + tid;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) mine[(i * MF + j) * 512] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    // release by ONE lane after the block barrier: every wave's stores have reached this XCD's L2 (s_waitcnt vmcnt(0) precedes the
    // barrier), and a single L2 write-back makes the whole partial visible device-wide before the ticket moves.  (A fence per
    // thread -- 8 write-backs per block, 2 000 per launch -- made the exchange cost 20-70 us per launch.)
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        const unsigned old = atomicAdd(g.tick + tile, 1u);
        const int last = old == (unsigned)(g.splits - 1);
        if (last) atomicExch(g.tick + tile, 0u);  // ready for the next launch on this stream
        *s_last = last;
        if (last) __threadfence();  // acquire: this CU's vector cache and the XCD's L2 drop stale lines of the workspace
    }
    __syncthreads();
    if (!*s_last) return false;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < g.splits; ++s) {
        const float4* p = base + (size_t)s * NFS * 512 + tid;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                const float4 v = p[(i * MF + j) * 512];
                acc[i][j][0] += v.x; acc[i][j][1] += v.y; acc[i][j][2] += v.z; acc[i][j][3] += v.w;
            }
    }
    return true;
}

// Cooperative form of the split-K exchange (GemmArgs::coop; plain epilogues only).  All `splits` blocks of a tile are resident at once,
// so after publishing its partial (as above) a block WAITS for the tile's other partials and then reduces the fragments it owns --
// fragment f = i * MF + j of every thread belongs to split f % splits -- over all partials IN SPLIT ORDER (the order the last-arriver
// form uses: identical bits) and stores them with store_frag.  Per block: one tile's worth of partial reads (splits x tile / splits)
// instead of `splits` tiles in ONE block at 62-70 GB/s (MI355X_MICROARCH.md "handoff-payload"): at M = 516, K = 10 240 (cfg1's
// dense|fc2 launch, 24 tiles x 10 splits) the serial read of 1.8 MB took about half of the launch.
// Hand-off protocol (cdna_hip_programming.md Guideline 16): plain stores -> vmcnt(0) -> barrier -> one lane: agent release fence ->
// arrival ticket; then ONE relaxed poll loop -> ONE agent acquire fence -> barrier -> plain loads.  tick[tile] counts arrivals,
// tick[2048 + tile] departures, tick[4096 + tile] holds the tile's mode; the last block to leave zeroes all three for the next launch.
template <int EPI, int MF, int NFS, bool F16 = false>
static __device__ __forceinline__ void splitk_coop_finish(const GemmArgs& g, f32x4 (&acc)[4][8], int tile, int split, int n0, int wn, int mrow0,
                                                          int fr, int fg, int* s_ml) {  // s_ml: two ints of LDS declared ONCE by the kernel
    const int tid = threadIdx.x;
    float4* base = g.ws + (size_t)tile * g.splits * NFS * 512;
    float4* mine = base + (size_t)split * NFS * 512 + tid;
    if (g.coop == 2) {
        // write-through (sc1) stores of the partial: the bytes leave the XCD's L2 as they are stored, so no release fence (an L2
        // write-back of up to 180 KB of fresh lines per block) is needed in front of the ticket (Guideline 16 R1 / "publish-large")
        // (wave-uniform descriptor: the slab of this split; the lane's 16 bytes are selected by the voffset)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(base + (size_t)split * NFS * 512), 0, -1, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                u32x4 v;
                v[0] = __float_as_uint(acc[i][j][0]); v[1] = __float_as_uint(acc[i][j][1]);
                v[2] = __float_as_uint(acc[i][j][2]); v[3] = __float_as_uint(acc[i][j][3]);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, tid * 16, (i * MF + j) * 512 * 16, 16 /* sc1 */);
            }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MF; ++j) mine[(i * MF + j) * 512] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // tick[tile] = arrivals, tick[2048 + tile] = departures, tick[4096 + tile] = the tile's reduction mode: 0 undecided, 1 cooperative
    // (every sibling was seen), 2 last-arriver (some block gave up waiting).  A block that does not see all its siblings within
    // g.coop_polls polls (~2 ms: they are not resident -- another process on the GPU, a CU mask the host did not know about) no longer
    // traps (round 5: __builtin_trap after ~100 s): it switches the TILE to the last-arriver form and leaves; the block whose arrival
    // completes the count then sums every partial alone.  Same split order per element in both forms: the same bits.
    // (The two block-wide flags come from the KERNEL: a function-scope __shared__ here is one variable per template instance, and the
    //  two wave groups of a tile with MF0 != MF1 call different instances -- group 1 then read flags nobody wrote.  Round 6, found by
    //  the full-size [2,387] fixture once the tuner picked a 5 + 4 tile for a split launch.)
    int& s_mode = s_ml[0];
    int& s_last = s_ml[1];
    if (tid == 0) {
        if (g.coop != 2) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned old = __hip_atomic_fetch_add(g.tick + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        bool all = false;
        while (!(all = __hip_atomic_load(g.tick + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)g.splits) && ++spins <= (unsigned)g.coop_polls)
            __builtin_amdgcn_s_sleep(4);
        unsigned expect = 0u;
        const unsigned want = all ? 1u : 2u;
        // the first block to decide fixes the tile's mode; everybody else adopts it (mode 1 implies that all siblings have arrived)
        const bool won = __hip_atomic_compare_exchange_strong(g.tick + 4096 + tile, &expect, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_mode = won ? (int)want : (int)expect;
        s_last = old == (unsigned)(g.splits - 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    const bool coop_mode = s_mode == 1;
    if (coop_mode || s_last) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + fg * 4;
            float bn[4];
            load_bias4(g, n, bn);
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                if (coop_mode && (i * MF + j) % g.splits != split) continue;  // block-uniform: cooperative = the fragments this split owns
                f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
                for (int sp = 0; sp < g.splits; ++sp) {
                    const float4 v = base[(size_t)sp * NFS * 512 + (i * MF + j) * 512 + tid];
                    sum[0] += v.x; sum[1] += v.y; sum[2] += v.z; sum[3] += v.w;
                }
                store_frag<EPI, F16>(g, sum, mrow0 + j * 16 + fr, n, bn);
            }
        }
    }
    __syncthreads();  // every partial read of this block has been issued and consumed
    if (tid == 0) {
        const unsigned left = __hip_atomic_fetch_add(g.tick + 2048 + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == (unsigned)(g.splits - 1)) {  // the last block to leave (in either mode) readies the three words for the next launch
            __hip_atomic_store(g.tick + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g.tick + 2048 + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g.tick + 4096 + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// epilogue of the 256-wide phase-split kernels: the wave holds 4 (n) x MF (m) 16x16 fragments; mrow0 = first row of
// the wave's m range, n0 + wn*64 = its first column.
template <int EPI, int MF, bool F16 = false>
static __device__ __forceinline__ void epilogue8p(const GemmArgs& g, f32x4 (&acc)[4][8], int n0, int wn, int mrow0, int fr, int fg,
                                                   bf16_t* stg = nullptr) {  // stg: this wave's 16 KiB LDS slice (or nullptr: direct stores)
    const int lane_ = fg * 16 + fr;
    constexpr int VRS = MF < 8 ? 16 * MF + 2 : 128;  // row stride (elements) of the transposed V staging image: +2 breaks the 4-way bank conflict
    const int rrow = lane_ >> 3, rchunk = lane_ & 7;  // read-back role: row inside an 8-row group, 16-B chunk of the 128-B row
    // Q / K rows keep their direct stores unless flags bit 5 is set: staging them measured -3...-5 % on the fused projection (the
    // per-token LayerNorm / RoPE math, not the stores, dominates those tiles; profiles/r2_gemm_harness.txt, r2n)
    bf16_t* stg_qk = (g.flags & 32) ? stg : nullptr;
    bf16_t* stg_v = (g.flags & 64) ? nullptr : stg;  // V^T transpose staging (bit 6 turns it off for A/B runs)
    if constexpr (EPI == EPI_QKV) {
        // The wave's 64 output columns are exactly one head of q, k or v (column tiles and wave tiles are head
        // aligned); a token's 64 values sit in the 4 lanes fr + 16*{0..3} (16 each: dims 16i + 4fg + r).
        // Replaces the bf16 round trip qkv -> showo_qk_prep: LayerNorm(64) and the rotation see the fp32 accumulators.
        const int nbase = n0 + wn * 64;
        if (n0 >= g.Nq) {  // fc1 tail of the fused [Wqkv ; W1] projection (block-uniform: Nq is a multiple of the tile width)
            const bool staged = stg != nullptr && (g.ldo2 % 8) == 0 && ((g.N - g.Nq) % 8) == 0 && ((((uintptr_t)g.out2) & 15) == 0) &&
                                (!g.pre || ((((uintptr_t)g.pre) & 15) == 0));
            if (g.pre) {  // training forward: save the pre-activation (bf16) and make the accumulators hold its rounded value
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = nbase + i * 16 + fg * 4;
                    float bn[4];
                    load_bias4(g, n, bn);
#pragma unroll
                    for (int j = 0; j < MF; ++j) {
                        const int m = mrow0 + j * 16 + fr;
                        uint2 pk;
                        pk.x = Op16<F16>::pack2(acc[i][j][0] + bn[0], acc[i][j][1] + bn[1]);
                        pk.y = Op16<F16>::pack2(acc[i][j][2] + bn[2], acc[i][j][3] + bn[3]);
                        // gelu below sees widen(pre) - bias so that (acc + bias) reproduces the rounded value exactly
                        acc[i][j][0] = Op16<F16>::lo_of(pk.x); acc[i][j][1] = Op16<F16>::hi_of(pk.x);
                        acc[i][j][2] = Op16<F16>::lo_of(pk.y); acc[i][j][3] = Op16<F16>::hi_of(pk.y);
                        if (staged) stage_frag_bf16(stg, j * 16 + fr, i, fg, pk);
                        else if (m < g.M && n < g.N) *reinterpret_cast<uint2*>(g.pre + (int64_t)m * g.ldo2 + (n - g.Nq)) = pk;
                    }
                }
                if (staged) {
#pragma unroll
                    for (int t = 0; t < 2 * MF; ++t) {
                        const int row = t * 8 + rrow, m = mrow0 + row, n = nbase + rchunk * 8;
                        const uint4 v = unstage_row16(stg, row, rchunk);
                        if (m < g.M && n < g.N) *reinterpret_cast<uint4*>(g.pre + (int64_t)m * g.ldo2 + (n - g.Nq)) = v;
                    }
                }
            }
            const bool rounded = g.pre != nullptr;  // acc already holds bf16(acc + bias): no second bias add
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = nbase + i * 16 + fg * 4;
                float bn[4];
                load_bias4(g, n, bn);
                if (rounded) { bn[0] = bn[1] = bn[2] = bn[3] = 0.f; }
#pragma unroll
                for (int j = 0; j < MF; ++j) {
                    const int m = mrow0 + j * 16 + fr;
                    uint2 pk;
                    // (training: acc holds the rounded pre-activation and bn is 0 -> the bits of showo_gelu_bf16 on the saved tensor)
                    float gv[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    gelu4_bias(gv, bn);
                    pk.x = Op16<F16>::pack2(gv[0], gv[1]);
                    pk.y = Op16<F16>::pack2(gv[2], gv[3]);
                    if (staged) stage_frag_bf16(stg, j * 16 + fr, i, fg, pk);
                    else if (m < g.M && n < g.N) *reinterpret_cast<uint2*>(g.out2 + (int64_t)m * g.ldo2 + (n - g.Nq)) = pk;
                }
            }
            if (staged) {
#pragma unroll
                for (int t = 0; t < 2 * MF; ++t) {
                    const int row = t * 8 + rrow, m = mrow0 + row, n = nbase + rchunk * 8;
                    const uint4 v = unstage_row16(stg, row, rchunk);
                    if (m < g.M && n < g.N) *reinterpret_cast<uint4*>(g.out2 + (int64_t)m * g.ldo2 + (n - g.Nq)) = v;
                }
            }
        } else if (nbase < g.N && g.Q == nullptr) {
            // raw-only save form (training forward, SHOWO_TRAIN_QKPREP): the q / k / v columns are stored as bf16(acc + bias) rows of
            // raw[m][n] and nothing else -- LayerNorm / RoPE / relayout run as showo_qk_prep on that tensor, which the backward keeps anyway
            const bool staged = stg != nullptr && (g.ldraw % 8) == 0 && ((((uintptr_t)g.raw) & 15) == 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = nbase + i * 16 + fg * 4;
                float bn[4];
                load_bias4(g, n, bn);
#pragma unroll
                for (int j = 0; j < MF; ++j) {
                    const int m = mrow0 + j * 16 + fr;
                    uint2 pk;
                    pk.x = Op16<F16>::pack2(acc[i][j][0] + bn[0], acc[i][j][1] + bn[1]);
                    pk.y = Op16<F16>::pack2(acc[i][j][2] + bn[2], acc[i][j][3] + bn[3]);
                    if (staged) stage_frag_bf16(stg, j * 16 + fr, i, fg, pk);
                    else if (m < g.M) *reinterpret_cast<uint2*>(g.raw + (int64_t)m * g.ldraw + n) = pk;
                }
            }
            if (staged) {
#pragma unroll
                for (int t = 0; t < 2 * MF; ++t) {
                    const int row = t * 8 + rrow, m = mrow0 + row;
                    const uint4 v = unstage_row16(stg, row, rchunk);
                    if (m < g.M) *reinterpret_cast<uint4*>(g.raw + (int64_t)m * g.ldraw + nbase + rchunk * 8) = v;
                }
            }
        } else if (nbase < g.N) {
            const int Hq = g.nH * 64;
            const int which = nbase / Hq;  // 0 = q, 1 = k, 2 = v (wave-uniform)
            const int head = (nbase - which * Hq) >> 6;
            float bn[4][4], lw[4][4], lb[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                load_bias4(g, nbase + i * 16 + fg * 4, bn[i]);
                if (which < 2) {
                    const float4 w4 = *reinterpret_cast<const float4*>((which ? g.kw : g.qw) + i * 16 + fg * 4);
                    const float4 b4 = *reinterpret_cast<const float4*>((which ? g.kb : g.qb) + i * 16 + fg * 4);
                    lw[i][0] = w4.x; lw[i][1] = w4.y; lw[i][2] = w4.z; lw[i][3] = w4.w;
                    lb[i][0] = b4.x; lb[i][1] = b4.y; lb[i][2] = b4.z; lb[i][3] = b4.w;
                }
            }
#pragma unroll
            for (int j = 0; j < MF; ++j) {
                const int m = mrow0 + j * 16 + fr;
                const bool valid = m < g.M;
                const int mm = valid ? m : g.M - 1;
                const int b = mm / g.L, l = mm - b * g.L, pos = g.pos0 + l;
                const int64_t bh = (int64_t)b * g.nH + head;
                float x[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[i][r] = acc[i][j][r] + bn[i][r];
                if (g.raw) {  // training forward: save bf16(qkv) for backward and continue from the rounded values
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        uint2 pk;
                        pk.x = Op16<F16>::pack2(x[i][0], x[i][1]);
                        pk.y = Op16<F16>::pack2(x[i][2], x[i][3]);
                        x[i][0] = Op16<F16>::lo_of(pk.x); x[i][1] = Op16<F16>::hi_of(pk.x);
                        x[i][2] = Op16<F16>::lo_of(pk.y); x[i][3] = Op16<F16>::hi_of(pk.y);
                        if (valid) *reinterpret_cast<uint2*>(g.raw + (int64_t)m * g.ldraw + nbase + i * 16 + fg * 4) = pk;
                    }
                }
                if (which == 2) {  // V^T[bh][d][pos]
                    if (stg_v) {  // staged transposed: [d][token] in the wave's LDS slice, stored below with the lanes along `pos`
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) stg_v[(i * 16 + fg * 4 + r) * VRS + j * 16 + fr] = Op16<F16>::cvt(x[i][r]);
                    } else if (valid) {
                        bf16_t* vp = g.Vt + bh * 64 * g.Lp + pos;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) vp[(int64_t)(i * 16 + fg * 4 + r) * g.Lp] = Op16<F16>::cvt(x[i][r]);
                    }
                    continue;
                }
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) sum += (x[i][0] + x[i][1]) + (x[i][2] + x[i][3]);
                sum = sum4_lanes16(sum);
                const float mean = sum * (1.0f / 64.0f);
                float sq = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { x[i][r] -= mean; sq += x[i][r] * x[i][r]; }
                sq = sum4_lanes16(sq);
                const float rstd = 1.0f / sqrtf(sq * (1.0f / 64.0f) + g.eps);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[i][r] = x[i][r] * rstd * lw[i][r] + lb[i][r];
                // partial rotary over dims [0, 32): rotate_half pairs d with d + 16 = fragments i = 0 and 1 of this lane
                const float4 c0 = *reinterpret_cast<const float4*>(g.cosT + (int64_t)pos * 32 + fg * 4);
                const float4 s0 = *reinterpret_cast<const float4*>(g.sinT + (int64_t)pos * 32 + fg * 4);
                const float4 c1 = *reinterpret_cast<const float4*>(g.cosT + (int64_t)pos * 32 + 16 + fg * 4);
                const float4 s1 = *reinterpret_cast<const float4*>(g.sinT + (int64_t)pos * 32 + 16 + fg * 4);
                const float cc0[4] = {c0.x, c0.y, c0.z, c0.w}, ss0[4] = {s0.x, s0.y, s0.z, s0.w};
                const float cc1[4] = {c1.x, c1.y, c1.z, c1.w}, ss1[4] = {s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y0 = x[0][r], y1 = x[1][r];
                    x[0][r] = y0 * cc0[r] - y1 * ss0[r];
                    x[1][r] = y1 * cc1[r] + y0 * ss1[r];
                }
                const float sc = which == 0 ? 0.125f : 1.0f;  // 1/sqrt(64) folded into Q (exact in bf16)
                if (stg_qk) {  // staged: the token's 64 values go to row j * 16 + fr of the wave's slice, stored as full 128-B rows below
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        uint2 pk;
                        pk.x = Op16<F16>::pack2(x[i][0] * sc, x[i][1] * sc);
                        pk.y = Op16<F16>::pack2(x[i][2] * sc, x[i][3] * sc);
                        stage_frag_bf16(stg_qk, j * 16 + fr, i, fg, pk);
                    }
                    continue;
                }
                if (!valid) continue;
                bf16_t* dst = which == 0 ? g.Q + (bh * g.L + l) * 64 : g.Kd + (bh * g.Lcap + pos) * 64;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint2 pk;
                    pk.x = Op16<F16>::pack2(x[i][0] * sc, x[i][1] * sc);
                    pk.y = Op16<F16>::pack2(x[i][2] * sc, x[i][3] * sc);
                    *reinterpret_cast<uint2*>(dst + i * 16 + fg * 4) = pk;
                }
            }
            if (stg_v && which == 2) {
                // V^T rows are contiguous along `pos`: with the lanes along the tile's tokens one store instruction writes up to 64
                // consecutive positions of ONE head dimension (1-2 lines) instead of one position of 64 dimensions (64 lines).
                // `pos` starts at an odd offset in the t2i loop (129 text rows), so the stores stay 2 bytes wide.
                bf16_t* vb[2];
                bool vok[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int mt = h * 64 + lane_, m = mrow0 + mt;
                    vok[h] = mt < 16 * MF && m < g.M;
                    const int mm = vok[h] ? m : 0;
                    const int b = mm / g.L, l = mm - b * g.L;
                    vb[h] = g.Vt + ((int64_t)b * g.nH + head) * 64 * g.Lp + g.pos0 + l;
                }
#pragma unroll 8
                for (int dd = 0; dd < 64; ++dd) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if (h * 64 < 16 * MF && vok[h]) vb[h][(int64_t)dd * g.Lp] = stg_v[dd * VRS + h * 64 + lane_];
                }
            }
            if (stg_qk && which < 2) {  // one (token, head) row = 128 B contiguous in Q / K: a wave instruction stores 8 of them
#pragma unroll
                for (int t = 0; t < 2 * MF; ++t) {
                    const int row = t * 8 + rrow, m = mrow0 + row;
                    const uint4 v = unstage_row16(stg_qk, row, rchunk);
                    if (m < g.M) {
                        const int b = m / g.L, l = m - b * g.L;
                        const int64_t bh = (int64_t)b * g.nH + head;
                        bf16_t* dst = which == 0 ? g.Q + (bh * g.L + l) * 64 : g.Kd + (bh * g.Lcap + g.pos0 + l) * 64;
                        *reinterpret_cast<uint4*>(dst + rchunk * 8) = v;
                    }
                }
            }
        }
    } else {
        if constexpr (EPI == SHOWO_EPI_BF16 || EPI == SHOWO_EPI_GELU_BF16) {
            if (stg != nullptr && g.vec_out && !g.bias_per_row && (g.ldo % 8) == 0 && (g.N % 8) == 0 && ((((uintptr_t)g.out) & 15) == 0)) {
                const int nbase = n0 + wn * 64;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float bn[4];
                    load_bias4(g, nbase + i * 16 + fg * 4, bn);
#pragma unroll
                    for (int j = 0; j < MF; ++j) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bn[r] + 0.f;  // same expression as store_frag (bias per row = 0)
                        if (EPI == SHOWO_EPI_GELU_BF16) {
                            gelu4(v);
                        }
                        uint2 pk;
                        pk.x = Op16<F16>::pack2(v[0], v[1]);
                        pk.y = Op16<F16>::pack2(v[2], v[3]);
                        stage_frag_bf16(stg, j * 16 + fr, i, fg, pk);
                    }
                }
                bf16_t* o = reinterpret_cast<bf16_t*>(g.out);
#pragma unroll
                for (int t = 0; t < 2 * MF; ++t) {
                    const int row = t * 8 + rrow, m = mrow0 + row, n = nbase + rchunk * 8;
                    const uint4 v = unstage_row16(stg, row, rchunk);
                    if (m < g.M && n < g.N) *reinterpret_cast<uint4*>(o + (int64_t)m * g.ldo + n) = v;
                }
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + fg * 4;
            float bn[4];
            load_bias4(g, n, bn);
#pragma unroll
            for (int j = 0; j < MF; ++j) store_frag<EPI, F16>(g, acc[i][j], mrow0 + j * 16 + fr, n, bn);
        }
    }
}

// gelu_new at fp32 accuracy (x * sigmoid(2u) with IEEE exp and division; transformers NewGELUActivation, phi.py:204-212)
static __device__ __forceinline__ float gelu_new_precise(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return x / (1.0f + expf(-2.0f * u));
}
// (hi, lo) bf16 halves of four values: hi = RNE(v), lo = RNE(v - hi)
static __device__ __forceinline__ void split_pack4(const float (&v)[4], uint2& hi, uint2& lo) {
    hi.x = pack_bf2(v[0], v[1]);
    hi.y = pack_bf2(v[2], v[3]);
    lo.x = pack_bf2(v[0] - bf2f((bf16_t)(hi.x & 0xffffu)), v[1] - bf2f((bf16_t)(hi.x >> 16)));
    lo.y = pack_bf2(v[2] - bf2f((bf16_t)(hi.y & 0xffffu)), v[3] - bf2f((bf16_t)(hi.y >> 16)));
}

// Epilogue of the fused [Wqkv ; W1] projection in ACCURACY MODE (EPI_QKV_SPLIT; the operands were K-concatenated (hi, lo) images, so the
// fp32 accumulators hold the product to ~1e-6): the arithmetic of epilogue8p<EPI_QKV> -- bias, per-head q / k LayerNorm(64), partial
// RoPE, 1/8 folded into Q, head-major relayout of Q, K, V^T; bias + gelu_new for the fc1 columns -- with every output written as a
// (hi, lo) bf16 pair (g.Q / g.Qlo, ...) and gelu_new evaluated with IEEE exp / division instead of the fast forms.
// Inference only (no save-for-backward outputs).  Stores reuse the staging of the bf16 epilogue: one pass per half.
template <int MF>
static __device__ __forceinline__ void epilogue_qkv_split(const GemmArgs& g, f32x4 (&acc)[4][8], int n0, int wn, int mrow0, int fr, int fg,
                                                          bf16_t* stg) {
    const int lane_ = fg * 16 + fr;
    constexpr int VRS = MF < 8 ? 16 * MF + 2 : 128;
    const int rrow = lane_ >> 3, rchunk = lane_ & 7;
    const int nbase = n0 + wn * 64;
    if (n0 >= g.Nq) {  // fc1 columns: bias + gelu_new -> out2 / out2lo [m][n - Nq]
        const bool staged = stg != nullptr && (g.ldo2 % 8) == 0 && ((g.N - g.Nq) % 8) == 0 && ((((uintptr_t)g.out2) & 15) == 0) &&
                            ((((uintptr_t)g.out2lo) & 15) == 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float bn[4];
            load_bias4(g, nbase + i * 16 + fg * 4, bn);
#pragma unroll
            for (int j = 0; j < MF; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = gelu_new_precise(acc[i][j][r] + bn[r]);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            bf16_t* dst = half ? g.out2lo : g.out2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = nbase + i * 16 + fg * 4;
#pragma unroll
                for (int j = 0; j < MF; ++j) {
                    const int m = mrow0 + j * 16 + fr;
                    const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    uint2 hi, lo;
                    split_pack4(v, hi, lo);
                    const uint2 pk = half ? lo : hi;
                    if (staged) stage_frag_bf16(stg, j * 16 + fr, i, fg, pk);
                    else if (m < g.M && n < g.N) *reinterpret_cast<uint2*>(dst + (int64_t)m * g.ldo2 + (n - g.Nq)) = pk;
                }
            }
            if (staged) {
#pragma unroll
                for (int t = 0; t < 2 * MF; ++t) {
                    const int row = t * 8 + rrow, m = mrow0 + row, n = nbase + rchunk * 8;
                    const uint4 v = unstage_row16(stg, row, rchunk);
                    if (m < g.M && n < g.N) *reinterpret_cast<uint4*>(dst + (int64_t)m * g.ldo2 + (n - g.Nq)) = v;
                }
            }
        }
        return;
    }
    if (nbase >= g.N) return;
    const int Hq = g.nH * 64;
    const int which = nbase / Hq;  // 0 = q, 1 = k, 2 = v (wave-uniform)
    const int head = (nbase - which * Hq) >> 6;
    float bn[4][4], lw[4][4], lb[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        load_bias4(g, nbase + i * 16 + fg * 4, bn[i]);
        if (which < 2) {
            const float4 w4 = *reinterpret_cast<const float4*>((which ? g.kw : g.qw) + i * 16 + fg * 4);
            const float4 b4 = *reinterpret_cast<const float4*>((which ? g.kb : g.qb) + i * 16 + fg * 4);
            lw[i][0] = w4.x; lw[i][1] = w4.y; lw[i][2] = w4.z; lw[i][3] = w4.w;
            lb[i][0] = b4.x; lb[i][1] = b4.y; lb[i][2] = b4.z; lb[i][3] = b4.w;
        }
    }
    if (which == 2) {  // V^T[bh][d][pos], hi then lo: staged transposed in the wave's LDS slice, stored with the lanes along `pos`
        bf16_t* vb[2];
        bool vok[2];
        int64_t vdelta = g.Vtlo - g.Vt;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int mt = h * 64 + lane_, m = mrow0 + mt;
            vok[h] = mt < 16 * MF && m < g.M;
            const int mm = vok[h] ? m : 0;
            const int b = mm / g.L, l = mm - b * g.L;
            vb[h] = g.Vt + ((int64_t)b * g.nH + head) * 64 * g.Lp + g.pos0 + l;
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int j = 0; j < MF; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = acc[i][j][r] + bn[i][r];
                        const bf16_t hb = f2bf(x);
                        const bf16_t ob = half ? f2bf(x - bf2f(hb)) : hb;
                        if (stg) stg[(i * 16 + fg * 4 + r) * VRS + j * 16 + fr] = ob;
                        else {
                            const int m = mrow0 + j * 16 + fr;
                            if (m < g.M) {
                                const int b = m / g.L, l = m - b * g.L;
                                (half ? g.Vtlo : g.Vt)[(((int64_t)b * g.nH + head) * 64 + i * 16 + fg * 4 + r) * g.Lp + g.pos0 + l] = ob;
                            }
                        }
                    }
            if (stg) {
#pragma unroll 8
                for (int dd = 0; dd < 64; ++dd) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if (h * 64 < 16 * MF && vok[h]) (vb[h] + (half ? vdelta : 0))[(int64_t)dd * g.Lp] = stg[dd * VRS + h * 64 + lane_];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < MF; ++j) {
        const int m = mrow0 + j * 16 + fr;
        const bool valid = m < g.M;
        const int mm = valid ? m : g.M - 1;
        const int b = mm / g.L, l = mm - b * g.L, pos = g.pos0 + l;
        const int64_t bh = (int64_t)b * g.nH + head;
        float x[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[i][r] = acc[i][j][r] + bn[i][r];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) sum += (x[i][0] + x[i][1]) + (x[i][2] + x[i][3]);
        sum = sum4_lanes16(sum);
        const float mean = sum * (1.0f / 64.0f);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { x[i][r] -= mean; sq += x[i][r] * x[i][r]; }
        sq = sum4_lanes16(sq);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / 64.0f) + g.eps);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[i][r] = x[i][r] * rstd * lw[i][r] + lb[i][r];
        const float4 c0 = *reinterpret_cast<const float4*>(g.cosT + (int64_t)pos * 32 + fg * 4);
        const float4 s0 = *reinterpret_cast<const float4*>(g.sinT + (int64_t)pos * 32 + fg * 4);
        const float4 c1 = *reinterpret_cast<const float4*>(g.cosT + (int64_t)pos * 32 + 16 + fg * 4);
        const float4 s1 = *reinterpret_cast<const float4*>(g.sinT + (int64_t)pos * 32 + 16 + fg * 4);
        const float cc0[4] = {c0.x, c0.y, c0.z, c0.w}, ss0[4] = {s0.x, s0.y, s0.z, s0.w};
        const float cc1[4] = {c1.x, c1.y, c1.z, c1.w}, ss1[4] = {s1.x, s1.y, s1.z, s1.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float y0 = x[0][r], y1 = x[1][r];
            x[0][r] = y0 * cc0[r] - y1 * ss0[r];
            x[1][r] = y1 * cc1[r] + y0 * ss1[r];
        }
        if (!valid) continue;
        const float sc = which == 0 ? 0.125f : 1.0f;  // 1/sqrt(64) folded into Q (a power of two: exact for both halves)
        const int64_t off = which == 0 ? (bh * g.L + l) * 64 : (bh * g.Lcap + pos) * 64;
        bf16_t* dhi = (which == 0 ? g.Q : g.Kd) + off;
        bf16_t* dlo = (which == 0 ? g.Qlo : g.Klo) + off;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v[4] = {x[i][0] * sc, x[i][1] * sc, x[i][2] * sc, x[i][3] * sc};
            uint2 hi, lo;
            split_pack4(v, hi, lo);
            *reinterpret_cast<uint2*>(dhi + i * 16 + fg * 4) = hi;
            *reinterpret_cast<uint2*>(dlo + i * 16 + fg * 4) = lo;
        }
    }
}

// production kernel (gemm2p.hip)
int gemm2p_dispatch(GemmArgs g, int epilogue, hipStream_t s);  // epilogue: SHOWO_EPI_* or EPI_QKV
// split-K policy and per-stream workspace of the production family (gemm2p.hip), shared with gemm_tn.hip
int gemm_splitk_count(int M, int N, int K, int cus);  // cus = showo_cu_usable(stream)
bool gemm_splitk_ws(hipStream_t s, size_t need, float4** ws, unsigned** tick);
int gemm_splitk_ticks();
// (the cooperative split-K reduction is used by gemm2p's launches only -- check, launch and record under ONE hold of its mutex
//  (gemm2p_kernel.h launch2p, called from gemm2p_dispatch); the wrappers gemm_tn once used were removed in round 6: ADVICE r5)
void gemm_count_launch(bool split);
extern int g_gemm_gn, g_gemm_bm, g_gemm_pf, g_gemm_stage, g_gemm_splitk;
// m-split kernel with a 3-deep weight ring (gemm3w.hip); rows = 256 | 240 | 224 | 208
int gemm3w_launch(const GemmArgs& g, int epilogue, int rows, hipStream_t s);

}  // namespace showo
