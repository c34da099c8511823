// Backward of the fused omni-attention for gfx950 (training path).
// Replaces (reference): autograd of F.scaled_dot_product_attention with the dense additive mask
// (models/phi.py:715-722) inside loss.backward() (training/train.py:612).
//
// With S = Qs K^T (Qs = q / 8 as stored by the forward path), P = exp(S - lse) on the visible keys, D = rowsum(dO * O):
//     dV = P^T dO      dP = dO V^T      dS = P * (dP - D)      dQs = dS K      dK = dS^T Qs
// Two flash-style kernels recompute P from the saved log-sum-exp (nothing of size L x L is stored):
//   attn_bwd_dq_kernel   : a wave owns 32 query rows (lane = query column of the swapped products S^T, dP^T), loops over
//                          64-key tiles of K, V (rows) and K^T staged in LDS, accumulates dQ^T = K^T dS^T.
//   attn_bwd_dkv_kernel  : a wave owns 32 keys (lane = key column of S, dP), loops over 64-query tiles of Q, dO (rows) and
//                          Q^T, dO^T staged in LDS, accumulates dV^T = dO^T P and dK^T = Qs^T dS.
// Every MFMA operand is k-contiguous: the transposed images Q^T, K^T, dO^T ([B,nH,64,Lp]) are made by
// showo_head_transpose / showo_attn_bwd_prep (HBM-bound, a few tens of microseconds).  As in the forward kernel the tile
// whose rows index the MFMA A-operand of the first product is stored with its rows in the order pi (4-row blocks 1 and
// 2 of every 16 swapped), so that the 16 values a lane holds afterwards belong to rows 16(r>>3) + 8hh + (r&7) and its
// packed fragments multiply 8 CONSECUTIVE columns of the transposed operand.  LDS images: [64 rows][8 chunks of 16 B],
// chunk c of row r at c ^ ((r >> 1) & 7); staged by global_load_lds (source-side swizzle), double-buffered.
#include "common.h"
#include "../../include/showo_hip.h"
#include "prof.h"
#include <cstdlib>

using namespace showo;

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int BT = 64 * 64;  // bf16 elements of one 64 x 64 tile

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) { return pack_bf2(lo, hi); }  // compiler-visible conversion: see common.h
__device__ __forceinline__ bf16x8 pack8v(const float* p) {
    uint4 u;
    u.x = cvt_pk_bf16(p[0], p[1]); u.y = cvt_pk_bf16(p[2], p[3]); u.z = cvt_pk_bf16(p[4], p[5]); u.w = cvt_pk_bf16(p[6], p[7]);
    return __builtin_bit_cast(bf16x8, u);
}
// row order pi inside a 64-row tile: LDS row i holds source row (i & ~15) + 4 * {0,2,1,3}[(i >> 2) & 3] + (i & 3)
__device__ __forceinline__ int pi_row(int i) {
    const int blk = (i >> 2) & 3;
    return (i & ~15) + 4 * (((blk & 1) << 1) | (blk >> 1)) + (i & 3);
}

// ---- X [.., rows, 64] (row stride ld, head h at column h*64) -> XT [B, nH, 64, Lp] (zero padded columns)
__global__ __launch_bounds__(256) void head_transpose_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ xt, int L, int Lp,
                                                             int nH, int64_t bstride, int64_t hstride, int ld) {
    __shared__ bf16_t t[64][66];
    const int tid = threadIdx.x, l0 = blockIdx.x * 64, head = blockIdx.y, b = blockIdx.z;
    const bf16_t* src = x + (int64_t)b * bstride + (int64_t)head * hstride;
    for (int i = tid; i < 64 * 8; i += 256) {  // 64 rows x 8 chunks of 16 B
        const int r = i >> 3, c = i & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (l0 + r < L) v = *reinterpret_cast<const uint4*>(src + (int64_t)(l0 + r) * ld + c * 8);
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
        for (int j = 0; j < 8; ++j) t[r][c * 8 + j] = e[j];
    }
    __syncthreads();
    bf16_t* dst = xt + ((int64_t)b * nH + head) * 64 * Lp;
    const int d = tid >> 2, ls = (tid & 3) * 16;
    uint32_t w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = (uint32_t)t[ls + 2 * j][d] | ((uint32_t)t[ls + 2 * j + 1][d] << 16);
    uint4* o = reinterpret_cast<uint4*>(dst + (int64_t)d * Lp + l0 + ls);
    o[0] = make_uint4(w[0], w[1], w[2], w[3]);
    o[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// ---- D[b,h,q] = sum_d dO * O (both token-major [B*L, ld]); one wave per (token, head) row pair handled 4 heads at a time
__global__ __launch_bounds__(256) void attn_bwd_delta_kernel(const bf16_t* __restrict__ O, const bf16_t* __restrict__ dO,
                                                             float* __restrict__ D, int B, int L, int nH, int ld) {
    const int gid = blockIdx.x * 256 + threadIdx.x;  // one 8-lane group per (token, head): lane handles 8 dims
    const int grp = gid >> 3, sub = gid & 7;
    const int total = B * L * nH;
    const int g2 = grp < total ? grp : total - 1;
    const int tok = g2 / nH, head = g2 - tok * nH;
    const uint4 a = *reinterpret_cast<const uint4*>(O + (int64_t)tok * ld + head * 64 + sub * 8);
    const uint4 c = *reinterpret_cast<const uint4*>(dO + (int64_t)tok * ld + head * 64 + sub * 8);
    const bf16_t* ea = reinterpret_cast<const bf16_t*>(&a);
    const bf16_t* ec = reinterpret_cast<const bf16_t*>(&c);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += bf2f(ea[j]) * bf2f(ec[j]);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (sub == 0 && grp < total) {
        const int b = tok / L, l = tok - b * L;
        D[((int64_t)b * nH + head) * L + l] = s;
    }
}

struct BwdArgs {
    const bf16_t *Q, *K, *QT, *KT, *dOT;  // head-major [B,nH,L,64] / transposed [B,nH,64,Lp]
    const bf16_t* V; int ldv;             // V rows: token-major, row stride ldv, head h at column h*64 (raw qkv, v section)
    const bf16_t* dO; int lddo;           // token-major [B*L, lddo]
    const float *lse, *D;                 // [B,nH,L]
    const int32_t* iv;                    // [B,L,4] or NULL (causal)
    bf16_t *dQ, *dK, *dV; int ldq, ldk, ldvo;  // token-major outputs (row strides), head h at column h*64
    int B, nH, L, Lp;
    int nxb;  // > 0: 1-D grid in the XCD-aware order of the forward kernel (bwd_block_coords), nxb = blocks per (batch, head); 0: 3-D grid
};

// Blocks of one (batch, head) stream the same Q / dO / K / V tiles.  Consecutive block ids land on different XCDs (id % 8), each with
// its own L2, so the natural (x, head, batch) grid fetches every tile once per block.  As in attention.hip::attn_block_coords, 8 (batch,
// head) pairs form a group of 8 * nxb ids in which pair j owns j, j + 8, j + 16, ..: same XCD, dispatched back to back.
__device__ __forceinline__ void bwd_block_coords(const BwdArgs& a, int& xb, int& head, int& b) {
    if (a.nxb == 0) { xb = blockIdx.x; head = blockIdx.y; b = blockIdx.z; return; }
    const int lin = blockIdx.x, nbh = a.nH * a.B, per = 8 * a.nxb, full = nbh & ~7;
    int bh;
    if (lin < (full >> 3) * per) {
        const int grp = lin / per, rem = lin - grp * per;
        xb = rem >> 3;
        bh = grp * 8 + (rem & 7);
    } else {
        const int t = lin - (full >> 3) * per, tail = nbh - full;
        xb = t / tail;
        bh = full + (t - xb * tail);
    }
    b = bh / a.nH;
    head = bh - b * a.nH;
}

__device__ __forceinline__ void load_iv(const BwdArgs& a, int b, int q, int& lo1, int& hi1, int& lo2, int& hi2) {
    if (a.iv) {
        const int4 v = *reinterpret_cast<const int4*>(a.iv + ((int64_t)b * a.L + q) * 4);
        lo1 = v.x; hi1 = min(v.y, a.L); lo2 = v.z; hi2 = min(v.w, a.L);
    } else {
        lo1 = 0; hi1 = q + 1; lo2 = 0; hi2 = 0;
    }
}

// ================================================================================================ dQ
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(BwdArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t sm[2 * 3 * BT];  // [buf][K rows(pi) | V rows(pi) | K^T]
    __shared__ int s_hull[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int xb_, head, b;
    bwd_block_coords(a, xb_, head, b);
    const int qblk = xb_ * 4 + wave;
    const bool wactive = qblk * 32 < a.L;
    const int qi = lane & 31, hh = lane >> 5;
    const int qrow_raw = qblk * 32 + qi;
    const int qrow = qrow_raw < a.L ? qrow_raw : a.L - 1;
    const int64_t bh = (int64_t)b * a.nH + head;
    bf16x8 qf[4], dof[4];
    {
        const bf16_t* Qp = a.Q + (bh * a.L + qrow) * 64 + 8 * hh;
        const bf16_t* Dp = a.dO + ((int64_t)b * a.L + qrow) * a.lddo + head * 64 + 8 * hh;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            qf[m] = *reinterpret_cast<const bf16x8*>(Qp + 16 * m);
            dof[m] = *reinterpret_cast<const bf16x8*>(Dp + 16 * m);
        }
    }
    const float lse_l2 = a.lse[bh * a.L + qrow] * LOG2E;
    const float Dq = a.D[bh * a.L + qrow];
    int lo1, hi1, lo2, hi2;
    load_iv(a, b, qrow, lo1, hi1, lo2, hi2);
    if (!wactive || qrow_raw >= a.L) { lo1 = hi1 = lo2 = hi2 = 0; }
    const unsigned len1 = (unsigned)max(hi1 - lo1, 0), len2 = (unsigned)max(hi2 - lo2, 0);
    const int wmin = wave_min_i(min(lo1 < hi1 ? lo1 : 0x7fffffff, lo2 < hi2 ? lo2 : 0x7fffffff));
    const int wmax = wave_max_i(max(lo1 < hi1 ? hi1 : 0, lo2 < hi2 ? hi2 : 0));
    if (lane == 0) { s_hull[wave] = wmin; s_hull[4 + wave] = wmax; }
    __syncthreads();
    const int bmin = min(min(s_hull[0], s_hull[1]), min(s_hull[2], s_hull[3]));
    const int bmax = max(max(s_hull[4], s_hull[5]), max(s_hull[6], s_hull[7]));

    // staging: 3 tiles x 8 pieces; wave w issues pieces w and w + 4 of each
    const int prow = lane >> 3;
    const bf16_t* Kg = a.K + bh * a.L * 64;
    const bf16_t* Vg = a.V + (int64_t)b * a.L * a.ldv + head * 64;
    const bf16_t* KTg = a.KT + bh * 64 * a.Lp;
    int pik[2], kch[2];
    int64_t toff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 8 * (wave + 4 * i) + prow;
        pik[i] = pi_row(r);
        kch[i] = ((lane & 7) ^ ((r >> 1) & 7)) << 3;
        toff[i] = (int64_t)r * a.Lp + kch[i];
    }
#define DQ_STAGE(KT_, BUF)                                                                             \
    do {                                                                                               \
        bf16_t* s_ = sm + (BUF) * 3 * BT;                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                \
            int key_ = (KT_) + pik[i];                                                                 \
            key_ = key_ < a.L ? key_ : a.L - 1;                                                        \
            glds16_untracked(Kg + (int64_t)key_ * 64 + kch[i], lds_addr_of(s_ + (wave + 4 * i) * 512));           \
            glds16_untracked(Vg + (int64_t)key_ * a.ldv + kch[i], lds_addr_of(s_ + BT + (wave + 4 * i) * 512));   \
            glds16_untracked(KTg + toff[i] + (KT_), lds_addr_of(s_ + 2 * BT + (wave + 4 * i) * 512));             \
        }                                                                                              \
    } while (0)

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    const int fsw = (qi >> 1) & 7;
    const int kt0 = bmin >= 0x7fffffff ? 0 : (bmin & ~63);
    if (kt0 < bmax) DQ_STAGE(kt0, 0);
    // retire every tracked load before the loop, registers named (see attention.hip: a pending one would put a vmcnt(0) inside the loop)
    float lse_p = lse_l2, Dq_p = Dq;
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]), "+v"(dof[0]), "+v"(dof[1]), "+v"(dof[2]), "+v"(dof[3]), "+v"(lse_p),
                   "+v"(Dq_p), "+v"(lo1), "+v"(hi1), "+v"(lo2), "+v"(hi2)::"memory");
    __syncthreads();
    int buf = 0;
    for (int kt = kt0; kt < bmax; kt += 64, buf ^= 1) {
        const bool more = kt + 64 < bmax;
        if (more) DQ_STAGE(kt + 64, buf ^ 1);
        const bf16_t* sK = sm + buf * 3 * BT;
        const bf16_t* sV = sK + BT;
        const bf16_t* sKT = sK + 2 * BT;
#pragma unroll 1
        for (int sub = 0; sub < 2; ++sub) {
            const int ks = kt + 32 * sub;
            if (ks >= wmax || ks + 32 <= wmin) continue;
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int off = (32 * sub + qi) * 64 + (((2 * m + hh) ^ fsw) << 3);
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + off);
                bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + off);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[m], s, 0, 0, 0);     // S^T[key][q]
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[m], dp, 0, 0, 0);  // dP^T[key][q]
            }
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = ks + 16 * (r >> 3) + 8 * hh + (r & 7);
                const bool vis = ((unsigned)(key - lo1) < len1) | ((unsigned)(key - lo2) < len2);
                const float p = vis ? __builtin_amdgcn_exp2f(fmaf(s[r], LOG2E, -lse_p)) : 0.f;
                ds[r] = p * (dp[r] - Dq_p);
            }
            bf16x8 db0 = pack8v(ds), db1 = pack8v(ds + 8);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int c = ((4 * sub + 2 * kk + hh) ^ fsw) << 3;
                bf16x8 t0 = *reinterpret_cast<const bf16x8*>(sKT + qi * 64 + c);
                bf16x8 t1 = *reinterpret_cast<const bf16x8*>(sKT + (32 + qi) * 64 + c);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t0, kk ? db1 : db0, o0, 0, 0, 0);  // dQ^T[d][q]
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t1, kk ? db1 : db0, o1, 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (!wactive || qrow_raw >= a.L) return;
    bf16_t* op = a.dQ + ((int64_t)b * a.L + qrow_raw) * a.ldq + head * 64 + 4 * hh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint2 w0, w1;
        w0.x = cvt_pk_bf16(o0[4 * g], o0[4 * g + 1]); w0.y = cvt_pk_bf16(o0[4 * g + 2], o0[4 * g + 3]);
        w1.x = cvt_pk_bf16(o1[4 * g], o1[4 * g + 1]); w1.y = cvt_pk_bf16(o1[4 * g + 2], o1[4 * g + 3]);
        *reinterpret_cast<uint2*>(op + 8 * g) = w0;
        *reinterpret_cast<uint2*>(op + 32 + 8 * g) = w1;
    }
#undef DQ_STAGE
}

// ================================================================================================ dK, dV
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(BwdArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t sm[2 * 4 * BT];  // [buf][Q rows(pi) | dO rows(pi) | Q^T | dO^T]
    __shared__ float s_lse[2][64], s_D[2][64];
    __shared__ int4 s_iv[2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int xb_, head, b;
    bwd_block_coords(a, xb_, head, b);
    const int kblk = xb_ * 4 + wave;
    const int ki = lane & 31, hh = lane >> 5;
    const int key_raw = kblk * 32 + ki;
    const int key = key_raw < a.L ? key_raw : a.L - 1;
    const bool kvalid = key_raw < a.L;
    const int64_t bh = (int64_t)b * a.nH + head;
    bf16x8 kf[4], vf[4];  // B operands: lane = key column, 8 consecutive d
    {
        const bf16_t* Kp = a.K + (bh * a.L + key) * 64 + 8 * hh;
        const bf16_t* Vp = a.V + ((int64_t)b * a.L + key) * a.ldv + head * 64 + 8 * hh;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            kf[m] = *reinterpret_cast<const bf16x8*>(Kp + 16 * m);
            vf[m] = *reinterpret_cast<const bf16x8*>(Vp + 16 * m);
        }
    }

    const int prow = lane >> 3;
    const bf16_t* Qg = a.Q + bh * a.L * 64;
    const bf16_t* dOg = a.dO + (int64_t)b * a.L * a.lddo + head * 64;
    const bf16_t* QTg = a.QT + bh * 64 * a.Lp;
    const bf16_t* dOTg = a.dOT + bh * 64 * a.Lp;
    int piq[2], qch[2];
    int64_t toff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 8 * (wave + 4 * i) + prow;
        piq[i] = pi_row(r);
        qch[i] = ((lane & 7) ^ ((r >> 1) & 7)) << 3;
        toff[i] = (int64_t)r * a.Lp + qch[i];
    }
    // The tile DMAs are untracked (common.h): nothing the compiler waits for may be in flight between their issue and the end-of-tile
    // wait.  The per-query scalars of the NEXT tile (lse, D, intervals) are therefore loaded into registers up front (DKV_LOAD, ahead
    // of the DMAs) and only stored to LDS at the end of the tile (DKV_STORE, next to the explicit wait).
    float st_lse = 0.f, st_D = 0.f;
    int4 st_iv = make_int4(0, 0, 0, 0);
#define DKV_LOAD(QT_)                                                                                  \
    if (tid < 64) {                                                                                    \
        const int q_ = (QT_) + tid;                                                                    \
        const bool ok_ = q_ < a.L;                                                                     \
        const int qc_ = ok_ ? q_ : a.L - 1;                                                            \
        st_lse = a.lse[bh * a.L + qc_] * LOG2E;                                                        \
        st_D = ok_ ? a.D[bh * a.L + qc_] : 0.f;                                                        \
        int l1_, h1_, l2_, h2_;                                                                        \
        load_iv(a, b, qc_, l1_, h1_, l2_, h2_);                                                        \
        if (!ok_) { l1_ = h1_ = l2_ = h2_ = 0; }                                                       \
        st_iv = make_int4(l1_, h1_, l2_, h2_);                                                         \
    }
#define DKV_STORE(BUF)                                                                                 \
    if (tid < 64) { s_lse[BUF][tid] = st_lse; s_D[BUF][tid] = st_D; s_iv[BUF][tid] = st_iv; }
#define DKV_STAGE(QT_, BUF)                                                                            \
    do {                                                                                               \
        bf16_t* s_ = sm + (BUF) * 4 * BT;                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                \
            int q_ = (QT_) + piq[i];                                                                   \
            q_ = q_ < a.L ? q_ : a.L - 1;                                                              \
            glds16_untracked(Qg + (int64_t)q_ * 64 + qch[i], lds_addr_of(s_ + (wave + 4 * i) * 512));                 \
            glds16_untracked(dOg + (int64_t)q_ * a.lddo + qch[i], lds_addr_of(s_ + BT + (wave + 4 * i) * 512));       \
            glds16_untracked(QTg + toff[i] + (QT_), lds_addr_of(s_ + 2 * BT + (wave + 4 * i) * 512));                 \
            glds16_untracked(dOTg + toff[i] + (QT_), lds_addr_of(s_ + 3 * BT + (wave + 4 * i) * 512));                \
        }                                                                                              \
    } while (0)

    f32x16 dv0, dv1, dk0, dk1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dv0[r] = 0.f; dv1[r] = 0.f; dk0[r] = 0.f; dk1[r] = 0.f; }
    const int fsw = (ki >> 1) & 7;
    const int nqt = (a.L + 63) / 64;
    DKV_LOAD(0)
    DKV_STAGE(0, 0);
    DKV_STORE(0)
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3])::"memory");
    __syncthreads();
    int buf = 0;
    for (int qt = 0; qt < nqt; ++qt, buf ^= 1) {
        const bool more = qt + 1 < nqt;
        if (more) {
            DKV_LOAD((qt + 1) * 64)
            // the scalar loads above are the only tracked VMEM of the loop: retire them HERE, before the DMAs are issued, so that no
            // compiler-placed wait (which would be a vmcnt(0): it does not count the DMAs) lands between the DMA issue and the tile
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(st_lse), "+v"(st_D), "+v"(st_iv.x), "+v"(st_iv.y), "+v"(st_iv.z), "+v"(st_iv.w)::"memory");
            DKV_STAGE((qt + 1) * 64, buf ^ 1);
        }
        const bf16_t* sQ = sm + buf * 4 * BT;
        const bf16_t* sdO = sQ + BT;
        const bf16_t* sQT = sQ + 2 * BT;
        const bf16_t* sdOT = sQ + 3 * BT;
#pragma unroll 1
        for (int sub = 0; sub < 2; ++sub) {
            const int qs = 32 * sub;  // tile-relative first query row of the sub-tile
            // visibility of this lane's key for its 16 query rows
            unsigned vmask = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int4 v = s_iv[buf][qs + 16 * (r >> 3) + 8 * hh + (r & 7)];
                const bool vis = kvalid & ((((unsigned)(key - v.x) < (unsigned)max(v.y - v.x, 0))) | ((unsigned)(key - v.z) < (unsigned)max(v.w - v.z, 0)));
                vmask |= vis ? (1u << r) : 0u;
            }
            if (!__any(vmask != 0)) continue;  // wave-uniform: no (query, key) pair of this sub-tile is visible
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int off = (qs + ki) * 64 + (((2 * m + hh) ^ fsw) << 3);
                bf16x8 qa = *reinterpret_cast<const bf16x8*>(sQ + off);
                bf16x8 da = *reinterpret_cast<const bf16x8*>(sdO + off);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[m], s, 0, 0, 0);    // S[q][key]
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[m], dp, 0, 0, 0);  // dP[q][key]
            }
            float p[16], ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = qs + 16 * (r >> 3) + 8 * hh + (r & 7);
                const float pe = __builtin_amdgcn_exp2f(fmaf(s[r], LOG2E, -s_lse[buf][qq]));
                p[r] = ((vmask >> r) & 1u) ? pe : 0.f;
                ds[r] = p[r] * (dp[r] - s_D[buf][qq]);
            }
            bf16x8 pb0 = pack8v(p), pb1 = pack8v(p + 8), db0 = pack8v(ds), db1 = pack8v(ds + 8);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int c = ((4 * sub + 2 * kk + hh) ^ fsw) << 3;
                bf16x8 t0 = *reinterpret_cast<const bf16x8*>(sdOT + ki * 64 + c);
                bf16x8 t1 = *reinterpret_cast<const bf16x8*>(sdOT + (32 + ki) * 64 + c);
                dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t0, kk ? pb1 : pb0, dv0, 0, 0, 0);  // dV^T[d][key]
                dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t1, kk ? pb1 : pb0, dv1, 0, 0, 0);
                bf16x8 u0 = *reinterpret_cast<const bf16x8*>(sQT + ki * 64 + c);
                bf16x8 u1 = *reinterpret_cast<const bf16x8*>(sQT + (32 + ki) * 64 + c);
                dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u0, kk ? db1 : db0, dk0, 0, 0, 0);  // dK^T[d][key]
                dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u1, kk ? db1 : db0, dk1, 0, 0, 0);
            }
        }
        if (more) { DKV_STORE(buf ^ 1) }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (!kvalid) return;
    const int64_t tok = (int64_t)b * a.L + key_raw;
    bf16_t* vp = a.dV + tok * a.ldvo + head * 64 + 4 * hh;
    bf16_t* kp = a.dK + tok * a.ldk + head * 64 + 4 * hh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = cvt_pk_bf16(dv0[4 * g], dv0[4 * g + 1]); w.y = cvt_pk_bf16(dv0[4 * g + 2], dv0[4 * g + 3]);
        *reinterpret_cast<uint2*>(vp + 8 * g) = w;
        w.x = cvt_pk_bf16(dv1[4 * g], dv1[4 * g + 1]); w.y = cvt_pk_bf16(dv1[4 * g + 2], dv1[4 * g + 3]);
        *reinterpret_cast<uint2*>(vp + 32 + 8 * g) = w;
        w.x = cvt_pk_bf16(dk0[4 * g], dk0[4 * g + 1]); w.y = cvt_pk_bf16(dk0[4 * g + 2], dk0[4 * g + 3]);
        *reinterpret_cast<uint2*>(kp + 8 * g) = w;
        w.x = cvt_pk_bf16(dk1[4 * g], dk1[4 * g + 1]); w.y = cvt_pk_bf16(dk1[4 * g + 2], dk1[4 * g + 3]);
        *reinterpret_cast<uint2*>(kp + 32 + 8 * g) = w;
    }
#undef DKV_STAGE
#undef DKV_STORE
#undef DKV_LOAD
}

}  // namespace

extern "C" int showo_head_transpose(const uint16_t* x, uint16_t* xt, int B, int nH, int L, int Lp, int64_t batch_stride,
                                    int64_t head_stride, int row_stride, void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if ((Lp % 64) || Lp < L || (row_stride % 8)) return set_error_msg(1, "head_transpose: bad Lp / row stride");
    head_transpose_kernel<<<dim3(Lp / 64, nH, B), dim3(256), 0, (hipStream_t)stream>>>(x, xt, L, Lp, nH, batch_stride, head_stride,
                                                                                   row_stride);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int showo_attn_bwd(const uint16_t* Q, const uint16_t* K, const uint16_t* QT, const uint16_t* KT, const uint16_t* V,
                              int ldv, const uint16_t* O, const uint16_t* dO, int lddo, uint16_t* dOT, const float* lse, float* D,
                              const int32_t* iv, const int32_t* flag, uint16_t* dQ, int ldq, uint16_t* dK, int ldk, uint16_t* dV,
                              int ldvo, int B, int nH, int L, int Lp, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (B <= 0 || L <= 0) return 0;
    if ((Lp % 64) || Lp < L) return set_error_msg(1, "attn_bwd: bad Lp");
    if ((ldv % 8) || (lddo % 8) || (ldq % 4) || (ldk % 4) || (ldvo % 4)) return set_error_msg(1, "attn_bwd: bad row strides");
    if (flag) {  // the backward implements interval masks only (every mask the reference builds is one)
        int32_t f = 0;
        SHOWO_CHECK_HIP(hipMemcpyAsync(&f, flag, 4, hipMemcpyDeviceToHost, s));
        SHOWO_CHECK_HIP(hipStreamSynchronize(s));
        if (f) return set_error_msg(6, "attn_bwd: mask is not interval-representable");
    }
    // D = rowsum(dO * O), dO^T
    const int64_t groups = (int64_t)B * L * nH * 8;
    attn_bwd_delta_kernel<<<dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s>>>(O, dO, D, B, L, nH, lddo);
    head_transpose_kernel<<<dim3(Lp / 64, nH, B), dim3(256), 0, s>>>(dO, dOT, L, Lp, nH, (int64_t)L * lddo, 64, lddo);
    BwdArgs a;
    a.Q = Q; a.K = K; a.QT = QT; a.KT = KT; a.dOT = dOT; a.V = V; a.ldv = ldv; a.dO = dO; a.lddo = lddo;
    a.lse = lse; a.D = D; a.iv = iv; a.dQ = dQ; a.dK = dK; a.dV = dV; a.ldq = ldq; a.ldk = ldk; a.ldvo = ldvo;
    a.B = B; a.nH = nH; a.L = L; a.Lp = Lp;
    ProfScope prof(PROF_ATTN, 10.0 * B * nH * (double)L * L * 64, s);  // 5 L x L x 64 products
    const int blocks = ((L + 31) / 32 + 3) / 4;
    static int xcd = -1;  // SHOWO_ATTN_XCD (default 1), shared with the forward kernel
    if (xcd < 0) { const char* e = getenv("SHOWO_ATTN_XCD"); xcd = e ? atoi(e) : 1; }
    if (xcd && (int64_t)blocks * nH * B < ((int64_t)1 << 31)) {
        a.nxb = blocks;
        attn_bwd_dq_kernel<<<dim3(blocks * nH * B), dim3(256), 0, s>>>(a);
        attn_bwd_dkv_kernel<<<dim3(blocks * nH * B), dim3(256), 0, s>>>(a);
    } else {
        a.nxb = 0;
        attn_bwd_dq_kernel<<<dim3(blocks, nH, B), dim3(256), 0, s>>>(a);
        attn_bwd_dkv_kernel<<<dim3(blocks, nH, B), dim3(256), 0, s>>>(a);
    }
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
