// Omni-attention for gfx950: q/k LayerNorm + partial RoPE + relayout, mask -> interval compression, and a
// fused flash-style attention forward whose mask is two visibility intervals per query row.
//
// Replaces (reference): models/phi.py:661-694 (view/transpose, q_layernorm/k_layernorm, partial rotary),
// models/phi.py:715-722 (SDPA with the dense additive [B,1,L,L] mask built by
// training/prompting_utils.py:466-511, 591-624).  Every mask those builders produce has, per query row r,
// the form  visible(c) = lo1<=c<hi1  or  lo2<=c<hi2  (SURVEY.md §8a A5), so the 4*L*L-byte mask is read
// once by showo_mask_compress instead of once per layer; anything else falls back to adding the dense mask.
//
// Attention kernel structure: one wave = 32 query rows, "swapped" QK^T (S^T = K Q^T on
// v_mfma_f32_32x32x16_bf16) so that a lane owns one query row: the online-softmax row max / row sum are
// 15 in-lane ops + one cross-half shuffle, and the probabilities are already in MFMA B-operand order for
// O^T = V^T P^T.  V is consumed from a transposed image Vt[d][key] written by showo_qk_prep, so the
// A-operand of the PV product is two contiguous 8-byte loads per lane.  K/V tiles are read straight from
// L2 (one head's K+V is 97 KB at L=387), waves of a block are independent (no LDS, no barriers).
#include "common.h"
#include "decode_common.h"
#include <type_traits>
#include "../../include/showo_hip.h"
#include "prof.h"
#include <cfloat>
#include <cstdlib>

using namespace showo;

namespace {

constexpr float LOG2E = 1.4426950408889634f;

// ------------------------------------------------------------------------------------------------
// q/k LayerNorm(64) + partial RoPE + relayout.  grid (ceil(L/64), nH, B), block 256.
// ------------------------------------------------------------------------------------------------
struct PrepArgs {
    const bf16_t* qkv;
    const float *qw, *qb, *kw, *kb, *cosT, *sinT;
    bf16_t *Q, *K, *Vt;
    int B, L, nH, rot, pos0, Lcap, Lp;
    float eps;
    const int* pos_dev;  // graph replay of a decode step: position of the first new token read on the device
};

__device__ inline float ln_rope_lane(float x, float w, float b, float eps, const float* cosr, const float* sinr, int rot, int d) {
    float mean = wave_sum(x) * (1.0f / 64.0f);
    float c = x - mean;
    float var = wave_sum(c * c) * (1.0f / 64.0f);
    float y = c * (1.0f / sqrtf(var + eps)) * w + b;
    // rotate_half over dims [0, rot): pair d with d +- rot/2 (phi.py:163-167)
    int half = rot >> 1;
    float partner = __shfl_xor(y, half, 64);  // valid pairing because rot/2 is a power of two (16)
    if (d < rot) {
        float r = (d < half) ? -partner : partner;
        y = y * cosr[d] + r * sinr[d];
    }
    return y;
}

template <bool F16>
__global__ __launch_bounds__(256) void qk_prep_kernel(PrepArgs a) {
    __shared__ bf16_t sV[64][66];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l0 = blockIdx.x * 64, head = blockIdx.y, b = blockIdx.z;
    if (a.pos_dev) a.pos0 = *a.pos_dev;
    const int H3 = 3 * a.nH * 64, Hq = a.nH * 64;
    const bool plain = a.qw == nullptr;
    const float qw = plain ? 1.f : a.qw[lane], qb = plain ? 0.f : a.qb[lane], kw = plain ? 1.f : a.kw[lane], kb = plain ? 0.f : a.kb[lane];
    for (int i = 0; i < 16; ++i) {
        int l = l0 + wave * 16 + i;
        if (l >= a.L) break;
        const bf16_t* row = a.qkv + ((int64_t)b * a.L + l) * H3 + head * 64;
        int pos = a.pos0 + l;
        const float* cosr = a.cosT + (int64_t)pos * a.rot;
        const float* sinr = a.sinT + (int64_t)pos * a.rot;
        // plain heads (qw == NULL: a standard multi-head attention such as the CLIP tower's): only the 1/8 scale and the relayout
        float q = plain ? Op16<F16>::tof(row[lane]) : ln_rope_lane(Op16<F16>::tof(row[lane]), qw, qb, a.eps, cosr, sinr, a.rot, lane);
        float k = plain ? Op16<F16>::tof(row[Hq + lane]) : ln_rope_lane(Op16<F16>::tof(row[Hq + lane]), kw, kb, a.eps, cosr, sinr, a.rot, lane);
        a.Q[(((int64_t)b * a.nH + head) * a.L + l) * 64 + lane] = Op16<F16>::cvt(q * 0.125f);  // 1/sqrt(64): a power of two, exact in either type
        a.K[(((int64_t)b * a.nH + head) * a.Lcap + pos) * 64 + lane] = Op16<F16>::cvt(k);
        sV[wave * 16 + i][lane] = row[2 * Hq + lane];
    }
    __syncthreads();
    bf16_t* vt = a.Vt + ((int64_t)b * a.nH + head) * 64 * a.Lp;
    if (a.pos0 == 0) {
        // transposed tile write: thread -> (d, 16 consecutive tokens), zero beyond L so pad keys are finite
        int d = tid >> 2, ls = (tid & 3) * 16;
        uint32_t w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int la = ls + 2 * j, lb = la + 1;
            uint32_t lo = (l0 + la < a.L) ? sV[la][d] : 0;
            uint32_t hi = (l0 + lb < a.L) ? sV[lb][d] : 0;
            w[j] = lo | (hi << 16);
        }
        uint4* dst = reinterpret_cast<uint4*>(vt + (int64_t)d * a.Lp + l0 + ls);
        dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
        dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    } else {
        // append path (KV-cache decode): few tokens, scalar column writes
        for (int idx = tid; idx < 64 * 64; idx += 256) {
            int lt = idx >> 6, d = idx & 63;
            if (l0 + lt < a.L) vt[(int64_t)d * a.Lp + a.pos0 + l0 + lt] = sV[lt][d];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// mask -> intervals.  One wave per (b, row); 64 columns per ballot.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_compress_kernel(const float* __restrict__ mask, int32_t* __restrict__ iv,
                                                            int32_t* __restrict__ flag, int rows, int Lk) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const float* mr = mask + (int64_t)r * Lk;
    int runs = 0, bad = 0;
    int lo[2] = {0, 0}, hi[2] = {0, 0};
    bool open = false;  // a visible run is open at the chunk boundary (lane-uniform state, kept by all lanes)
    for (int c0 = 0; c0 < Lk; c0 += 64) {
        int c = c0 + lane;
        float v = (c < Lk) ? mr[c] : -FLT_MAX;
        bool vis = (c < Lk) && (v == 0.0f);
        bool odd = (c < Lk) && !vis && !(v <= -1.0e9f);  // neither 0 nor "minus infinity"-like (NaN included)
        unsigned long long bits = __ballot(vis);
        if (__ballot(odd)) bad = 1;
        int pos = 0;
        while (pos < 64) {
            if (!open) {
                unsigned long long rest = bits >> pos;
                if (rest == 0) break;
                int s = __ffsll((long long)rest) - 1;
                pos += s;
                if (runs < 2) lo[runs] = c0 + pos;
                open = true;
            } else {
                unsigned long long rest = (~bits) >> pos;
                if (rest == 0) { pos = 64; break; }
                int s = __ffsll((long long)rest) - 1;
                pos += s;
                if (runs < 2) hi[runs] = c0 + pos;
                runs++;
                open = false;
            }
        }
    }
    if (open) {
        if (runs < 2) hi[runs] = Lk;
        runs++;
    }
    if (lane == 0) {
        if (runs > 2 || bad) atomicOr(flag, 1);
        int4 o = make_int4(lo[0], runs > 0 ? hi[0] : 0, runs > 1 ? lo[1] : 0, runs > 1 ? hi[1] : 0);
        *reinterpret_cast<int4*>(iv + (int64_t)r * 4) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// fused attention forward
// ------------------------------------------------------------------------------------------------
struct AttnArgs {
    const bf16_t *Q, *K, *Vt;
    const int32_t* iv;
    const int32_t* flag;
    const float* dense;
    bf16_t* O;
    int B, nH, Lq, Lk, Lcap, Lp, ldo;
    float* lse;  // optional (training): log-sum-exp of every score row, fp32 [B, nH, Lq]
    const int* pos_dev;  // graph replay of a decode step (Lq = 1): Lk = *pos_dev + 1 read on the device
    int nqb = 0;  // LDS-tiled form launched as a 1-D grid of nqb * nH * B blocks in the XCD-aware order below (0: 3-D grid (qb, head, b))
    // accuracy mode (attn_lds_body<SPLIT>): low halves of the operands (same layouts) and of the output
    const bf16_t *Qlo = nullptr, *Klo = nullptr, *Vtlo = nullptr;
    bf16_t* Olo = nullptr;
};

// XCD-aware block order of the LDS-tiled forward.  Hardware places block `id` on XCD `id % 8`, each XCD has its own 4 MiB L2.  With
// the natural (qb, head, b) grid the q-blocks of one (batch, head) -- which all stream the SAME K / V^T tiles -- have consecutive ids
// and land on different XCDs: every tile crosses the fabric once per q-block (PMC r2: 158 MB fetched per launch against 68 MB of
// Q / K / V^T at Lq = 258, where a (batch, head) has 3 q-blocks).  Here 8 (batch, head) pairs form a group of 8 * nqb blocks in which
// pair j owns the ids j, j + 8, j + 16, ...: same XCD, dispatched back to back, so the second and third q-block find the tiles in
// that XCD's L2.  Pairs beyond the last full group of 8 keep the natural order.
__device__ __forceinline__ void attn_block_coords(const AttnArgs& a, int& qb, int& head, int& b) {
    if (a.nqb == 0) { qb = blockIdx.x; head = blockIdx.y; b = blockIdx.z; return; }
    const int lin = blockIdx.x, nbh = a.nH * a.B, per = 8 * a.nqb, full = nbh & ~7;
    int bh;
    if (lin < (full >> 3) * per) {
        const int grp = lin / per, rem = lin - grp * per;
        qb = rem >> 3;
        bh = grp * 8 + (rem & 7);
    } else {
        const int t = lin - (full >> 3) * per, tail = nbh - full;
        qb = t / tail;
        bh = full + (t - qb * tail);
    }
    b = bh / a.nH;
    head = bh - b * a.nH;
}

template <bool F16>
__device__ inline bf16x8 pack8(const float* p) {
    uint4 u;
    u.x = Op16<F16>::pack2_bounded(p[0], p[1]);
    u.y = Op16<F16>::pack2_bounded(p[2], p[3]);
    u.z = Op16<F16>::pack2_bounded(p[4], p[5]);
    u.w = Op16<F16>::pack2_bounded(p[6], p[7]);
    return __builtin_bit_cast(bf16x8, u);
}

template <bool F16>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
    if (a.pos_dev) a.Lk = *a.pos_dev + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qblk = blockIdx.x * 4 + wave;
    if (qblk * 32 >= a.Lq) return;
    const int head = blockIdx.y, b = blockIdx.z;
    const int qi = lane & 31, hh = lane >> 5;
    const int qrow_raw = qblk * 32 + qi;
    const int qrow = qrow_raw < a.Lq ? qrow_raw : a.Lq - 1;
    const int64_t bh = (int64_t)b * a.nH + head;

    const bf16_t* Qp = a.Q + (bh * a.Lq + qrow) * 64 + 8 * hh;
    bf16x8 qf[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) qf[m] = *reinterpret_cast<const bf16x8*>(Qp + 16 * m);

    const bool dense = (a.flag != nullptr) && (a.dense != nullptr) && (*a.flag != 0);
    int lo1, hi1, lo2, hi2;
    if (dense) {
        lo1 = 0; hi1 = a.Lk; lo2 = 0; hi2 = 0;
    } else if (a.iv) {
        int4 v = *reinterpret_cast<const int4*>(a.iv + ((int64_t)b * a.Lq + qrow) * 4);
        lo1 = v.x; hi1 = v.y; lo2 = v.z; hi2 = v.w;
    } else {  // no mask given: causal (SDPA is_causal path, phi.py:713)
        lo1 = 0; hi1 = qrow + 1 + (a.Lk - a.Lq); lo2 = 0; hi2 = 0;
    }
    hi1 = min(hi1, a.Lk);
    hi2 = min(hi2, a.Lk);
    int kmin = wave_min_i(min(lo1 < hi1 ? lo1 : 0x7fffffff, lo2 < hi2 ? lo2 : 0x7fffffff));
    int kmax = wave_max_i(max(lo1 < hi1 ? hi1 : 0, lo2 < hi2 ? hi2 : 0));
    const float* drow = dense ? a.dense + ((int64_t)b * a.Lq + qrow) * a.Lk : nullptr;

    const bf16_t* Kb = a.K + bh * a.Lcap * 64 + 8 * hh;
    const bf16_t* Vb = a.Vt + bh * 64 * a.Lp + 4 * hh;

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }

    for (int kt = (kmin >= 0x7fffffff ? 0 : (kmin & ~31)); kt < kmax; kt += 32) {
        int krow = kt + qi;
        krow = krow < a.Lk ? krow : a.Lk - 1;
        const bf16_t* Kp = Kb + (int64_t)krow * 64;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kp + 16 * m);
            s = Op16<F16>::mfma32(kf, qf[m], s);
        }
        float sv[16];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int key = kt + (r & 3) + 8 * (r >> 2) + 4 * hh;
            bool vis = ((key >= lo1) & (key < hi1)) | ((key >= lo2) & (key < hi2));
            float x = s[r];
            if (dense) x += (key < a.Lk) ? drow[key] : 0.f;
            sv[r] = vis ? x : -INFINITY;
            mx = fmaxf(mx, sv[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float m_new = fmaxf(m_run, mx);
        float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * LOG2E);
        float p[16];
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = __builtin_amdgcn_exp2f((sv[r] - m_use) * LOG2E);
            ps += p[r];
        }
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        bf16x8 pb0 = pack8<F16>(p), pb1 = pack8<F16>(p + 8);
        // O^T[d][q] += Vt[d][keys] * P^T[keys][q]; lane's 8 k-slots of product kk are keys
        // kt + 16kk + 4hh + {0..3} and kt + 16kk + 8 + 4hh + {0..3}  (same keys as p[8kk .. 8kk+7])
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16_t* v0 = Vb + (int64_t)qi * a.Lp + kt + 16 * kk;
            const bf16_t* v1 = Vb + (int64_t)(32 + qi) * a.Lp + kt + 16 * kk;
            uint2 a0 = *reinterpret_cast<const uint2*>(v0), a1 = *reinterpret_cast<const uint2*>(v0 + 8);
            uint2 b0 = *reinterpret_cast<const uint2*>(v1), b1 = *reinterpret_cast<const uint2*>(v1 + 8);
            bf16x8 vf0 = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
            bf16x8 vf1 = __builtin_bit_cast(bf16x8, make_uint4(b0.x, b0.y, b1.x, b1.y));
            o0 = Op16<F16>::mfma32(vf0, kk ? pb1 : pb0, o0);
            o1 = Op16<F16>::mfma32(vf1, kk ? pb1 : pb0, o1);
        }
    }
    float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    float inv = 1.0f / l_tot;
    if (qrow_raw < a.Lq) {
        bf16_t* op = a.O + ((int64_t)b * a.Lq + qrow_raw) * a.ldo + head * 64 + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w0, w1;
            w0.x = Op16<F16>::pack2_bounded(o0[4 * g] * inv, o0[4 * g + 1] * inv);
            w0.y = Op16<F16>::pack2_bounded(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            w1.x = Op16<F16>::pack2_bounded(o1[4 * g] * inv, o1[4 * g + 1] * inv);
            w1.y = Op16<F16>::pack2_bounded(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(op + 8 * g) = w0;        // d = 8g + 4hh + {0..3}
            *reinterpret_cast<uint2*>(op + 32 + 8 * g) = w1;   // d = 32 + 8g + 4hh + {0..3}
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-tiled form (prefill / t2i): block = 4 waves = 128 query rows of one (b, head); K and V^T are streamed
// through LDS in 64-key tiles shared by the 4 waves (coalesced 16-B global loads, register-staged,
// double-buffered: the loads of tile t+1 are in flight under the MFMAs of tile t; one barrier per tile).
// The per-wave math is that of attn_fwd_kernel.  The gather form above spends its time in the texture
// addresser (every lane of a V^T load touches its own 128-B line); here a tile costs 16 fully coalesced
// wave-loads per block.
// LDS image of both tiles: [64 rows][8 chunks of 16 B], chunk c of row r stored at c ^ ((r >> 1) & 7): a
// ds_read_b128 of 32 consecutive rows at one logical chunk (the 32x32x16 operand fetch) is conflict-free.
// K rows = keys in the order pi (key blocks 4-7 and 8-11 of every 16 swapped), so that the 16 scores a lane holds
// after the swapped QK^T (C-layout rows (r&3) + 8(r>>2) + 4hh) are the keys 16(r>>3) + 8hh + (r&7): its two
// P fragments then multiply 8 CONSECUTIVE keys each, one 16-B chunk of the natural-order V^T rows.
// ------------------------------------------------------------------------------------------------
constexpr int AT_TILE = 64 * 64;  // bf16 elements of one K (or V^T) tile
constexpr float AT_DEFER = 8.0f;   // deferred-rescale threshold (natural-log units of the score)

// hardware RNE f32 -> packed bf16 (same rounding as f2bf on finite values).  Through the compiler's own conversion (it selects
// v_cvt_pk_bf16_f32 on gfx950), NOT inline assembly: the results feed MFMA operands, and the hazard recognizer does not look inside an
// asm block -- in the split + dense-mask variant an asm-written P fragment was consumed by the next-but-one MFMA and the products
// came out at bf16 accuracy (round 5: tools/diag_split_attn.py, 4e-3 instead of 3e-5 against the fp64 SDPA).
template <bool F16 = false>
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) { return Op16<F16>::pack2_bounded(lo, hi); }  // P <= e^AT_DEFER, O = a convex combination of V

// WPB = waves (32-row query tiles) per block: 4, or 5 when that saves a block per (batch, head) -- at Lq = 258 (the 18-step t2i
// loop: <soi> + 256 image tokens + <eoi>) the ninth query tile would otherwise get a block of its own that streams every K / V^T
// tile for 2 rows.  Waves 0..3 stage the tiles; a fifth wave only consumes them.
// SPLIT (accuracy mode, showo_attn_fwd_split): Q, K, V^T arrive as (hi, lo) bf16 pairs; S = Khi Qhi + Khi Qlo + Klo Qhi and
// O = Vhi Phi + Vhi Plo + Vlo Phi with P split in registers (P - bf16(P) rounded to bf16): three MFMAs per fragment pair, fp32
// softmax as before -> the fp32 SDPA of models/phi.py:715-722 to ~1e-5.  LDS holds four tiles per stage (K, V^T, Klo, V^Tlo).
// F16 (precision 2): Q, K, V^T, P and O are IEEE half (common.h Op16): v_mfma_f32_32x32x16_f16.  P <= e^AT_DEFER = 2981 stays far below
// 65504, and probabilities below the fp16 normal range (6.1e-5 of the running maximum's scale) keep their subnormal encoding.
template <bool DENSE, int WPB, bool SPLIT = false, bool F16 = false>
__device__ __forceinline__ void attn_lds_body(const AttnArgs& a, bf16_t* sm, int* s_hull) {
    static_assert(!(SPLIT && F16), "the (hi, lo) form is a bfloat16 scheme");
    constexpr int NT = SPLIT ? 4 : 2;  // tiles per stage
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int qb_, head, b;
    attn_block_coords(a, qb_, head, b);
    const int qblk = qb_ * WPB + wave;
    const bool wactive = qblk * 32 < a.Lq;
    const int qi = lane & 31, hh = lane >> 5;
    const int qrow_raw = qblk * 32 + qi;
    const int qrow = qrow_raw < a.Lq ? qrow_raw : a.Lq - 1;
    const int64_t bh = (int64_t)b * a.nH + head;

    bf16x8 qf[4], qfl[SPLIT ? 4 : 1];
    {
        const bf16_t* Qp = a.Q + (bh * a.Lq + qrow) * 64 + 8 * hh;
#pragma unroll
        for (int m = 0; m < 4; ++m) qf[m] = *reinterpret_cast<const bf16x8*>(Qp + 16 * m);
        if constexpr (SPLIT) {
            const bf16_t* Ql = a.Qlo + (bh * a.Lq + qrow) * 64 + 8 * hh;
#pragma unroll
            for (int m = 0; m < 4; ++m) qfl[m] = *reinterpret_cast<const bf16x8*>(Ql + 16 * m);
        }
    }
    constexpr bool dense = DENSE;
    int lo1, hi1, lo2, hi2;
    if (dense) {
        lo1 = 0; hi1 = a.Lk; lo2 = 0; hi2 = 0;
    } else if (a.iv) {
        int4 v = *reinterpret_cast<const int4*>(a.iv + ((int64_t)b * a.Lq + qrow) * 4);
        lo1 = v.x; hi1 = v.y; lo2 = v.z; hi2 = v.w;
    } else {
        lo1 = 0; hi1 = qrow + 1 + (a.Lk - a.Lq); lo2 = 0; hi2 = 0;
    }
    hi1 = min(hi1, a.Lk);
    hi2 = min(hi2, a.Lk);
    if (!wactive) { lo1 = hi1 = lo2 = hi2 = 0; }
    // key hull of the wave and of the block
    const int wmin = wave_min_i(min(lo1 < hi1 ? lo1 : 0x7fffffff, lo2 < hi2 ? lo2 : 0x7fffffff));
    const int wmax = wave_max_i(max(lo1 < hi1 ? hi1 : 0, lo2 < hi2 ? hi2 : 0));
    if (lane == 0) { s_hull[wave] = wmin; s_hull[WPB + wave] = wmax; }
    __syncthreads();
    int bmin = s_hull[0], bmax = s_hull[WPB];
#pragma unroll
    for (int w = 1; w < WPB; ++w) { bmin = min(bmin, s_hull[w]); bmax = max(bmax, s_hull[WPB + w]); }
    const float* drow = dense ? a.dense + ((int64_t)b * a.Lq + qrow) * a.Lk : nullptr;

    // ---- staging by DMA (global_load_lds, 16 B/lane, no VGPR round trip).  One wave-instruction fills 8 rows x 128 B
    // (lane-linear LDS image, so the chunk swizzle is applied to the SOURCE address); wave w issues pieces w and w + 4
    // of both tiles.  LDS row i of the K tile holds key pi(i) = (i & ~15) + 4 * {0,2,1,3}[(i >> 2) & 3] + (i & 3).
    const int prow = lane >> 3;
    const bf16_t* Kg = a.K + bh * a.Lcap * 64;
    const bf16_t* Vg = a.Vt + bh * 64 * a.Lp;
    const bf16_t* Kgl = SPLIT ? a.Klo + bh * a.Lcap * 64 : nullptr;
    const bf16_t* Vgl = SPLIT ? a.Vtlo + bh * 64 * a.Lp : nullptr;
    int pik[2], kch[2];
    int64_t voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 8 * ((wave & 3) + 4 * i) + prow;
        const int f = (r >> 1) & 7;
        const int blk = (r >> 2) & 3;
        pik[i] = (r & ~15) + 4 * (((blk & 1) << 1) | (blk >> 1)) + (r & 3);
        kch[i] = ((lane & 7) ^ f) << 3;
        voff[i] = (int64_t)r * a.Lp + kch[i];
    }
#define AT_STAGE(KT, BUF)                                                                              \
    if (WPB == 4 || wave < 4) {                                                                        \
        bf16_t* sK_ = sm + (BUF) * NT * AT_TILE;                                                       \
        bf16_t* sV_ = sK_ + AT_TILE;                                                                   \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                \
            int key_ = (KT) + pik[i];                                                                  \
            key_ = key_ < a.Lk ? key_ : a.Lk - 1; /* clamped rows are masked out below */              \
            glds16_untracked(Kg + (int64_t)key_ * 64 + kch[i], lds_addr_of(sK_ + (wave + 4 * i) * 512)); \
            glds16_untracked(Vg + voff[i] + (KT), lds_addr_of(sV_ + (wave + 4 * i) * 512));            \
            if constexpr (SPLIT) {                                                                     \
                glds16_untracked(Kgl + (int64_t)key_ * 64 + kch[i], lds_addr_of(sK_ + 2 * AT_TILE + (wave + 4 * i) * 512)); \
                glds16_untracked(Vgl + voff[i] + (KT), lds_addr_of(sV_ + 2 * AT_TILE + (wave + 4 * i) * 512));             \
            }                                                                                          \
        }                                                                                              \
    }

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    const bool one_iv = __all(lo2 >= hi2);  // wave-uniform: every row of the wave has ONE visible interval

    const int fsw = (qi >> 1) & 7;  // swizzle of the fragment rows qi and 32 + qi (same (r >> 1) & 7)
    const int kt0 = bmin >= 0x7fffffff ? 0 : (bmin & ~63);
    if (kt0 < bmax) AT_STAGE(kt0, 0);
    // everything loaded so far is retired HERE, with the registers named: a tracked load still pending at loop entry would make the
    // compiler wait for it inside the loop -- with a vmcnt(0) that also drains the (untracked) prefetch DMAs of every iteration
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]), "+v"(lo1), "+v"(hi1), "+v"(lo2), "+v"(hi2)::"memory");
    if constexpr (SPLIT) asm volatile("" : "+v"(qfl[0]), "+v"(qfl[1]), "+v"(qfl[2]), "+v"(qfl[3]));
    __syncthreads();
    int buf = 0;
    for (int kt = kt0; kt < bmax; kt += 64, buf ^= 1) {
        const bool more = kt + 64 < bmax;
        if (more) AT_STAGE(kt + 64, buf ^ 1);  // lands under this tile's MFMAs (untracked DMA: no compiler wait in front of the reads)
        const bf16_t* sK = sm + buf * NT * AT_TILE;
        const bf16_t* sV = sK + AT_TILE;
#pragma unroll  // both sub-tiles in one block of code (round 6: 29.5 -> 27.8 us at the t2i shape, no spill left; `unroll 1` dated from the scalar soft-max section)
        for (int sub = 0; sub < 2; ++sub) {
            const int ks = kt + 32 * sub;
            if (ks >= wmax || ks + 32 <= wmin) continue;  // wave-uniform: nothing visible to this wave here
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (32 * sub + qi) * 64 + (((2 * m + hh) ^ fsw) << 3));
                s = Op16<F16>::mfma32(kf, qf[m], s);
                if constexpr (SPLIT) {
                    bf16x8 kl = *reinterpret_cast<const bf16x8*>(sK + 2 * AT_TILE + (32 * sub + qi) * 64 + (((2 * m + hh) ^ fsw) << 3));
                    s = Op16<F16>::mfma32(kf, qfl[m], s);
                    s = Op16<F16>::mfma32(kl, qf[m], s);
                }
            }
            // interior sub-tile: every row of the wave sees all 32 keys -> no per-element mask work
            const bool inner = !dense && __all(((lo1 <= ks) & (ks + 32 <= hi1)) | ((lo2 <= ks) & (ks + 32 <= hi2)));
            if (!inner) {  // masked in place (no second copy of the score tile)
                // a lane's 16 keys are kb + c_r, kb = ks + 8 hh, c_r = 16 (r >> 3) + (r & 7) (K rows are stored in pi order).
                // One-interval rows (the t2i masks) whose sub-tile is clipped on ONE side only need one compare of the constant c_r
                // against a per-lane bound instead of the two-interval range test (6 VALU per score): the tail sub-tile (key < hi1)
                // and the sub-tile that holds the first visible key (key >= lo1) -- 2 of the 10 sub-tiles of the t2i shape.
                const int kb = ks + 8 * hh;
                if (!dense && one_iv && __all(lo1 <= ks)) {
                    const int T = hi1 - kb;  // visible <=> c_r < T
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = (16 * (r >> 3) + (r & 7) < T) ? s[r] : -INFINITY;
                } else if (!dense && one_iv && __all(hi1 >= ks + 32)) {
                    const int T = lo1 - kb;  // visible <=> c_r >= T
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = (16 * (r >> 3) + (r & 7) >= T) ? s[r] : -INFINITY;
                } else {
                    const unsigned len1 = (unsigned)max(hi1 - lo1, 0), len2 = (unsigned)max(hi2 - lo2, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kb + 16 * (r >> 3) + (r & 7);
                        const bool vis = ((unsigned)(key - lo1) < len1) | ((unsigned)(key - lo2) < len2);
                        float x = s[r];
                        if (dense) x += (key < a.Lk) ? drow[key] : 0.f;
                        s[r] = vis ? x : -INFINITY;
                    }
                }
            }
            float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
            mx = fmaxf(mx, s[15]);
            {   // the row's other 16 keys sit 32 lanes away: one VALU swap instead of a trip through the LDS crossbar
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            // deferred rescale: the running max is only advanced (and O, l rescaled) when some row's tile max exceeds
            // it by more than AT_DEFER; otherwise P = exp(S - m_old) <= e^AT_DEFER, still exact in the final ratio
            if (__any(mx > m_run + AT_DEFER)) {
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f((m_run - ((m_new == -INFINITY) ? 0.f : m_new)) * LOG2E);
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            }
            const float mb = ((m_run == -INFINITY) ? 0.f : m_run) * LOG2E;
            float p[16];
            f32x2_t ps2 = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {  // two scores per v_pk_fma_f32 / v_pk_add_f32 (same fp32 operations per element)
                const f32x2_t t = __builtin_elementwise_fma((f32x2_t){s[r], s[r + 1]}, (f32x2_t){LOG2E, LOG2E}, (f32x2_t){-mb, -mb});
                p[r] = __builtin_amdgcn_exp2f(t[0]);
                p[r + 1] = __builtin_amdgcn_exp2f(t[1]);
                ps2 += (f32x2_t){p[r], p[r + 1]};
            }
            l_run += ps2[0] + ps2[1];
            bf16x8 pb0, pb1;
            uint4 u0, u1;
            {
                u0.x = cvt_pk_bf16<F16>(p[0], p[1]); u0.y = cvt_pk_bf16<F16>(p[2], p[3]); u0.z = cvt_pk_bf16<F16>(p[4], p[5]); u0.w = cvt_pk_bf16<F16>(p[6], p[7]);
                u1.x = cvt_pk_bf16<F16>(p[8], p[9]); u1.y = cvt_pk_bf16<F16>(p[10], p[11]); u1.z = cvt_pk_bf16<F16>(p[12], p[13]); u1.w = cvt_pk_bf16<F16>(p[14], p[15]);
                pb0 = __builtin_bit_cast(bf16x8, u0);
                pb1 = __builtin_bit_cast(bf16x8, u1);
            }
            bf16x8 pl0, pl1;
            if constexpr (SPLIT) {  // low halves of P: p - bf16(p), rounded to bf16 (the high halves are taken from the packed words)
                const uint32_t hw[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
                uint32_t lw[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    lw[i] = cvt_pk_bf16<F16>(p[2 * i] - __uint_as_float(hw[i] << 16), p[2 * i + 1] - __uint_as_float(hw[i] & 0xffff0000u));
                pl0 = __builtin_bit_cast(bf16x8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
                pl1 = __builtin_bit_cast(bf16x8, make_uint4(lw[4], lw[5], lw[6], lw[7]));
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int c = ((4 * sub + 2 * kk + hh) ^ fsw) << 3;
                bf16x8 vf0 = *reinterpret_cast<const bf16x8*>(sV + qi * 64 + c);
                bf16x8 vf1 = *reinterpret_cast<const bf16x8*>(sV + (32 + qi) * 64 + c);
                o0 = Op16<F16>::mfma32(vf0, kk ? pb1 : pb0, o0);
                o1 = Op16<F16>::mfma32(vf1, kk ? pb1 : pb0, o1);
                if constexpr (SPLIT) {
                    bf16x8 vl0 = *reinterpret_cast<const bf16x8*>(sV + 2 * AT_TILE + qi * 64 + c);
                    bf16x8 vl1 = *reinterpret_cast<const bf16x8*>(sV + 2 * AT_TILE + (32 + qi) * 64 + c);
                    o0 = Op16<F16>::mfma32(vf0, kk ? pl1 : pl0, o0);
                    o1 = Op16<F16>::mfma32(vf1, kk ? pl1 : pl0, o1);
                    o0 = Op16<F16>::mfma32(vl0, kk ? pb1 : pb0, o0);
                    o1 = Op16<F16>::mfma32(vl1, kk ? pb1 : pb0, o1);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (!wactive) return;
    float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    float inv = 1.0f / l_tot;
    if (a.lse && hh == 0 && qrow_raw < a.Lq) a.lse[bh * a.Lq + qrow_raw] = m_run + __logf(l_tot);
    if (qrow_raw < a.Lq) {
        bf16_t* op = a.O + ((int64_t)b * a.Lq + qrow_raw) * a.ldo + head * 64 + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w0, w1;
            w0.x = cvt_pk_bf16<F16>(o0[4 * g] * inv, o0[4 * g + 1] * inv);
            w0.y = cvt_pk_bf16<F16>(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            w1.x = cvt_pk_bf16<F16>(o1[4 * g] * inv, o1[4 * g + 1] * inv);
            w1.y = cvt_pk_bf16<F16>(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(op + 8 * g) = w0;
            *reinterpret_cast<uint2*>(op + 32 + 8 * g) = w1;
            if constexpr (SPLIT) {
                bf16_t* ol = a.Olo + ((int64_t)b * a.Lq + qrow_raw) * a.ldo + head * 64 + 4 * hh;
                uint2 l0, l1;
                l0.x = cvt_pk_bf16<F16>(o0[4 * g] * inv - bf2f((bf16_t)(w0.x & 0xffffu)), o0[4 * g + 1] * inv - bf2f((bf16_t)(w0.x >> 16)));
                l0.y = cvt_pk_bf16<F16>(o0[4 * g + 2] * inv - bf2f((bf16_t)(w0.y & 0xffffu)), o0[4 * g + 3] * inv - bf2f((bf16_t)(w0.y >> 16)));
                l1.x = cvt_pk_bf16<F16>(o1[4 * g] * inv - bf2f((bf16_t)(w1.x & 0xffffu)), o1[4 * g + 1] * inv - bf2f((bf16_t)(w1.x >> 16)));
                l1.y = cvt_pk_bf16<F16>(o1[4 * g + 2] * inv - bf2f((bf16_t)(w1.y & 0xffffu)), o1[4 * g + 3] * inv - bf2f((bf16_t)(w1.y >> 16)));
                *reinterpret_cast<uint2*>(ol + 8 * g) = l0;
                *reinterpret_cast<uint2*>(ol + 32 + 8 * g) = l1;
            }
        }
    }
}

// accuracy-mode launch of the LDS-tiled forward: 64 KiB of dynamic LDS (2 stages x 4 tiles)
template <int WPB = 4>
__global__ __launch_bounds__(64 * WPB, 2) void attn_fwd_lds_split_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char at_dyn[];
    bf16_t* sm = reinterpret_cast<bf16_t*>(at_dyn);
    __shared__ int s_hull[2 * WPB];
    const bool dense = (a.flag != nullptr) && (a.dense != nullptr) && (*a.flag != 0);  // block-uniform
    if (dense) attn_lds_body<true, WPB, true>(a, sm, s_hull);
    else attn_lds_body<false, WPB, true>(a, sm, s_hull);
}


// (Round 6 measured the two stray query rows of the 258-row t2i step (258 = 2 x 128 + 2) as VALU single-row blocks co-scheduled in the
// same launch -- v_dot2 scores, block soft-max, P V from the key-contiguous V^T rows -- instead of an MFMA block with 30 rows of padding:
// SLOWER, 33.4 vs 29.5 us (tools/attn_bench.py, profiles/r6_attention_ab.txt: the serial load -> score -> soft-max -> load chain of a
// stray block outlasts the MFMA stub), and rows then get different bits depending on which path serves them, which breaks the
// prefix-reuse == recompute property.  Not in the library.)
// (Round 5 measured a RESIDENT form -- the whole key range of a head staged into LDS up front, one 576-thread block per (batch, head),
// no barrier in the key loop; bit-identical to attn_lds_body -- and it was slower: 312 vs 346 TF/s at the t2i shape, 37.8 vs 38.4
// images/s on one box (profiles/r5m_attention_resident_ab.txt): 112 KiB of LDS leave one block per CU and the up-front stage is bound by
// what one CU can pull.  The kernel is in the git history, not in the library.)
template <int MINW, int WPB = 4, bool F16 = false>
__global__ __launch_bounds__(64 * WPB, MINW) void attn_fwd_lds_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t sm[4 * AT_TILE];  // [buf][K | Vt]
    __shared__ int s_hull[2 * WPB];
    const bool dense = (a.flag != nullptr) && (a.dense != nullptr) && (*a.flag != 0);  // block-uniform
    if (dense) attn_lds_body<true, WPB, false, F16>(a, sm, s_hull);
    else attn_lds_body<false, WPB, false, F16>(a, sm, s_hull);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// Single-query form (AR decode step, Lq = 1): one 1024-thread block per (head, batch).  The 32x32 MFMA tiles of the
// prefill kernels would run one query row through ~20 dependent key tiles on ONE wave (51 us per layer); here every
// thread scores one key (q . K[k], fp32 FMAs over the 128-B key row), the soft-max is a block reduction, and wave w then
// accumulates head dimensions 4w..4w+3: the key-contiguous V^T rows are read coalesced (lane -> 8 consecutive keys) and
// reduced across the wave.  FUSED additionally does qk_prep's work for the new token of its head first (q/k LayerNorm +
// partial RoPE, K row and V^T column appended to the cache) so a decode layer needs no separate prep launch; the new
// key / value are taken from LDS, never re-read from global memory inside the launch.
// ------------------------------------------------------------------------------------------------
struct DecPrep {
    const bf16_t* qkv;  // [3 * nH * 64] projection row of the new token (B = 1)
    const float *qw, *qb, *kw, *kb, *cosT, *sinT;
    int rot, pos;
    float eps;
};

template <bool FUSED, bool F16 = false>
static __device__ __forceinline__ void attn_decode_body(AttnArgs a, DecPrep f, const int head, const int b) {
    extern __shared__ float sp[];  // probabilities, zero padded to a multiple of 512 keys
    __shared__ float red[32];
    __shared__ float sq[64], sk[64], sv[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t bh = (int64_t)b * a.nH + head;
    // a.Lk from the host is exact for a direct launch and an UPPER BOUND under graph replay (pos_dev set: the engine passes the last
    // position the captured loop can reach); the cache loads below are predicated on it, so they do not wait for the pos_dev round trip.
    const int Lhint = a.Lk;
    // ONE memory round trip for everything that does not depend on the position, issued in this order:
    //   (1) FUSED: the new token's q / k / v element of this head and the q / k LayerNorm weight + bias of this lane -- first in program
    //       order, so that the wave can wait for them (vmcnt counts in order) while the cache loads behind them are still in flight;
    //   (2) the cache: the first 1024 key rows (one per thread) and the first 1024 keys of this wave's four V^T rows.
    // No load sits under a branch (a wait behind a conditionally issued load can only be vmcnt(0), which made the prologue four
    // SERIAL round trips: position -> [token element + all cache loads] -> LayerNorm parameters -> RoPE row): rows / columns at or
    // beyond the hint are clamped onto the last live one (same address for every such lane: no extra traffic).
    const bf16_t* Kb = a.K + bh * a.Lcap * 64;
    const int d0 = wave * 4;
    const bf16_t* vr = a.Vt + (bh * 64 + d0) * a.Lp;
    // key rows are read coalesced: a wave instruction covers 8 whole rows (lane -> row lane>>3, 16-B chunk lane&7), wave w of
    // the block takes row group it*16 + w; the first 8 groups per wave (1024 keys) are requested before anything else.
    const int r8 = lane >> 3, ch = lane & 7;
    int pos = -1;  // FUSED: index of the key that lives in LDS
    if (FUSED) pos = a.pos_dev ? a.pos_dev[b] : f.pos;  // one position per sequence (b = 0 in the batch-1 launches); requested first
    // the mask operands too (absent ones read a valid dummy address instead of branching)
    const int flagv = *(a.flag ? a.flag : reinterpret_cast<const int32_t*>(a.K));
    const int4 ivrow = *reinterpret_cast<const int4*>(a.iv ? a.iv + (int64_t)b * 4 : reinterpret_cast<const int32_t*>(a.K));
    bf16_t xin = 0;
    float lnw = 0.f, lnb = 0.f;
    if (FUSED) {
        const int Hq = a.nH * 64;
        f.qkv += (int64_t)b * 3 * Hq;  // projection row of sequence b
        const int sel = wave < 2 ? wave : 2;  // wave 0: q, wave 1: k, the others: v (one cache line)
        xin = f.qkv[sel * Hq + head * 64 + lane];
        lnw = (sel == 0 ? f.qw : f.kw)[lane];
        lnb = (sel == 0 ? f.qb : f.kb)[lane];
    }
    uint4 ku[8], vu[2][4];
    const int klast = Lhint - 1, vlast = (Lhint - 1) & ~7;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int k = min((it * 16 + wave) * 8 + r8, klast);
        ku[it] = *reinterpret_cast<const uint4*>(Kb + (int64_t)k * 64 + ch * 8);
    }
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            vu[cc][i] = *reinterpret_cast<const uint4*>(vr + (int64_t)i * a.Lp + min(cc * 512 + 8 * lane, vlast));
    // rows / columns at or beyond the live length (and the row being appended by this launch, k == pos) may hold stale or
    // uninitialised bits: score() and accum() never let them reach a result
    if (FUSED) {
        a.Lk = pos + 1;
    } else if (a.pos_dev) {
        a.Lk = *a.pos_dev + 1;
    }
    float qv[8];  // this lane's 8 query dimensions (chunk ch)
    if (FUSED) {
        // the RoPE row of the new token: the only loads that depend on the position (every wave issues them: no branch before the wait)
        const int dl = lane < f.rot ? lane : 0;
        const float cs = f.cosT[(int64_t)pos * f.rot + dl], sn = f.sinT[(int64_t)pos * f.rot + dl];
        __builtin_amdgcn_sched_barrier(0);  // every load of the prologue is issued before the first wait (the scheduler otherwise starts the LayerNorm -- and its wait for the token element -- ahead of the RoPE loads)
        // ln_rope_lane's expressions with the statistics on the VALU-only reductions (same bits) and the RoPE operands in registers
        const float x0 = Op16<F16>::tof(xin);
        const float mean = wave_sum_swap(x0) * (1.0f / 64.0f);
        const float c = x0 - mean;
        const float var = wave_sum_swap(c * c) * (1.0f / 64.0f);
        float y = c * (1.0f / sqrtf(var + f.eps)) * lnw + lnb;
        const int half = f.rot >> 1;
        const float partner = __shfl_xor(y, half, 64);
        if (lane < f.rot) y = y * cs + (lane < half ? -partner : partner) * sn;
        if (wave == 0) {
            sq[lane] = Op16<F16>::tof(Op16<F16>::cvt(y * 0.125f));
        } else if (wave == 1) {
            const bf16_t kb = Op16<F16>::cvt(y);
            sk[lane] = Op16<F16>::tof(kb);
            const_cast<bf16_t*>(a.K)[(bh * a.Lcap + pos) * 64 + lane] = kb;
        } else if (wave == 2) {
            sv[lane] = Op16<F16>::tof(xin);
            const_cast<bf16_t*>(a.Vt)[(bh * 64 + lane) * a.Lp + pos] = xin;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] = sq[ch * 8 + j];
    } else {
        const uint4 u = *reinterpret_cast<const uint4*>(a.Q + bh * 64 + ch * 8);  // Lq = 1
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&u);
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] = Op16<F16>::tof(e[j]);
    }
    // the new token's key chunk / value dimensions in registers: the selects below stay selects (an LDS read under a per-element
    // condition compiles to a branch per element)
    float skv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, svv[4] = {0.f, 0.f, 0.f, 0.f};
    if (FUSED) {
#pragma unroll
        for (int j = 0; j < 8; ++j) skv[j] = sk[ch * 8 + j];
#pragma unroll
        for (int i = 0; i < 4; ++i) svv[i] = sv[d0 + i];
    }
    const bool dense = (a.flag != nullptr) && (a.dense != nullptr) && (flagv != 0);
    int lo1, hi1, lo2, hi2;
    if (dense) { lo1 = 0; hi1 = a.Lk; lo2 = 0; hi2 = 0; }
    else if (a.iv) {
        lo1 = ivrow.x; hi1 = ivrow.y; lo2 = ivrow.z; hi2 = ivrow.w;
    } else { lo1 = 0; hi1 = a.Lk; lo2 = 0; hi2 = 0; }  // causal: the newest token sees every key
    hi1 = min(hi1, a.Lk); hi2 = min(hi2, a.Lk);
    const float* drow = dense ? a.dense + (int64_t)b * a.Lk : nullptr;
    const int Lkp = (a.Lk + 511) & ~511;
    float mx = -INFINITY;
    // all 8 lanes of a row end with the row's score.  Branch-free up to the LDS store, so the eight unrolled calls interleave; the
    // 8-lane sum runs on DPP (partners 1, 2 = quad permutes; partner 4 = the half-row mirror, whose source lane sits in the other quad
    // and -- quads being uniform after the first two steps -- holds lane^4's value): __shfl_xor's values and order, no LDS round trip.
    auto score = [&](auto dense_c, int k, const uint4& kr) {  // dense_c: the (block-uniform) dense-mask case as a compile-time constant
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&kr);
        float sc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sc = fmaf(qv[j], (FUSED && k == pos) ? skv[j] : Op16<F16>::tof(e[j]), sc);
        sc += dpp_move<DPP_XOR1>(sc);
        sc += dpp_move<DPP_XOR2>(sc);
        sc += dpp_move<DPP_HALF_MIRROR>(sc);
        const bool live = k < a.Lk;
        const bool vis = ((k >= lo1) & (k < hi1)) | ((k >= lo2) & (k < hi2));
        if (decltype(dense_c)::value && live) sc += drow[k];
        sc = (live && vis) ? sc : -INFINITY;
        if (live && ch == 0) sp[k] = sc;
        mx = fmaxf(mx, sc);
    };
    if (dense) {
#pragma unroll
        for (int it = 0; it < 8; ++it) score(std::true_type{}, (it * 16 + wave) * 8 + r8, ku[it]);
    } else {
#pragma unroll
        for (int it = 0; it < 8; ++it) score(std::false_type{}, (it * 16 + wave) * 8 + r8, ku[it]);
    }
    for (int g = 128 + wave; g * 8 < a.Lk; g += 16) {
        const int k = g * 8 + r8;
        const uint4 kr = (k < a.Lk && k != pos) ? *reinterpret_cast<const uint4*>(Kb + (int64_t)k * 64 + ch * 8) : make_uint4(0, 0, 0, 0);
        if (dense) score(std::true_type{}, k, kr);
        else score(std::false_type{}, k, kr);
    }
    mx = wave_max_swap(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
    const float mu = (mx == -INFINITY) ? 0.f : mx;
    float sum = 0.f;
    for (int k = tid; k < Lkp; k += 1024) {
        float p = 0.f;
        if (k < a.Lk) {
            p = __builtin_amdgcn_exp2f((sp[k] - mu) * LOG2E);
            sum += p;
            p = Op16<F16>::tof(Op16<F16>::cvt(p));  // P is rounded to the operand type like the MFMA paths before it multiplies V
        }
        sp[k] = p;
    }
    sum = wave_sum_swap(sum);
    if (lane == 0) red[16 + wave] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[16 + w];
    const float inv = 1.0f / tot;
    // o[d] = sum_k p[k] V^T[d][k]
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    auto accum = [&](int c, const uint4 (&u)[4]) {
        const int kk = c + 8 * lane;
        const float4 p0 = *reinterpret_cast<const float4*>(sp + kk);
        const float4 p1 = *reinterpret_cast<const float4*>(sp + kk + 4);
        const float pr[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        const int nvalid = a.Lk - kk;  // cache slots beyond Lk hold stale or uninitialised bits: never multiply them
        const int jn = FUSED ? pos - kk : -1;  // the new token's value comes from LDS (same summation slot as the cached form)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t* e = reinterpret_cast<const bf16_t*>(&u[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[i] = fmaf(pr[j], j == jn ? svv[i] : (j < nvalid ? Op16<F16>::tof(e[j]) : 0.f), o[i]);
        }
    };
    accum(0, vu[0]);
    if (Lkp > 512) accum(512, vu[1]);
    for (int c = 1024; c < Lkp; c += 512) {
        const int kk = c + 8 * lane;
        uint4 u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = kk < a.Lk ? *reinterpret_cast<const uint4*>(vr + (int64_t)i * a.Lp + kk) : make_uint4(0, 0, 0, 0);
        accum(c, u);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = wave_sum_swap(o[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a.O[(int64_t)b * a.ldo + head * 64 + d0 + i] = Op16<F16>::cvt(o[i] * inv);
    }
}

template <bool FUSED, bool F16 = false>
__global__ __launch_bounds__(1024) void attn_decode_kernel(AttnArgs a, DecPrep f) {
    attn_decode_body<FUSED, F16>(a, f, blockIdx.x, blockIdx.y);
}

// Co-scheduled decode launch: blocks [0, nH) are the fused single-query attention (32 latency-bound blocks on a 256-CU chip); the
// remaining blocks stream the fc2 weights (33.5 MB at Phi-1.5's shape) and write y2 = fc2(gelu(fc1)) + b2, which does not depend on
// the attention (Phi's block is parallel-residual, models/phi.py:806-835).  No synchronisation between the roles: the following
// out_gemv2_kernel<1, 2> launch adds dense(attn) and y2 into the residual row.  One launch instead of a stream fork / join.
template <bool F16 = false>
__global__ __launch_bounds__(1024) void attn_decode_co_kernel(AttnArgs a, DecPrep f, showo::OutGemvArgs g, showo::DecodePrefetch pf) {
    if ((int)blockIdx.x >= a.nH) {
        extern __shared__ float sp[];
        const int nrole = gridDim.x - a.nH - pf.blocks;
        if ((int)blockIdx.x >= a.nH + nrole) showo::prefetch_role(pf, blockIdx.x - a.nH - nrole, pf.blocks, sp);
        else showo::fc2_columns_role<4, F16>(g, blockIdx.x - a.nH, nrole, 16, reinterpret_cast<bf16_t*>(sp));
        return;
    }
    attn_decode_body<true, F16>(a, f, blockIdx.x, 0);
}

// Batched form of the co-scheduled launch (decode_batch.hip): blocks [0, nH * B) = the (sequence, head) attention blocks, the rest
// stream the fc2 weights ONCE for all NB sequences (fc2_columns_roleB: the NB activation rows as bf16 in LDS).
template <int NB, bool F16 = false>
__global__ __launch_bounds__(1024) void attn_decode_coB_kernel(AttnArgs a, DecPrep f, showo::OutGemvBArgs g, showo::DecodePrefetch pf) {
    const int nab = a.nH * a.B;
    if ((int)blockIdx.x >= nab) {
        extern __shared__ float sp[];
        const int nrole = gridDim.x - nab - pf.blocks;
        if ((int)blockIdx.x >= nab + nrole) showo::prefetch_role(pf, blockIdx.x - nab - nrole, pf.blocks, sp);
        else showo::fc2_columns_roleB<4, NB, F16>(g, blockIdx.x - nab, nrole, 16, reinterpret_cast<bf16_t*>(sp));
        return;
    }
    attn_decode_body<true, F16>(a, f, blockIdx.x % a.nH, blockIdx.x / a.nH);
}

// decode-step graph replay (engine-internal): when set, single-token qk_prep / attention launches take the position from
// device memory, so the captured launch sequence is the same for every token
static const int* g_decode_pos_dev = nullptr;
static int g_decode_lk_max = 0;  // upper bound of the key count over the replays of the captured loop (0: cache capacity)
namespace showo { void attn_set_decode_pos(const int* p, int lk_max) { g_decode_pos_dev = p; g_decode_lk_max = p ? lk_max : 0; } }

extern "C" int showo_qk_prep_op16(const uint16_t* qkv, const float* qln_w, const float* qln_b, const float* kln_w,
                                  const float* kln_b, const float* cos_tab, const float* sin_tab, uint16_t* Q, uint16_t* K,
                                  uint16_t* Vt, int B, int L, int nH, int rot, float eps, int pos0, int Lcap, int Lp, int op,
                                  void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if (op != SHOWO_OP_BF16 && op != SHOWO_OP_F16) return set_error_msg(1, "qk_prep: op must be SHOWO_OP_BF16 or SHOWO_OP_F16");
    if (qln_w && rot != 32 && rot != 16 && rot != 64 && rot != 8) return set_error_msg(1, "qk_prep: rotary dim must be a power of two <= 64");
    if ((Lp % 64) || Lp < pos0 + L || Lcap < pos0 + L) return set_error_msg(1, "qk_prep: bad Lp/Lcap");
    PrepArgs a;
    a.qkv = qkv; a.qw = qln_w; a.qb = qln_b; a.kw = kln_w; a.kb = kln_b; a.cosT = cos_tab; a.sinT = sin_tab;
    a.Q = Q; a.K = K; a.Vt = Vt; a.B = B; a.L = L; a.nH = nH; a.rot = rot; a.pos0 = pos0; a.Lcap = Lcap; a.Lp = Lp;
    a.eps = eps;
    a.pos_dev = (L == 1 && B == 1) ? g_decode_pos_dev : nullptr;
    if (op) qk_prep_kernel<true><<<dim3((L + 63) / 64, nH, B), dim3(256), 0, (hipStream_t)stream>>>(a);
    else qk_prep_kernel<false><<<dim3((L + 63) / 64, nH, B), dim3(256), 0, (hipStream_t)stream>>>(a);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
extern "C" int showo_qk_prep(const uint16_t* qkv, const float* qln_w, const float* qln_b, const float* kln_w,
                             const float* kln_b, const float* cos_tab, const float* sin_tab, uint16_t* Q, uint16_t* K,
                             uint16_t* Vt, int B, int L, int nH, int rot, float eps, int pos0, int Lcap, int Lp,
                             void* stream) {
    return showo_qk_prep_op16(qkv, qln_w, qln_b, kln_w, kln_b, cos_tab, sin_tab, Q, K, Vt, B, L, nH, rot, eps, pos0, Lcap, Lp, SHOWO_OP_BF16, stream);
}

extern "C" int showo_mask_compress(const float* mask, int32_t* iv, int32_t* flag, int B, int Lq, int Lk, void* stream) {
    int rows = B * Lq;
    if (rows <= 0) return 0;
    SHOWO_CHECK_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), (hipStream_t)stream));
    mask_compress_kernel<<<dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>(mask, iv, flag, rows, Lk);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
static int g_attn_forced = -1;

extern "C" int showo_attn_set_impl(int impl) {
    g_attn_forced = (impl >= 1 && impl <= 2) ? impl : 0;  // 1 gather, 2 LDS-tiled
    return 0;
}

// XCD-aware 1-D grid of the LDS-tiled form (SHOWO_ATTN_XCD=0: the natural 3-D grid)
static dim3 grid_f16(AttnArgs& a, int nqb, int nH, int B) {
    static int xcd = -1;
    if (xcd < 0) { const char* e = getenv("SHOWO_ATTN_XCD"); xcd = e ? (atoi(e) != 0) : 1; }
    if (xcd) { a.nqb = nqb; return dim3((unsigned)nqb * nH * B); }
    a.nqb = 0;
    return dim3(nqb, nH, B);
}

static int attn_fwd_impl(const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, const int32_t* iv, const int32_t* flag,
                         const float* dense_mask, uint16_t* O, float* lse, int B, int nH, int Lq, int Lk, int Lcap, int Lp, int ldo,
                         void* stream, int op = 0) {
    if (B <= 0 || Lq <= 0 || Lk <= 0) return 0;
    if (op != SHOWO_OP_BF16 && op != SHOWO_OP_F16) return set_error_msg(1, "attn: op must be SHOWO_OP_BF16 or SHOWO_OP_F16");
    if (op && lse) return set_error_msg(1, "attn: the training forward (lse) has bf16 operands only");
    if ((Lp % 64) || Lp < Lk || Lcap < Lk || (ldo % 4)) return set_error_msg(1, "attn: bad Lp/Lcap/ldo");
    AttnArgs a;
    a.Q = Q; a.K = K; a.Vt = Vt; a.iv = iv; a.flag = flag; a.dense = dense_mask; a.O = O;
    a.B = B; a.nH = nH; a.Lq = Lq; a.Lk = Lk; a.Lcap = Lcap; a.Lp = Lp; a.ldo = ldo; a.lse = lse;
    a.pos_dev = (Lq == 1) ? g_decode_pos_dev : nullptr;
    int qblocks = (Lq + 31) / 32;
    ProfScope prof(PROF_ATTN, 4.0 * B * nH * (double)Lq * Lk * 64, (hipStream_t)stream);  // dense QK^T + PV flops
    if (g_attn_forced < 0) { const char* e = getenv("SHOWO_ATTN_IMPL"); g_attn_forced = e ? atoi(e) : 0; }
    const int forced = g_attn_forced;  // 1 = gather form, 2 = LDS-tiled form, else by shape
    const bool tiled = lse != nullptr || forced >= 2 || (forced != 1 && Lq >= 64);  // only the tiled form writes lse  // decode steps (a few query rows) keep the gather form
    if (Lq == 1 && forced != 1 && !lse) {  // AR decode step
        a.pos_dev = g_decode_pos_dev;
        if (g_decode_pos_dev) a.Lk = (g_decode_lk_max > 0 && g_decode_lk_max < Lcap) ? g_decode_lk_max : Lcap;  // load bound, see the kernel
        const size_t smem = (size_t)(((g_decode_pos_dev ? Lcap : Lk) + 511) & ~511) * sizeof(float);  // graph replay: Lk grows
        if (smem <= 60000) {
            if (op) attn_decode_kernel<false, true><<<dim3(nH, B), dim3(1024), smem, (hipStream_t)stream>>>(a, DecPrep{});
            else attn_decode_kernel<false><<<dim3(nH, B), dim3(1024), smem, (hipStream_t)stream>>>(a, DecPrep{});
            SHOWO_CHECK_HIP(hipGetLastError());
            return 0;
        }
    }
    if (op) {  // IEEE-half operands (precision 2): the default tile shapes only (the A/B variants stay bf16)
        if (tiled) { const dim3 g = grid_f16(a, (qblocks + 3) / 4, nH, B); attn_fwd_lds_kernel<4, 4, true><<<g, dim3(256), 0, (hipStream_t)stream>>>(a); }
        else attn_fwd_kernel<true><<<dim3((qblocks + 3) / 4, nH, B), dim3(256), 0, (hipStream_t)stream>>>(a);
        SHOWO_CHECK_HIP(hipGetLastError());
        return 0;
    }
    // SHOWO_ATTN_XCD (default 1): XCD-aware 1-D block order (attn_block_coords); 0 = the natural 3-D grid (A/B runs).
    static int xcd = -1;
    if (xcd < 0) { const char* e = getenv("SHOWO_ATTN_XCD"); xcd = e ? (atoi(e) != 0) : 1; }
    auto grid = [&](int nqb) { if (xcd) { a.nqb = nqb; return dim3((unsigned)nqb * nH * B); } a.nqb = 0; return dim3(nqb, nH, B); };
    if (tiled) { const dim3 g = grid((qblocks + 3) / 4); attn_fwd_lds_kernel<4><<<g, dim3(256), 0, (hipStream_t)stream>>>(a); }
    else attn_fwd_kernel<false><<<dim3((qblocks + 3) / 4, nH, B), dim3(256), 0, (hipStream_t)stream>>>(a);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int showo_attn_fwd(const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, const int32_t* iv,
                              const int32_t* flag, const float* dense_mask, uint16_t* O, int B, int nH, int Lq, int Lk,
                              int Lcap, int Lp, int ldo, void* stream) {
    return attn_fwd_impl(Q, K, Vt, iv, flag, dense_mask, O, nullptr, B, nH, Lq, Lk, Lcap, Lp, ldo, stream);
}
extern "C" int showo_attn_fwd_op16(const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, const int32_t* iv,
                                   const int32_t* flag, const float* dense_mask, uint16_t* O, int B, int nH, int Lq, int Lk,
                                   int Lcap, int Lp, int ldo, int op, void* stream) {
    return attn_fwd_impl(Q, K, Vt, iv, flag, dense_mask, O, nullptr, B, nH, Lq, Lk, Lcap, Lp, ldo, stream, op);
}

// Accuracy mode (showo_engine_set_precision 1): every operand and the output as (hi, lo) bf16 pairs of identical layout; three MFMAs
// per fragment pair, fp32 soft-max: models/phi.py:715-722 in fp32 to ~1e-5.  Any Lq >= 1 (a decode step runs one query tile).
extern "C" int showo_attn_fwd_split(const uint16_t* Q, const uint16_t* Qlo, const uint16_t* K, const uint16_t* Klo, const uint16_t* Vt,
                                    const uint16_t* Vtlo, const int32_t* iv, const int32_t* flag, const float* dense_mask, uint16_t* O,
                                    uint16_t* Olo, int B, int nH, int Lq, int Lk, int Lcap, int Lp, int ldo, void* stream) {
    if (B <= 0 || Lq <= 0 || Lk <= 0) return 0;
    if (!Qlo || !Klo || !Vtlo || !Olo) return set_error_msg(1, "attn_fwd_split: the low halves are required");
    if ((Lp % 64) || Lp < Lk || Lcap < Lk || (ldo % 4)) return set_error_msg(1, "attn: bad Lp/Lcap/ldo");
    AttnArgs a;
    a.Q = Q; a.K = K; a.Vt = Vt; a.iv = iv; a.flag = flag; a.dense = dense_mask; a.O = O;
    a.B = B; a.nH = nH; a.Lq = Lq; a.Lk = Lk; a.Lcap = Lcap; a.Lp = Lp; a.ldo = ldo; a.lse = nullptr; a.pos_dev = nullptr;
    a.Qlo = Qlo; a.Klo = Klo; a.Vtlo = Vtlo; a.Olo = Olo;
    const int qblocks = (Lq + 31) / 32, nqb = (qblocks + 3) / 4;
    ProfScope prof(PROF_ATTN, 3.0 * 4.0 * B * nH * (double)Lq * Lk * 64, (hipStream_t)stream);  // executed flops: three products
    static bool attr_set = false;
    constexpr int SMEM = 2 * 4 * AT_TILE * 2;
    if (!attr_set) {
        SHOWO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_lds_split_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set = true;
    }
    a.nqb = nqb;
    attn_fwd_lds_split_kernel<4><<<dim3((unsigned)nqb * nH * B), dim3(256), SMEM, (hipStream_t)stream>>>(a);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

// training forward: additionally writes lse fp32 [B, nH, Lq] (natural-log sum of exp of the masked score row)
extern "C" int showo_attn_fwd_lse(const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, const int32_t* iv,
                                  const int32_t* flag, const float* dense_mask, uint16_t* O, float* lse, int B, int nH, int Lq,
                                  int Lk, int Lcap, int Lp, int ldo, void* stream) {
    if (!lse) return set_error_msg(1, "attn_fwd_lse: lse is required");
    return attn_fwd_impl(Q, K, Vt, iv, flag, dense_mask, O, lse, B, nH, Lq, Lk, Lcap, Lp, ldo, stream);
}

// Engine-internal: decode-layer attention with the prep of the new token fused in (B = 1).  qkv = the new token's
// projection row [3 * nH * 64]; K / Vt = this layer's cache (appended at pos); iv int32[4] = the token's mask row.
namespace showo {
// Batched decode step (decode_batch.hip): B sequences, caches [B][nH][Lcap][64] / [B][nH][64][Lp], one position per sequence in
// pos_dev[B] (device), qkv [B, 3 nH 64], iv int32 [B, 4], O [B, nH 64].  lk_max: upper bound of every sequence's key count over the
// replays of the captured loop (load predicate, see attn_decode_body).  Same kernel and arithmetic as the batch-1 launch per sequence.
int attn_decode_fused_batch(const bf16_t* qkv, const float* qw, const float* qb, const float* kw, const float* kb, const float* cosT,
                            const float* sinT, bf16_t* K, bf16_t* Vt, const int32_t* iv, bf16_t* O, int B, int nH, int rot, float eps,
                            const int* pos_dev, int lk_max, int Lcap, int Lp, hipStream_t s, int op) {
    if ((Lp % 64) || !pos_dev || B < 1) return set_error_msg(1, "batched decode attention: bad arguments");
    AttnArgs a;
    a.Q = nullptr; a.K = K; a.Vt = Vt; a.iv = iv; a.flag = nullptr; a.dense = nullptr; a.O = O;
    a.B = B; a.nH = nH; a.Lq = 1; a.Lcap = Lcap; a.Lp = Lp; a.ldo = nH * 64; a.lse = nullptr;
    a.pos_dev = pos_dev;
    a.Lk = (lk_max > 0 && lk_max < Lcap) ? lk_max : Lcap;
    DecPrep f{qkv, qw, qb, kw, kb, cosT, sinT, rot, 0, eps};
    const size_t smem = (size_t)((Lcap + 511) & ~511) * sizeof(float);
    if (smem > 60000) return set_error_msg(5, "decode attention: cache longer than the single-block kernel supports");
    if (op) attn_decode_kernel<true, true><<<dim3(nH, B), dim3(1024), smem, s>>>(a, f);
    else attn_decode_kernel<true><<<dim3(nH, B), dim3(1024), smem, s>>>(a, f);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
int attn_decode_co_batch(const bf16_t* qkv, const float* qw, const float* qb, const float* kw, const float* kb, const float* cosT,
                         const float* sinT, bf16_t* K, bf16_t* Vt, const int32_t* iv, bf16_t* O, int B, int nH, int rot, float eps,
                         const int* pos_dev, int lk_max, int Lcap, int Lp, const OutGemvBArgs& fc2, int co_blocks, hipStream_t s,
                         const DecodePrefetch* pfp, int op) {
    DecodePrefetch pf{};
    if (pfp) pf = *pfp;
    if ((Lp % 64) || !pos_dev || B < 2 || B > 4 || fc2.K1 != 8192 || !fc2.y2 || co_blocks < 1)
        return set_error_msg(1, "batched co-scheduled decode attention: 2..4 sequences, fc2 with K = 8192 and y2 required");
    AttnArgs a;
    a.Q = nullptr; a.K = K; a.Vt = Vt; a.iv = iv; a.flag = nullptr; a.dense = nullptr; a.O = O;
    a.B = B; a.nH = nH; a.Lq = 1; a.Lcap = Lcap; a.Lp = Lp; a.ldo = nH * 64; a.lse = nullptr;
    a.pos_dev = pos_dev;
    a.Lk = (lk_max > 0 && lk_max < Lcap) ? lk_max : Lcap;
    DecPrep f{qkv, qw, qb, kw, kb, cosT, sinT, rot, 0, eps};
    const size_t smem_a = (size_t)((Lcap + 511) & ~511) * sizeof(float);
    if (smem_a > 60000) return set_error_msg(5, "decode attention: cache longer than the single-block kernel supports");
    const size_t smem_f = (size_t)B * fc2.K1 * sizeof(bf16_t);
    const size_t smem = smem_a > smem_f ? smem_a : smem_f;
    const dim3 grid(nH * B + co_blocks + pf.blocks);
    static bool attr[20] = {};
    auto launch = [&](auto kfn) -> int {
        if (!attr[B + (op ? 10 : 0)]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            if (e != hipSuccess) return set_error_hip(e, "hipFuncSetAttribute(attn_decode_coB)", __FILE__, __LINE__);
            attr[B + (op ? 10 : 0)] = true;
        }
        kfn<<<grid, dim3(1024), smem, s>>>(a, f, fc2, pf);
        return 0;
    };
    const int rc = op ? (B == 2 ? launch(attn_decode_coB_kernel<2, true>) : B == 3 ? launch(attn_decode_coB_kernel<3, true>) : launch(attn_decode_coB_kernel<4, true>))
                      : (B == 2 ? launch(attn_decode_coB_kernel<2>) : B == 3 ? launch(attn_decode_coB_kernel<3>) : launch(attn_decode_coB_kernel<4>));
    if (rc) return rc;
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
int attn_decode_fused(const bf16_t* qkv, const float* qw, const float* qb, const float* kw, const float* kb, const float* cosT,
                      const float* sinT, bf16_t* K, bf16_t* Vt, const int32_t* iv, bf16_t* O, int nH, int rot, float eps, int pos,
                      int Lcap, int Lp, hipStream_t s, const bf16_t* W2, const bf16_t* ffn, const float* b2, int F, int Hout, float* y2,
                      int co_blocks, const DecodePrefetch* pfp, int op) {
    if ((Lp % 64) || Lp <= pos || Lcap <= pos) return set_error_msg(1, "decode attention: bad Lp/Lcap");
    DecodePrefetch pf{};
    if (pfp && W2) pf = *pfp;
    AttnArgs a;
    a.Q = nullptr; a.K = K; a.Vt = Vt; a.iv = iv; a.flag = nullptr; a.dense = nullptr; a.O = O;
    a.B = 1; a.nH = nH; a.Lq = 1; a.Lk = pos + 1; a.Lcap = Lcap; a.Lp = Lp; a.ldo = nH * 64; a.lse = nullptr;
    a.pos_dev = g_decode_pos_dev;
    if (g_decode_pos_dev) a.Lk = (g_decode_lk_max > 0 && g_decode_lk_max < Lcap) ? g_decode_lk_max : Lcap;  // load bound, see the kernel
    DecPrep f{qkv, qw, qb, kw, kb, cosT, sinT, rot, pos, eps};
    const size_t smem = (size_t)(((g_decode_pos_dev ? Lcap : pos + 1) + 511) & ~511) * sizeof(float);
    if (smem > 60000) return set_error_msg(5, "decode attention: cache longer than the single-block kernel supports");
    if (W2) {  // co-scheduled fc2 role (see attn_decode_co_kernel): needs K1 = 4 x 2048
        if (F != 8192 || !y2 || !ffn || !b2) return set_error_msg(1, "decode attention: co-scheduled fc2 needs F = 8192 and y2");
        showo::OutGemvArgs g{nullptr, nullptr, nullptr, nullptr, 0, W2, ffn, b2, F, Hout, y2};
        const size_t smem2 = smem > (size_t)F * sizeof(bf16_t) ? smem : (size_t)F * sizeof(bf16_t);
        if (op) attn_decode_co_kernel<true><<<dim3(nH + co_blocks + pf.blocks, 1), dim3(1024), smem2, s>>>(a, f, g, pf);
        else attn_decode_co_kernel<false><<<dim3(nH + co_blocks + pf.blocks, 1), dim3(1024), smem2, s>>>(a, f, g, pf);
    } else if (op) {
        attn_decode_kernel<true, true><<<dim3(nH, 1), dim3(1024), smem, s>>>(a, f);
    } else {
        attn_decode_kernel<true><<<dim3(nH, 1), dim3(1024), smem, s>>>(a, f);
    }
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
}  // namespace showo
