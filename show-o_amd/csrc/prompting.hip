// On-device attention-mask construction from token ids.
// Replaces (reference): create_attention_mask_predict_next (training/prompting_utils.py:466-511),
// create_attention_mask_for_mmu (:591-604), create_attention_mask_for_mmu_vit (:606-624) -- host-orchestrated torch ops with
// a Python loop over the batch that materialise a [B,1,L,L] fp32/int64 tensor (43 MB at cfg3) per batch.
// Two products from ONE visibility predicate per family:
//   * the per-row visibility intervals the fused attention kernels consume (no dense mask at all), and
//   * the dense additive mask with the reference's exact values (0 / float(iinfo(int64).min)) for API compatibility.
// Predicate of predict_next (derived from the reference code, quirks kept):
//   in_img[l]  = (#soi up to l) > (#eoi up to l)  or  ids[l] in {soi, eoi}
//   text row r : c <= r, and with rm_pad_in_image additionally not (r > last_pad and c <= last_pad)
//   image row r: every column, and with rm_pad_in_image no pad column for rows r >= index of the sample's first <soi>
#include "common.h"
#include "../../include/showo_hip.h"
#include <cfloat>

using namespace showo;

namespace {

constexpr int MAXL = 4096;

struct PnArgs {
    const int64_t* ids;
    int L;
    int64_t pad_id, soi_id, eoi_id;
    int rm_pad;
};

// per-sample state in LDS: in_img / is_pad bit arrays, last pad index, first soi index
struct PnState {
    unsigned char in_img[MAXL];
    unsigned char is_pad[MAXL];
    int last_pad, first_soi;
};

__device__ void pn_prepare(const PnArgs& a, int b, PnState& st) {
    // one block per sample row; sequential prefix over L (L <= 4096: a few microseconds, once per batch)
    const int64_t* s = a.ids + (int64_t)b * a.L;
    for (int l = threadIdx.x; l < a.L; l += blockDim.x) st.is_pad[l] = s[l] == a.pad_id;
    if (threadIdx.x == 0) {
        int cs = 0, ce = 0, lp = -1, fs = -1;
        for (int l = 0; l < a.L; ++l) {
            const int64_t t = s[l];
            const bool soi = t == a.soi_id, eoi = t == a.eoi_id;
            cs += soi; ce += eoi;
            st.in_img[l] = (cs > ce) || soi || eoi;
            if (t == a.pad_id) lp = l;
            if (soi && fs < 0) fs = l;
        }
        st.last_pad = lp;
        st.first_soi = fs < 0 ? a.L : fs;
    }
    __syncthreads();
}
__device__ __forceinline__ bool pn_visible(const PnArgs& a, const PnState& st, int r, int c) {
    if (!st.in_img[r]) {
        if (c > r) return false;
        if (a.rm_pad && st.last_pad >= 0 && r > st.last_pad && c <= st.last_pad) return false;
        return true;
    }
    if (a.rm_pad && r >= st.first_soi && st.is_pad[c]) return false;
    return true;
}

// run-length extraction of one row (wave-cooperative, 64 columns per ballot): up to two visible runs, else flag
template <class Vis>
__device__ void row_intervals(Vis vis, int Lk, int32_t* iv_row, int32_t* flag) {
    const int lane = threadIdx.x & 63;
    int runs = 0;
    int lo[2] = {0, 0}, hi[2] = {0, 0};
    bool open = false;
    for (int c0 = 0; c0 < Lk; c0 += 64) {
        const int c = c0 + lane;
        const unsigned long long bits = __ballot(c < Lk && vis(c));
        int pos = 0;
        while (pos < 64) {
            if (!open) {
                const unsigned long long rest = bits >> pos;
                if (rest == 0) break;
                pos += __ffsll((long long)rest) - 1;
                if (runs < 2) lo[runs] = c0 + pos;
                open = true;
            } else {
                const unsigned long long rest = (~bits) >> pos;
                if (rest == 0) { pos = 64; break; }
                pos += __ffsll((long long)rest) - 1;
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
        if (runs > 2) atomicOr(flag, 1);
        *reinterpret_cast<int4*>(iv_row) = make_int4(lo[0], runs > 0 ? hi[0] : 0, runs > 1 ? lo[1] : 0, runs > 1 ? hi[1] : 0);
    }
}

__global__ __launch_bounds__(256) void pn_intervals_kernel(PnArgs a, int32_t* __restrict__ iv, int32_t* __restrict__ flag) {
    __shared__ PnState st;
    const int b = blockIdx.x, wave = threadIdx.x >> 6;
    pn_prepare(a, b, st);
    // gridDim.y blocks share the rows of a sequence (one block per sequence left all but B CUs idle: 210 us at the stage-1 batch)
    for (int r = blockIdx.y * 4 + wave; r < a.L; r += 4 * gridDim.y)
        row_intervals([&](int c) { return pn_visible(a, st, r, c); }, a.L, iv + ((int64_t)b * a.L + r) * 4, flag);
}
__global__ __launch_bounds__(256) void pn_dense_kernel(PnArgs a, float* __restrict__ mask, float neg) {
    __shared__ PnState st;
    const int b = blockIdx.x;
    pn_prepare(a, b, st);
    float* mb = mask + (int64_t)b * a.L * a.L;
    for (int64_t i = threadIdx.x; i < (int64_t)a.L * a.L; i += blockDim.x) {
        const int r = (int)(i / a.L), c = (int)(i - (int64_t)r * a.L);
        mb[i] = pn_visible(a, st, r, c) ? 0.0f : neg;
    }
}

// mmu: causal, plus columns [0, e] for every row, e = first <eoi> of the FIRST sample (reference: eoi_image[0])
// mmu_vit: causal, plus columns [lo, hi) for every row
__global__ void band_intervals_kernel(const int64_t* __restrict__ ids, int L, int64_t eoi_id, int lo_fixed, int hi_fixed,
                                      int32_t* __restrict__ iv, int total) {
    __shared__ int e_s;
    if (threadIdx.x == 0) {
        int e = -1;
        if (ids) {
            for (int l = 0; l < L; ++l)
                if (ids[l] == eoi_id) { e = l; break; }
        }
        e_s = e;
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int r = i % L;
    int lo = ids ? 0 : lo_fixed, hi = ids ? e_s + 1 : hi_fixed;  // the always-visible band [lo, hi)
    int4 o;
    if (hi <= lo) o = make_int4(0, r + 1, 0, 0);                               // no band: causal
    else if (lo <= r + 1) o = make_int4(0, max(r + 1, hi), 0, 0);               // band touches the causal prefix: one run
    else o = make_int4(0, r + 1, lo, hi);                                       // causal prefix + separate band
    *reinterpret_cast<int4*>(iv + (int64_t)i * 4) = o;
}

__global__ void dense_from_intervals_kernel(const int32_t* __restrict__ iv, float* __restrict__ mask, int L, int64_t total, float neg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t row = i / L;
    const int c = (int)(i - row * L);
    const int4 v = *reinterpret_cast<const int4*>(iv + row * 4);
    mask[i] = ((c >= v.x && c < v.y) || (c >= v.z && c < v.w)) ? 0.0f : neg;
}


// ------------------------------------------------------------------------------------------------
// MLM corruption of the image tokens of a training batch (training/utils.py:77-154 mask_or_random_replace_tokens).
// The reference draws noise = rand(B, N), takes perm = argsort(noise) and masks position j iff perm[j] < n_b: position
// rank(i) is masked for every i < n_b, where rank(i) = #{j : noise[j] < noise[i]} (ties by index).  One block per row;
// only the first n_b elements need their rank, so the work is n_b * N compares out of LDS.
// rect != NULL selects the contiguous-region form instead (rows [y0,y1) x cols [x0,x1) of the res x res grid).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_tokens_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ noise,
                                                          const int32_t* __restrict__ num_masked, const int32_t* __restrict__ rect,
                                                          int res, int N, int64_t mask_id, int64_t ignore_id, int predict_all,
                                                          int64_t* __restrict__ input_ids, int64_t* __restrict__ labels,
                                                          uint8_t* __restrict__ mask) {
    extern __shared__ float sn[];                            // [N] noise row
    uint8_t* flags = reinterpret_cast<uint8_t*>(sn + N);     // [N]
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < N; j += 256) flags[j] = 0;
    if (rect) {
        __syncthreads();
        const int y0 = rect[b * 4 + 0], y1 = rect[b * 4 + 1], x0 = rect[b * 4 + 2], x1 = rect[b * 4 + 3];
        for (int j = tid; j < N; j += 256) {
            const int y = j / res, x = j - y * res;
            flags[j] = (y >= y0 && y < y1 && x >= x0 && x < x1) ? 1 : 0;
        }
    } else {
        for (int j = tid; j < N; j += 256) sn[j] = noise[(int64_t)b * N + j];
        __syncthreads();
        const int n = min(num_masked[b], N);
        for (int i = tid; i < n; i += 256) {
            const float v = sn[i];
            int rank = 0;
            for (int j = 0; j < N; ++j) {
                const float w = sn[j];
                rank += (w < v) | ((w == v) & (j < i));
            }
            flags[rank] = 1;
        }
    }
    __syncthreads();
    for (int j = tid; j < N; j += 256) {
        const int64_t t = tokens[(int64_t)b * N + j];
        const bool m = flags[j] != 0;
        input_ids[(int64_t)b * N + j] = m ? mask_id : t;
        labels[(int64_t)b * N + j] = (m || predict_all) ? t : ignore_id;
        if (mask) mask[(int64_t)b * N + j] = m ? 1 : 0;
    }
}

}  // namespace

static const float NEG_MASK = -9223372036854775808.0f;  // float(torch.iinfo(torch.int64).min), prompting_utils.py:505-509

extern "C" int showo_mask_predict_next(const int64_t* ids, int B, int L, int64_t pad_id, int64_t soi_id, int64_t eoi_id,
                                       int rm_pad_in_image, int32_t* iv, int32_t* flag, float* dense, void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if (L > MAXL) return set_error_msg(1, "mask_predict_next: L <= 4096 supported");
    if (!iv && !dense) return set_error_msg(1, "mask_predict_next: nothing to write");
    if (iv && !flag) return set_error_msg(1, "mask_predict_next: flag required with iv");
    hipStream_t s = (hipStream_t)stream;
    PnArgs a{ids, L, pad_id, soi_id, eoi_id, rm_pad_in_image};
    if (iv) {
        SHOWO_CHECK_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), s));
        pn_intervals_kernel<<<dim3(B, (unsigned)((L + 31) / 32 < 16 ? (L + 31) / 32 : 16)), dim3(256), 0, s>>>(a, iv, flag);
    }
    if (dense) pn_dense_kernel<<<dim3(B), dim3(256), 0, s>>>(a, dense, NEG_MASK);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int showo_mask_mmu(const int64_t* ids, int B, int L, int64_t eoi_id, int32_t* iv, float* dense, void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if (!ids || !iv) return set_error_msg(1, "mask_mmu: ids and iv are required");
    hipStream_t s = (hipStream_t)stream;
    const int total = B * L;
    band_intervals_kernel<<<dim3((total + 255) / 256), dim3(256), 0, s>>>(ids, L, eoi_id, 0, 0, iv, total);
    if (dense) {
        const int64_t n = (int64_t)total * L;
        dense_from_intervals_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(iv, dense, L, n, NEG_MASK);
    }
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int showo_mask_mmu_vit(int B, int L, int system_prompt_len, int num_image_tokens, int32_t* iv, float* dense, void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if (!iv) return set_error_msg(1, "mask_mmu_vit: iv is required");
    hipStream_t s = (hipStream_t)stream;
    const int total = B * L;
    const int lo = 1 + system_prompt_len + 1, hi = min(lo + num_image_tokens, L);
    band_intervals_kernel<<<dim3((total + 255) / 256), dim3(256), 0, s>>>(nullptr, L, 0, lo, hi, iv, total);
    if (dense) {
        const int64_t n = (int64_t)total * L;
        dense_from_intervals_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(iv, dense, L, n, NEG_MASK);
    }
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

// tokens int64 [B,N]; noise fp32 [B,N] and num_masked int32 [B] (random form) or rect int32 [B,4] = y0,y1,x0,x1 on the
// res x res grid (contiguous form; noise / num_masked unused); out: input_ids, labels int64 [B,N], mask uint8 [B,N] (optional)
extern "C" int showo_mask_tokens(const int64_t* tokens, const float* noise, const int32_t* num_masked, const int32_t* rect, int res,
                                 int B, int N, int64_t mask_id, int64_t ignore_id, int predict_all, int64_t* input_ids,
                                 int64_t* labels, uint8_t* mask, void* stream) {
    if (B <= 0 || N <= 0) return 0;
    if (!tokens || !input_ids || !labels) return set_error_msg(1, "mask_tokens: tokens, input_ids and labels are required");
    if (!rect && (!noise || !num_masked)) return set_error_msg(1, "mask_tokens: noise and num_masked (or rect) are required");
    if (rect && (res <= 0 || res * res != N)) return set_error_msg(1, "mask_tokens: contiguous form needs N = res * res");
    if (N > 12000) return set_error_msg(5, "mask_tokens: row longer than the LDS-resident kernel supports");
    const size_t smem = (size_t)N * sizeof(float) + (size_t)N;
    mask_tokens_kernel<<<dim3(B), dim3(256), smem, (hipStream_t)stream>>>(tokens, noise, num_masked, rect, res, N, mask_id, ignore_id,
                                                                         predict_all, input_ids, labels, mask);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
