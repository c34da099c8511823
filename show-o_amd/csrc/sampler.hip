// Mask-predict (MaskGIT) sampler kernels for gfx950.
// Replaces (reference): models/modeling_showo.py:140-179 (CFG combine, slice, softmax, multinomial,
// confidence gather, mask_len clamp, write-back) and models/sampling.py:10-16,31-36 (log, gumbel_noise,
// mask_by_random_topk).  Arithmetic order follows the reference expression by expression (explicit
// __fmul_rn/__fadd_rn so hipcc cannot contract them into FMAs); only the libm-level functions
// (exp/log) can differ in the last ulp from the CPU.
#include "common.h"
#include "../../include/showo_hip.h"
#include <cfloat>

using namespace showo;

namespace {

struct SampleArgs {
    const float *lc, *lu;
    int ld;
    float a1, a2;  // (1+w), w as fp32 scalars (python float -> fp32 tensor op)
    const int64_t* cur;
    int64_t mask_id;
    const float* exp_noise;
    uint64_t seed;
    uint32_t step;
    int64_t* sampled;
    float* sel;
    int N, V;
    // graph replay: the step index (and the seed: int32 pair at step_dev + 2) live in device memory -- the captured launch is
    // identical for every step and every call
    const int* step_dev;
    int64_t noise_stride;  // elements of injected noise per step
};

// one block per image token row; z staged in LDS (V*4 bytes, 32 KB at V=8192)
__global__ __launch_bounds__(256) void cfg_softmax_sample_kernel(SampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sz[];
    __shared__ float red_f[4];
    __shared__ int red_i[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.step_dev) {
        a.step = (uint32_t)*a.step_dev;
        a.seed = (uint64_t)(uint32_t)a.step_dev[2] | ((uint64_t)(uint32_t)a.step_dev[3] << 32);
        if (a.exp_noise) a.exp_noise += (int64_t)a.step * a.noise_stride;
    }
    const int64_t c = a.cur[row];
    if (c != a.mask_id) {  // known token: keeps its id, confidence = finfo.max (modeling_showo.py:154,164)
        if (tid == 0) { a.sampled[row] = c; a.sel[row] = FLT_MAX; }
        return;
    }
    const float* lc = a.lc + (int64_t)row * a.ld;
    const float* lu = a.lu ? a.lu + (int64_t)row * a.ld : nullptr;
    float mx = -INFINITY;
    for (int i = tid; i < a.V; i += 256) {
        float z = lu ? __fsub_rn(__fmul_rn(a.a1, lc[i]), __fmul_rn(a.a2, lu[i])) : lc[i];
        sz[i] = z;
        mx = fmaxf(mx, z);
    }
    mx = wave_max(mx);
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < a.V; i += 256) {
        float e = expf(sz[i] - mx);
        sz[i] = e;
        s += e;
    }
    s = wave_sum(s);
    if (lane == 0) red_f[wave] = s;
    __syncthreads();
    s = (red_f[0] + red_f[1]) + (red_f[2] + red_f[3]);
    __syncthreads();
    // argmax_i p_i / E_i  (what torch.multinomial(p, 1) computes: q.exponential_(1); argmax(p / q))
    Philox ph(a.seed);
    float best = -1.f;
    int bi = 0x7fffffff;
    for (int i0 = tid * 4; i0 < a.V; i0 += 1024) {
        float e4[4];
        if (a.exp_noise) {
#pragma unroll
            for (int j = 0; j < 4; ++j) e4[j] = (i0 + j < a.V) ? a.exp_noise[(int64_t)row * a.V + i0 + j] : 1.f;
        } else {
            uint32_t r4[4];
            ph.gen((uint32_t)(i0 >> 2), (uint32_t)row, a.step, 0x51u, r4);
#pragma unroll
            for (int j = 0; j < 4; ++j) e4[j] = -logf(u32_to_unit(r4[j]));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int i = i0 + j;
            if (i < a.V) {
                float p = __fdiv_rn(sz[i], s);
                float sc = __fdiv_rn(p, e4[j]);
                if (sc > best || (sc == best && i < bi)) { best = sc; bi = i; }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { red_f[wave] = best; red_i[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (red_f[w] > best || (red_f[w] == best && red_i[w] < bi)) { best = red_f[w]; bi = red_i[w]; }
        a.sampled[row] = bi;
        a.sel[row] = __fdiv_rn(sz[bi], s);
    }
}

struct TopkArgs {
    const float* sel;
    const int64_t* sampled;
    int64_t *cur, *ids_c, *ids_u;
    int ld_ids, img_start;
    int64_t mask_id, id_offset;
    float mask_len_f, temp;
    const float* uniform;
    uint64_t seed;
    uint32_t step;
    uint8_t* masking_out;
    int N;
    const int* step_dev;   // graph replay: step index, schedule constants sched[0..steps) = mask_len, [steps..2 steps) = temperature
    const float* sched;
    int steps;
    int64_t noise_stride;
};

// one block per sample; confidences in LDS; k-th smallest by rank counting (N <= 4096)
__global__ __launch_bounds__(256) void mask_by_topk_kernel(TopkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float conf[];
    __shared__ int cnt_unknown;
    __shared__ float cut_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (a.step_dev) {
        a.step = (uint32_t)*a.step_dev;
        a.seed = (uint64_t)(uint32_t)a.step_dev[2] | ((uint64_t)(uint32_t)a.step_dev[3] << 32);
        a.mask_len_f = a.sched[a.step];
        a.temp = a.sched[a.steps + a.step];
        if (a.uniform) a.uniform += (int64_t)a.step * a.noise_stride;
    }
    if (tid == 0) { cnt_unknown = 0; cut_s = INFINITY; }
    __syncthreads();
    Philox ph(a.seed);
    int local_unknown = 0;
    for (int i = tid; i < a.N; i += 256) {
        int64_t idx = (int64_t)b * a.N + i;
        if (a.cur[idx] == a.mask_id) local_unknown++;
        float u;
        if (a.uniform) {
            u = a.uniform[idx];
        } else {
            uint32_t r4[4];
            ph.gen((uint32_t)i, (uint32_t)b, a.step, 0x6bu, r4);
            u = u32_to_unit(r4[0]);
        }
        // gumbel_noise: -log(-log(u)) with log(t) = log(clamp(t, 1e-20)) (sampling.py:10-16)
        float g = -logf(fmaxf(-logf(fmaxf(u, 1e-20f)), 1e-20f));
        conf[i] = __fadd_rn(logf(fmaxf(a.sel[idx], 1e-20f)), __fmul_rn(a.temp, g));
    }
    if (local_unknown) atomicAdd(&cnt_unknown, local_unknown);
    __syncthreads();
    // mask_len = max(1, min(unknown - 1, floor(N * ratio)))  (modeling_showo.py:166-171), as floats then .long()
    float ml = fmaxf(1.0f, fminf((float)(cnt_unknown - 1), a.mask_len_f));
    int k = (int)ml;
    if (k > a.N - 1) k = a.N - 1;
    for (int i = tid; i < a.N; i += 256) {
        float x = conf[i];
        int less = 0, leq = 0;
        for (int j = 0; j < a.N; ++j) {
            float y = conf[j];
            less += (y < x);
            leq += (y <= x);
        }
        if (less <= k && k < leq) cut_s = x;  // same value from every writer
    }
    __syncthreads();
    const float cut = cut_s;
    for (int i = tid; i < a.N; i += 256) {
        int64_t idx = (int64_t)b * a.N + i;
        bool m = conf[i] < cut;
        int64_t sid = a.sampled[idx];
        int64_t tok = m ? a.mask_id : sid + a.id_offset;
        a.ids_c[(int64_t)b * a.ld_ids + a.img_start + i] = tok;
        if (a.ids_u) a.ids_u[(int64_t)b * a.ld_ids + a.img_start + i] = tok;
        a.cur[idx] = m ? a.mask_id : sid;
        if (a.masking_out) a.masking_out[idx] = m ? 1 : 0;
    }
}

__global__ void step_inc_kernel(int* step) {
    if (threadIdx.x == 0) *step += 1;
}

}  // namespace

// Device-side step mode (engine-internal, used while a denoise step is captured into a hipGraph): when set, both sampler
// kernels take the step index from *step_dev, the schedule constants from sched and offset injected noise themselves.
static const int* g_sampler_step_dev = nullptr;
static const float* g_sampler_sched = nullptr;
static int g_sampler_steps = 0;
namespace showo {
void sampler_set_device_step(const int* step_dev, const float* sched, int steps) {
    g_sampler_step_dev = step_dev; g_sampler_sched = sched; g_sampler_steps = steps;
}
int sampler_step_inc(int* step_dev, hipStream_t s) {
    step_inc_kernel<<<1, 64, 0, s>>>(step_dev);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
}  // namespace showo

extern "C" int showo_cfg_softmax_sample(const float* logits_c, const float* logits_u, int ld, float guidance,
                                        const int64_t* cur, int64_t mask_id, const float* exp_noise, uint64_t seed,
                                        uint32_t step, int64_t* sampled, float* sel_prob, int B, int N, int V,
                                        void* stream) {
    int rows = B * N;
    if (rows <= 0) return 0;
    if (V <= 0 || V > 40000) return set_error_msg(1, "sampler: V out of range (LDS staging supports V <= 40000)");
    SampleArgs a;
    a.lc = logits_c; a.lu = logits_u; a.ld = ld;
    a.a1 = (float)(1.0 + (double)guidance); a.a2 = guidance;
    a.cur = cur; a.mask_id = mask_id; a.exp_noise = exp_noise; a.seed = seed; a.step = step;
    a.step_dev = g_sampler_step_dev; a.noise_stride = (int64_t)rows * V;
    a.sampled = sampled; a.sel = sel_prob; a.N = N; a.V = V;
    size_t smem = (size_t)V * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        SHOWO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cfg_softmax_sample_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160000));
        attr_set = true;
    }
    cfg_softmax_sample_kernel<<<dim3(rows), dim3(256), smem, (hipStream_t)stream>>>(a);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int showo_mask_by_topk(const float* sel_prob, const int64_t* sampled, int64_t* cur, int64_t* ids_cond,
                                  int64_t* ids_uncond, int ld_ids, int img_start, int64_t mask_id, int64_t id_offset,
                                  float mask_len_f, float temperature, const float* uniform, uint64_t seed,
                                  uint32_t step, uint8_t* masking_out, int B, int N, void* stream) {
    if (B <= 0 || N <= 0) return 0;
    if (N > 4096) return set_error_msg(1, "mask_by_topk: N <= 4096 supported");
    TopkArgs a;
    a.sel = sel_prob; a.sampled = sampled; a.cur = cur; a.ids_c = ids_cond; a.ids_u = ids_uncond;
    a.ld_ids = ld_ids; a.img_start = img_start; a.mask_id = mask_id; a.id_offset = id_offset;
    a.mask_len_f = mask_len_f; a.temp = temperature; a.uniform = uniform; a.seed = seed; a.step = step;
    a.step_dev = g_sampler_step_dev; a.sched = g_sampler_sched; a.steps = g_sampler_steps; a.noise_stride = (int64_t)B * N;
    a.masking_out = masking_out; a.N = N;
    mask_by_topk_kernel<<<dim3(B), dim3(256), (size_t)N * sizeof(float), (hipStream_t)stream>>>(a);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Next-token sampling of the AR decode (modeling_showo.py:220-228): x = logits / temperature; keep x >= (top_k-th
// largest x) (`logits[logits < v[:, [-1]]] = -inf`: ties with the k-th value stay); p = softmax(x); token =
// multinomial(p, 1) = argmax_i p_i / E_i with E ~ Exp(1) (asserted against torch in oracle/make_golden.py).
// One 1024-thread block; the vocabulary row (58 498 fp32 = 234 KB) stays in L2 and is re-read per pass: four 8-bit
// radix-select passes over an order-preserving key find the exact k-th largest value, then max, sum and the arg-max.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct DecSampleArgs {
    const float* logits;
    int V, top_k;
    float temperature;
    const float* exp_noise;   // optional [*, V]: row `step` is used (parity tests); else Philox(seed; step)
    int64_t noise_stride;
    uint64_t seed;
    int step;                 // index of this draw within the generation ...
    const int* pos_dev;       // ... or step + (*pos_dev - pos_base) when the position lives on the device (graph replay)
    int pos_base;
    int64_t* tok;
};
__device__ inline uint32_t order_key(float x) {  // ascending float order -> ascending unsigned order
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ __launch_bounds__(1024) void sample_topk_kernel(DecSampleArgs a) {
    __shared__ unsigned hist[256];
    __shared__ unsigned sel[2];
    __shared__ float red_f[16];
    __shared__ int red_i[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int step = a.step + (a.pos_dev ? *a.pos_dev - a.pos_base : 0);
    const float T = a.temperature;
    uint32_t thr = 0;  // keep keys >= thr
    if (a.top_k > 0 && a.top_k < a.V) {
        uint32_t prefix = 0, mask = 0;
        unsigned k = (unsigned)a.top_k;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < a.V; i += 1024) {
                const uint32_t key = order_key(__fdiv_rn(a.logits[i], T));
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned cum = 0;
                int b = 255;
                for (; b > 0; --b) {
                    if (cum + hist[b] >= k) break;
                    cum += hist[b];
                }
                sel[0] = (unsigned)b;
                sel[1] = k - cum;  // rank of the wanted element inside bin b
            }
            __syncthreads();
            prefix |= sel[0] << shift;
            mask |= 255u << shift;
            k = sel[1];
            __syncthreads();
        }
        thr = prefix;
    }
    // max of the kept values
    float mx = -INFINITY;
    for (int i = tid; i < a.V; i += 1024) {
        const float x = __fdiv_rn(a.logits[i], T);
        if (order_key(x) >= thr) mx = fmaxf(mx, x);
    }
    mx = wave_max(mx);
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = red_f[0];
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red_f[w]);
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < a.V; i += 1024) {
        const float x = __fdiv_rn(a.logits[i], T);
        if (order_key(x) >= thr) sum += expf(x - mx);
    }
    sum = wave_sum(sum);
    if (lane == 0) red_f[wave] = sum;
    __syncthreads();
    sum = 0.f;
    for (int w = 0; w < 16; ++w) sum += red_f[w];
    __syncthreads();
    Philox ph(a.seed);
    const float* en = a.exp_noise ? a.exp_noise + (int64_t)step * a.noise_stride : nullptr;
    float best = -1.f;
    int bi = 0x7fffffff;
    for (int i0 = tid * 4; i0 < a.V; i0 += 4096) {
        float e4[4];
        if (en) {
#pragma unroll
            for (int j = 0; j < 4; ++j) e4[j] = (i0 + j < a.V) ? en[i0 + j] : 1.f;
        } else {
            uint32_t r4[4];
            ph.gen((uint32_t)(i0 >> 2), 0u, (uint32_t)step, 0x77u, r4);
#pragma unroll
            for (int j = 0; j < 4; ++j) e4[j] = -logf(u32_to_unit(r4[j]));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + j;
            if (i < a.V) {
                const float x = __fdiv_rn(a.logits[i], T);
                if (order_key(x) >= thr) {
                    const float sc = __fdiv_rn(__fdiv_rn(expf(x - mx), sum), e4[j]);
                    if (sc > best || (sc == best && i < bi)) { best = sc; bi = i; }
                }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { red_f[wave] = best; red_i[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (red_f[w] > best || (red_f[w] == best && red_i[w] < bi)) { best = red_f[w]; bi = red_i[w]; }
        a.tok[0] = bi;
    }
}
}  // namespace

namespace showo {
int sample_topk_launch(const float* logits, int V, int top_k, float temperature, const float* exp_noise, int64_t noise_stride,
                       uint64_t seed, int step, const int* pos_dev, int pos_base, int64_t* tok, hipStream_t s) {
    if (!logits || !tok || V <= 0) return set_error_msg(1, "sample_topk: bad arguments");
    if (!(temperature > 0.f)) return set_error_msg(1, "sample_topk: temperature must be > 0");
    DecSampleArgs a{logits, V, top_k, temperature, exp_noise, noise_stride, seed, step, pos_dev, pos_base, tok};
    sample_topk_kernel<<<dim3(1), dim3(1024), 0, s>>>(a);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
}  // namespace showo

extern "C" int showo_sample_topk(const float* logits, int V, int top_k, float temperature, const float* exp_noise, uint64_t seed,
                                 int step, int64_t* tok, void* stream) {
    return showo::sample_topk_launch(logits, V, top_k, temperature, exp_noise, (int64_t)V, seed, step, nullptr, 0, tok,
                                     (hipStream_t)stream);
}
