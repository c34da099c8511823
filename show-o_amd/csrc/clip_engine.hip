// CLIP ViT vision tower + mm_projector for the w_clip_vit understanding path.
//
// Reference: models/clip_encoder.py:6-51 wraps transformers' CLIPVisionModel (openai/clip-vit-large-patch14-336) and returns
// hidden_states[-2][:, 1:] (penultimate layer, class token dropped); inference_mmu.py:133-141 feeds that through
// `model.mm_projector` (Linear 1024->2048, exact GELU, Linear 2048->2048; modeling_showo.py:48-53).  The arithmetic lives in
// the third-party package (transformers; pinned 4.41.1 by requirements.txt:203): patch embedding = 14x14 stride-14 convolution
// without bias, class token, learned position embeddings, pre-LayerNorm, then pre-LN blocks
//     x += out_proj(MHA(LN1(x)))   (16 heads x 64, scale 1/8 applied to q, no mask)
//     x += fc2(quick_gelu(fc1(LN2(x))))      quick_gelu(v) = v * sigmoid(1.702 v)
// Built from the same gfx950 kernels as the Phi stack: the patch convolution is a GEMM over an im2col image written in
// bf16 (K = 3*14*14 = 588 padded to 640), q/k/v are one packed [3H,H] projection, attention is the LDS-tiled flash kernel
// with full-visibility intervals, LayerNorm / residual epilogues as in engine.hip.  Only the layers the selected feature
// needs are run (23 of 24 for select_layer = -2; the post-LayerNorm is never used by the reference).
#include "common.h"
#include "../../include/showo_hip.h"
#include <cstring>
#include <set>
#include <string>
#include <vector>

using namespace showo;
struct showo_projector;
extern "C" int showo_projector_precise_ready(const showo_projector* p);

namespace showo {  // csrc/precise.hip: the fp32-class building blocks of the accuracy mode (shared with the Phi engine)
int precise_ln_split(const float* x, const float* w, const float* b, const int32_t* row_index, bf16_t* hi, bf16_t* lo, int rows, int H,
                     float eps, hipStream_t s);
int precise_attention(const float* Q, const float* K, const float* V, const int32_t* iv, const int32_t* flag, const float* dense, float* O,
                      int B, int nH, int Lq, int Lk, int Lcap, int ldo, hipStream_t s);
}

namespace {

#define TRY(expr)            \
    do {                     \
        int _rc = (expr);    \
        if (_rc) return _rc; \
    } while (0)

// pixel_values fp32 [B,3,S,S] -> bf16 patch rows [B*P, Kp]; column k = c*ps*ps + dy*ps + dx (the flattening of the conv weight
// [hidden,3,ps,ps]); columns >= 3*ps*ps are zero
__global__ void patchify_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, bf16_t* __restrict__ out_lo, int S, int ps, int G,
                                int Kp, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % Kp);
    const int64_t r = i / Kp;
    const int p = (int)(r % (G * G));
    const int64_t b = r / (G * G);
    const int pp = ps * ps;
    float v = 0.f;
    if (k < 3 * pp) {
        const int c = k / pp, dy = (k % pp) / ps, dx = k % ps;
        const int y = (p / G) * ps + dy, x = (p % G) * ps + dx;
        v = img[((b * 3 + c) * S + y) * S + x];
    }
    const bf16_t h = f2bf(v);
    out[i] = h;
    if (out_lo) out_lo[i] = f2bf(v - bf2f(h));
}

// x[b,0,:] = cls + pos[0];  x[b,1+p,:] = patch[b*P+p,:] + pos[1+p]      (CLIPVisionEmbeddings.forward)
__global__ void assemble_kernel(const float* __restrict__ patch, const float* __restrict__ cls, const float* __restrict__ pos,
                                float* __restrict__ x, int L, int H, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int h = (int)(i % H);
    const int64_t r = i / H;
    const int l = (int)(r % L);
    const int64_t b = r / L;
    const float v = l == 0 ? cls[h] : patch[(b * (L - 1) + (l - 1)) * H + h];
    x[i] = v + pos[(int64_t)l * H + h];
}

// LayerNorm fp32 -> fp32 in place (pre_layrnorm: its output IS the residual stream); one wave per row, two passes
__global__ __launch_bounds__(256) void ln_f32_kernel(float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                     int rows, int H, float eps) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    float* xr = x + (int64_t)r * H;
    float s = 0.f;
    for (int i = lane; i < H; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
    for (int i = lane; i < H; i += 64) { const float a = xr[i] - mean; q += a * a; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
    for (int i = lane; i < H; i += 64) xr[i] = (xr[i] - mean) * rstd * w[i] + b[i];
}

// activation between the two GEMMs of an MLP, fp32 pre-activation -> bf16 operand.  MODE 0: quick_gelu (CLIP), 1: exact GELU (erf)
template <int MODE>
__global__ void act_kernel(const float* __restrict__ f, bf16_t* __restrict__ a, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float v = f[i];
        float y;
        if (MODE == 0) y = v / (1.0f + __expf(-1.702f * v));
        else y = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
        a[i] = f2bf(y);
    }
}
// accuracy mode: the same activations with IEEE expf, output as a (hi, lo) bf16 pair
template <int MODE>
__global__ void act_split_kernel(const float* __restrict__ f, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float v = f[i];
        float y;
        if (MODE == 0) y = v / (1.0f + expf(-1.702f * v));
        else y = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
        const bf16_t h = f2bf(y);
        hi[i] = h;
        lo[i] = f2bf(y - bf2f(h));
    }
}
// accuracy mode: qkv fp32 [T, 3 nH 64] -> Q, K, V fp32 [B, nH, L, 64] (CLIPAttention: no q/k norm, no rotary; the 1/8 scale is applied
// by the attention kernel).  One thread per element.
__global__ void qkv_heads_f32_kernel(const float* __restrict__ qkv, float* __restrict__ Q, float* __restrict__ K, float* __restrict__ V, int L,
                                     int nH, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int Hq = nH * 64;
    const int c = (int)(i % (3 * Hq));
    const int64_t t = i / (3 * Hq);
    const int which = c / Hq, head = (c % Hq) / 64, d = c % 64;
    const int64_t b = t / L, l = t % L;
    float* dst = which == 0 ? Q : (which == 1 ? K : V);
    dst[((b * nH + head) * L + l) * 64 + d] = qkv[i];
}

__global__ void fill_full_intervals_kernel(int32_t* iv, int L, int rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) *reinterpret_cast<int4*>(iv + (int64_t)i * 4) = make_int4(0, L, 0, 0);
}

struct ClipLayer {
    bf16_t *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;
    bf16_t *wqkv_lo = nullptr, *wo_lo = nullptr, *w1_lo = nullptr, *w2_lo = nullptr;  // low halves (accuracy mode)
    float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
    float *ln1_w = nullptr, *ln1_b = nullptr, *ln2_w = nullptr, *ln2_b = nullptr;
};

int launch1d(int64_t n) { return (int)((n + 255) / 256); }

}  // namespace

struct showo_clip {
    showo_clip_config cfg;
    int H, F, nH, G, P, L, Kp, nRun;
    std::vector<void*> allocs;
    std::set<std::string> loaded;
    int expected = 0;
    bf16_t *wpatch = nullptr, *wpatch_lo = nullptr;
    int precision = 0;
    // accuracy-mode workspace (allocated by the first showo_clip_set_precision(c, 1))
    bf16_t *patches_lo = nullptr, *h_lo = nullptr, *act_lo = nullptr;
    std::set<std::string> lo_keys;  // GEMM weights uploaded since the low-half images exist (1 + 6 per layer when complete)
    float *p_qkv = nullptr, *p_Q = nullptr, *p_K = nullptr, *p_V = nullptr, *p_a = nullptr;
    float *cls = nullptr, *pos = nullptr, *pre_w = nullptr, *pre_b = nullptr;
    std::vector<ClipLayer> layers;
    // workspace
    bf16_t *patches = nullptr, *h = nullptr, *qkv = nullptr, *Q = nullptr, *K = nullptr, *Vt = nullptr, *attn = nullptr, *act = nullptr;
    float *pout = nullptr, *x = nullptr, *f = nullptr;
    int32_t* iv = nullptr;

    template <class T>
    int alloc(T** p, int64_t n) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, (size_t)(n > 0 ? n : 1) * sizeof(T));
        if (e != hipSuccess) return set_error_hip(e, "hipMalloc", __FILE__, __LINE__);
        allocs.push_back(q);
        *p = (T*)q;
        return 0;
    }
};

extern "C" void showo_clip_destroy(showo_clip* c) {
    if (!c) return;
    for (void* p : c->allocs) hipFree(p);
    delete c;
}

extern "C" int showo_clip_create(const showo_clip_config* cf, showo_clip** out) {
    if (!cf || !out) return set_error_msg(1, "clip_create: null argument");
    if (cf->hidden != cf->heads * 64 || (cf->hidden % 64) || (cf->ffn % 64)) return set_error_msg(1, "clip: head_dim must be 64, hidden/ffn multiples of 64");
    if (cf->patch_size <= 0 || cf->image_size % cf->patch_size) return set_error_msg(1, "clip: image_size must be a multiple of patch_size");
    if (cf->run_layers < 0 || cf->run_layers > cf->layers) return set_error_msg(1, "clip: run_layers out of range");
    showo_clip* c = new showo_clip();
    c->cfg = *cf;
    c->H = cf->hidden; c->F = cf->ffn; c->nH = cf->heads;
    c->G = cf->image_size / cf->patch_size; c->P = c->G * c->G; c->L = c->P + 1;
    c->Kp = ((3 * cf->patch_size * cf->patch_size + 63) / 64) * 64;
    c->nRun = cf->run_layers;
    const int64_t H = c->H, F = c->F, B = cf->max_batch, T = B * c->L;
    const int Lp = ((c->L + 63) / 64) * 64;
    int rc = 0;
    rc |= c->alloc(&c->wpatch, H * c->Kp); rc |= c->alloc(&c->cls, H); rc |= c->alloc(&c->pos, c->L * H);
    rc |= c->alloc(&c->pre_w, H); rc |= c->alloc(&c->pre_b, H);
    c->layers.resize(c->nRun);
    for (auto& l : c->layers) {
        rc |= c->alloc(&l.wqkv, 3 * H * H); rc |= c->alloc(&l.bqkv, 3 * H); rc |= c->alloc(&l.wo, H * H); rc |= c->alloc(&l.bo, H);
        rc |= c->alloc(&l.w1, F * H); rc |= c->alloc(&l.b1, F); rc |= c->alloc(&l.w2, H * F); rc |= c->alloc(&l.b2, H);
        rc |= c->alloc(&l.ln1_w, H); rc |= c->alloc(&l.ln1_b, H); rc |= c->alloc(&l.ln2_w, H); rc |= c->alloc(&l.ln2_b, H);
    }
    rc |= c->alloc(&c->patches, B * c->P * c->Kp); rc |= c->alloc(&c->pout, B * c->P * H);
    rc |= c->alloc(&c->x, T * H); rc |= c->alloc(&c->h, T * H); rc |= c->alloc(&c->qkv, T * 3 * H);
    rc |= c->alloc(&c->Q, T * H); rc |= c->alloc(&c->K, T * H); rc |= c->alloc(&c->Vt, B * H * Lp);
    rc |= c->alloc(&c->attn, T * H); rc |= c->alloc(&c->f, T * F); rc |= c->alloc(&c->act, T * F);
    rc |= c->alloc(&c->iv, T * 4);
    if (rc) { showo_clip_destroy(c); return rc; }
    hipMemset(c->Vt, 0, (size_t)B * H * Lp * sizeof(bf16_t));
    hipMemset(c->wpatch, 0, (size_t)H * c->Kp * sizeof(bf16_t));
    c->expected = 5 + c->nRun * 16;
    *out = c;
    return 0;
}

extern "C" int showo_clip_missing(const showo_clip* c) { return c ? c->expected - (int)c->loaded.size() : -1; }

namespace {
int copy_f32(float* dst, const float* src, int64_t n, int64_t expect, hipStream_t s) {
    if (n != expect) return set_error_msg(2, "clip_load: element count mismatch");
    SHOWO_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}
// bf16 image + its low half (w = hi + lo to 2^-17): the hi half is the round-to-nearest cast the bf16 path has always used
// GEMM weight: bf16 image, plus the low half when accuracy mode has allocated it (dst_lo == nullptr: precision 0 only -- the low halves
// are made lazily by the first showo_*_set_precision(., 1), which also asks for a re-upload: ADVICE r4)
int split_w(bf16_t* dst, bf16_t* dst_lo, const float* src, int64_t n, int64_t expect, hipStream_t s) {
    if (!dst_lo) {
        if (n != expect) return set_error_msg(2, "clip_load: element count mismatch");
        return showo_cast_f32_bf16(src, dst, n, s);
    }
    if (n != expect) return set_error_msg(2, "clip_load: element count mismatch");
    return showo_split_f32_bf16(src, dst, dst_lo, n, s);
}
// conv weight fp32 [H, 3*ps*ps] -> bf16 [H, Kp] (row stride Kp, pad columns stay zero)
__global__ void pad_rows_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, bf16_t* __restrict__ dst_lo, int K, int Kp,
                                int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const bf16_t h = f2bf(src[i]);
    dst[(i / K) * Kp + (i % K)] = h;
    if (dst_lo) dst_lo[(i / K) * Kp + (i % K)] = f2bf(src[i] - bf2f(h));
}
}  // namespace

// Load one tensor by its transformers state-dict key ("vision_model.embeddings.patch_embedding.weight",
// "vision_model.encoder.layers.7.self_attn.q_proj.weight", ...) or a projector key ("mm_projector.0.weight", ...).
// Tensors the selected feature does not need (layers >= run_layers, post_layernorm, position_ids) are accepted and ignored.
extern "C" int showo_clip_load(showo_clip* c, const char* key, const float* src, int64_t n, void* stream) {
    if (!c || !key || !src) return set_error_msg(1, "clip_load: null argument");
    hipStream_t s = (hipStream_t)stream;
    const int64_t H = c->H, F = c->F;
    std::string k(key);
    if (k.rfind("vision_tower.", 0) == 0) k = k.substr(13);  // keys of the reference's CLIPVisionTower wrapper
    // transformers >= 5 dropped the "vision_model." level of CLIPVisionModel's state dict; checkpoints (and 4.41) have it
    if (k.rfind("embeddings.", 0) == 0 || k.rfind("encoder.", 0) == 0 || k.rfind("pre_layrnorm.", 0) == 0 ||
        k.rfind("post_layernorm.", 0) == 0)
        k = "vision_model." + k;
    int rc = -1, li = -1;
    char sub[128];
    const int K = 3 * c->cfg.patch_size * c->cfg.patch_size;
    if (k == "vision_model.embeddings.patch_embedding.weight") {
        if (n != H * K) return set_error_msg(2, "clip_load: element count mismatch");
        pad_rows_kernel<<<dim3(launch1d(n)), dim3(256), 0, s>>>(src, c->wpatch, c->wpatch_lo, K, c->Kp, n);
        rc = 0;
    } else if (k == "vision_model.embeddings.class_embedding") rc = copy_f32(c->cls, src, n, H, s);
    else if (k == "vision_model.embeddings.position_embedding.weight") rc = copy_f32(c->pos, src, n, (int64_t)c->L * H, s);
    else if (k == "vision_model.pre_layrnorm.weight") rc = copy_f32(c->pre_w, src, n, H, s);  // [sic] transformers' spelling
    else if (k == "vision_model.pre_layrnorm.bias") rc = copy_f32(c->pre_b, src, n, H, s);
    else if (k == "vision_model.post_layernorm.weight" || k == "vision_model.post_layernorm.bias" ||
             k == "vision_model.embeddings.position_ids") return 0;
    else if (sscanf(k.c_str(), "vision_model.encoder.layers.%d.%127s", &li, sub) == 2 && li >= 0 && li < c->cfg.layers) {
        if (li >= c->nRun) return 0;
        ClipLayer& l = c->layers[li];
        std::string t(sub);
        if (t == "self_attn.q_proj.weight") rc = split_w(l.wqkv, l.wqkv_lo, src, n, H * H, s);
        else if (t == "self_attn.k_proj.weight") rc = split_w(l.wqkv + H * H, l.wqkv_lo ? l.wqkv_lo + H * H : nullptr, src, n, H * H, s);
        else if (t == "self_attn.v_proj.weight") rc = split_w(l.wqkv + 2 * H * H, l.wqkv_lo ? l.wqkv_lo + 2 * H * H : nullptr, src, n, H * H, s);
        else if (t == "self_attn.q_proj.bias") rc = copy_f32(l.bqkv, src, n, H, s);
        else if (t == "self_attn.k_proj.bias") rc = copy_f32(l.bqkv + H, src, n, H, s);
        else if (t == "self_attn.v_proj.bias") rc = copy_f32(l.bqkv + 2 * H, src, n, H, s);
        else if (t == "self_attn.out_proj.weight") rc = split_w(l.wo, l.wo_lo, src, n, H * H, s);
        else if (t == "self_attn.out_proj.bias") rc = copy_f32(l.bo, src, n, H, s);
        else if (t == "layer_norm1.weight") rc = copy_f32(l.ln1_w, src, n, H, s);
        else if (t == "layer_norm1.bias") rc = copy_f32(l.ln1_b, src, n, H, s);
        else if (t == "layer_norm2.weight") rc = copy_f32(l.ln2_w, src, n, H, s);
        else if (t == "layer_norm2.bias") rc = copy_f32(l.ln2_b, src, n, H, s);
        else if (t == "mlp.fc1.weight") rc = split_w(l.w1, l.w1_lo, src, n, F * H, s);
        else if (t == "mlp.fc1.bias") rc = copy_f32(l.b1, src, n, F, s);
        else if (t == "mlp.fc2.weight") rc = split_w(l.w2, l.w2_lo, src, n, H * F, s);
        else if (t == "mlp.fc2.bias") rc = copy_f32(l.b2, src, n, H, s);
    }
    if (rc == -1) return set_error_msg(3, "clip_load: unknown state-dict key");
    if (rc == 0) c->loaded.insert(k);
    const auto ends = [&](const char* suf) { const size_t m = strlen(suf); return k.size() >= m && k.compare(k.size() - m, m, suf) == 0; };
    if (rc == 0 && c->wpatch_lo && (ends("proj.weight") || ends("fc1.weight") || ends("fc2.weight") || ends("patch_embedding.weight")))
        c->lo_keys.insert(k);  // this GEMM weight now has a current low half
    return rc;
}

// 0: bf16 GEMM / attention operands (the timed default).  1: accuracy mode -- split-bf16 GEMMs, fp32 LayerNorm / attention / quick_gelu;
// the low halves of the weights are kept from every load, the fp32 workspace is allocated on first use.
extern "C" int showo_clip_set_precision(showo_clip* c, int precision) {
    if (!c) return set_error_msg(1, "clip_set_precision: null handle");
    if (precision != 0 && precision != 1) return set_error_msg(1, "clip_set_precision: 0 = bf16 operands, 1 = split bf16 (fp32-class)");
    if (precision == 1 && !c->p_qkv) {
        const int64_t H = c->H, F = c->F, T = (int64_t)c->cfg.max_batch * c->L;
        int rc = 0;
        // low halves of the GEMM weights: made here, on first use (0.6 GB at ViT-L/14-336 that a precision-0 user never pays); the
        // weights loaded so far have no low half yet -> showo_clip_precise_ready() == 0 until they are uploaded again
        rc |= c->alloc(&c->wpatch_lo, H * c->Kp);
        for (auto& l : c->layers) {
            rc |= c->alloc(&l.wqkv_lo, 3 * H * H); rc |= c->alloc(&l.wo_lo, H * H); rc |= c->alloc(&l.w1_lo, F * H); rc |= c->alloc(&l.w2_lo, H * F);
        }
        if (!rc) hipMemset(c->wpatch_lo, 0, (size_t)H * c->Kp * sizeof(bf16_t));
        c->lo_keys.clear();
        rc |= c->alloc(&c->patches_lo, (int64_t)c->cfg.max_batch * c->P * c->Kp); rc |= c->alloc(&c->h_lo, T * H);
        rc |= c->alloc(&c->act_lo, T * F); rc |= c->alloc(&c->p_qkv, T * 3 * H); rc |= c->alloc(&c->p_Q, T * H);
        rc |= c->alloc(&c->p_K, T * H); rc |= c->alloc(&c->p_V, T * H); rc |= c->alloc(&c->p_a, T * H);
        if (rc) { c->p_qkv = nullptr; return rc; }
    }
    c->precision = precision;
    return 0;
}
extern "C" int showo_clip_get_precision(const showo_clip* c) { return c ? c->precision : -1; }
// 1 when every GEMM weight has a current low half (patch embedding + q, k, v, out, fc1, fc2 per layer): precision 1 can run
extern "C" int showo_clip_precise_ready(const showo_clip* c) { return c && c->wpatch_lo && (int)c->lo_keys.size() == 1 + 6 * c->nRun; }

// images fp32 [B,3,S,S] (already normalised by the image processor) -> features fp32 [B, P, hidden] =
// CLIPVisionModel(images, output_hidden_states=True).hidden_states[run_layers][:, 1:]   (clip_encoder.py:29-37, 40-49)
extern "C" int showo_clip_features(showo_clip* c, const float* images, int B, float* features, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!c || !images || !features) return set_error_msg(1, "clip_features: null argument");
    if (B <= 0 || B > c->cfg.max_batch) return set_error_msg(5, "clip_features: batch exceeds the configured workspace");
    if (showo_clip_missing(c) != 0) return set_error_msg(4, "clip: weights missing (showo_clip_missing() != 0)");
    const int H = c->H, F = c->F, nH = c->nH, L = c->L, P = c->P, S = c->cfg.image_size;
    const int T = B * L, Lp = ((L + 63) / 64) * 64;
    const float eps = c->cfg.ln_eps;
    {   // patch embedding: im2col (bf16) + GEMM, then class token / position embeddings, then pre-LayerNorm (fp32 in place)
        const int64_t n = (int64_t)B * P * c->Kp;
        const bool precise = c->precision == 1;
        if (precise && !showo_clip_precise_ready(c))
            return set_error_msg(4, "clip (precision 1): the low halves of the weights are missing (weights were loaded before "
                                    "showo_clip_set_precision(c, 1)): upload the weights again");
        patchify_kernel<<<dim3(launch1d(n)), dim3(256), 0, s>>>(images, c->patches, precise ? c->patches_lo : nullptr, S, c->cfg.patch_size,
                                                                c->G, c->Kp, n);
        if (precise)
            TRY(showo_gemm_bf16x3(c->patches, c->patches_lo, c->Kp, c->wpatch, c->wpatch_lo, c->Kp, nullptr, 0, c->pout, H, nullptr, 0, B * P, H,
                                  c->Kp, s));
        else
            TRY(showo_gemm_bf16(c->patches, c->Kp, c->wpatch, c->Kp, nullptr, 0, c->pout, H, nullptr, 0, B * P, H, c->Kp, SHOWO_EPI_F32, s));
        const int64_t m = (int64_t)T * H;
        assemble_kernel<<<dim3(launch1d(m)), dim3(256), 0, s>>>(c->pout, c->cls, c->pos, c->x, L, H, m);
        ln_f32_kernel<<<dim3((T + 3) / 4), dim3(256), 0, s>>>(c->x, c->pre_w, c->pre_b, T, H, eps);
        fill_full_intervals_kernel<<<dim3(launch1d(T)), dim3(256), 0, s>>>(c->iv, L, T);
    }
    for (int li = 0; li < c->nRun && c->precision == 1; ++li) {
        // accuracy mode: split-bf16 (hi + lo) MFMA GEMMs, fp32 LayerNorm / attention / quick_gelu -- the same recipe as the Phi engine's
        // precision 1 (csrc/precise.hip); transformers' CLIPEncoderLayer order (modeling_clip.py, pinned 4.41.1)
        ClipLayer& l = c->layers[li];
        TRY(showo::precise_ln_split(c->x, l.ln1_w, l.ln1_b, nullptr, c->h, c->h_lo, T, H, eps, s));
        TRY(showo_gemm_bf16x3(c->h, c->h_lo, H, l.wqkv, l.wqkv_lo, H, l.bqkv, 0, c->p_qkv, 3 * H, nullptr, 0, T, 3 * H, H, s));
        const int64_t nq = (int64_t)T * 3 * H;
        qkv_heads_f32_kernel<<<dim3(launch1d(nq)), dim3(256), 0, s>>>(c->p_qkv, c->p_Q, c->p_K, c->p_V, L, nH, nq);
        TRY(showo::precise_attention(c->p_Q, c->p_K, c->p_V, c->iv, nullptr, nullptr, c->p_a, B, nH, L, L, L, H, s));
        TRY(showo_split_f32_bf16(c->p_a, c->attn, c->act_lo, (int64_t)T * H, s));
        TRY(showo_gemm_bf16x3(c->attn, c->act_lo, H, l.wo, l.wo_lo, H, l.bo, 0, c->x, H, c->x, H, T, H, H, s));
        TRY(showo::precise_ln_split(c->x, l.ln2_w, l.ln2_b, nullptr, c->h, c->h_lo, T, H, eps, s));
        TRY(showo_gemm_bf16x3(c->h, c->h_lo, H, l.w1, l.w1_lo, H, l.b1, 0, c->f, F, nullptr, 0, T, F, H, s));
        act_split_kernel<0><<<dim3(2048), dim3(256), 0, s>>>(c->f, c->act, c->act_lo, (int64_t)T * F);
        TRY(showo_gemm_bf16x3(c->act, c->act_lo, F, l.w2, l.w2_lo, F, l.b2, 0, c->x, H, c->x, H, T, H, F, s));
    }
    for (int li = 0; li < c->nRun && c->precision == 0; ++li) {
        ClipLayer& l = c->layers[li];
        TRY(showo_layernorm_f32_bf16(c->x, l.ln1_w, l.ln1_b, c->h, nullptr, T, H, eps, s));
        TRY(showo_gemm_bf16(c->h, H, l.wqkv, H, l.bqkv, 0, c->qkv, 3 * H, nullptr, 0, T, 3 * H, H, SHOWO_EPI_BF16, s));
        TRY(showo_qk_prep(c->qkv, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, c->Q, c->K, c->Vt, B, L, nH, 0, eps, 0, L, Lp, s));
        TRY(showo_attn_fwd(c->Q, c->K, c->Vt, c->iv, nullptr, nullptr, c->attn, B, nH, L, L, L, Lp, H, s));
        TRY(showo_gemm_bf16(c->attn, H, l.wo, H, l.bo, 0, c->x, H, c->x, H, T, H, H, SHOWO_EPI_RESID_F32, s));
        TRY(showo_layernorm_f32_bf16(c->x, l.ln2_w, l.ln2_b, c->h, nullptr, T, H, eps, s));
        TRY(showo_gemm_bf16(c->h, H, l.w1, H, l.b1, 0, c->f, F, nullptr, 0, T, F, H, SHOWO_EPI_F32, s));
        act_kernel<0><<<dim3(2048), dim3(256), 0, s>>>(c->f, c->act, (int64_t)T * F);
        TRY(showo_gemm_bf16(c->act, F, l.w2, F, l.b2, 0, c->x, H, c->x, H, T, H, F, SHOWO_EPI_RESID_F32, s));
    }
    SHOWO_CHECK_HIP(hipGetLastError());
    // drop the class token: rows 1..P of every image
    SHOWO_CHECK_HIP(hipMemcpy2DAsync(features, (size_t)P * H * sizeof(float), c->x + H, (size_t)L * H * sizeof(float),
                                     (size_t)P * H * sizeof(float), B, hipMemcpyDeviceToDevice, s));
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// mm_projector (modeling_showo.py:48-53: Linear(in, out) -> nn.GELU() (exact, erf) -> Linear(out, out)), applied by
// inference_mmu.py:134 / training/train_w_clip_vit.py:533-536 to the tower's features.
// ------------------------------------------------------------------------------------------------------------------
struct showo_projector {
    int in_dim, out_dim, max_rows;
    bf16_t *w0 = nullptr, *w1 = nullptr, *xb = nullptr, *act = nullptr;
    bf16_t *w0_lo = nullptr, *w1_lo = nullptr, *xb_lo = nullptr, *act_lo = nullptr;  // low halves (accuracy mode; made by the first set_precision(1))
    std::set<std::string> lo_keys;
    int precision = 0;
    float *b0 = nullptr, *b1 = nullptr, *f = nullptr;
    std::set<std::string> loaded;
    // backward workspace (allocated by the first showo_projector_backward): transposed weight images, bf16 gradients, transposes
    bf16_t *w0T = nullptr, *w1T = nullptr, *dout16 = nullptr, *dact16 = nullptr, *df16 = nullptr, *tA = nullptr, *tB = nullptr, *dx16 = nullptr;
    float* colpart = nullptr;
    bool wT_valid = false;
    int last_T = 0;
};

extern "C" void showo_projector_destroy(showo_projector* p) {
    if (!p) return;
    for (void* q : {(void*)p->w0_lo, (void*)p->w1_lo, (void*)p->xb_lo, (void*)p->act_lo})
        if (q) hipFree(q);
    for (void* q : {(void*)p->w0, (void*)p->w1, (void*)p->xb, (void*)p->act, (void*)p->b0, (void*)p->b1, (void*)p->f, (void*)p->w0T,
                    (void*)p->w1T, (void*)p->dout16, (void*)p->dact16, (void*)p->df16, (void*)p->tA, (void*)p->tB, (void*)p->dx16,
                    (void*)p->colpart})
        if (q) hipFree(q);
    delete p;
}

extern "C" int showo_projector_create(int in_dim, int out_dim, int max_rows, showo_projector** out) {
    if (!out || in_dim <= 0 || out_dim <= 0 || max_rows <= 0) return set_error_msg(1, "projector_create: bad argument");
    if ((in_dim % 64) || (out_dim % 64)) return set_error_msg(1, "projector: dimensions must be multiples of 64");
    showo_projector* p = new showo_projector();
    p->in_dim = in_dim; p->out_dim = out_dim; p->max_rows = max_rows;
    const int64_t I = in_dim, D = out_dim, T = max_rows;
    bool ok = hipMalloc((void**)&p->w0, D * I * 2) == hipSuccess && hipMalloc((void**)&p->w1, D * D * 2) == hipSuccess &&
              hipMalloc((void**)&p->b0, D * 4) == hipSuccess && hipMalloc((void**)&p->b1, D * 4) == hipSuccess &&
              hipMalloc((void**)&p->xb, T * I * 2) == hipSuccess && hipMalloc((void**)&p->f, T * D * 4) == hipSuccess &&
              hipMalloc((void**)&p->act, T * D * 2) == hipSuccess;
    if (!ok) { showo_projector_destroy(p); return set_error_msg(7, "projector_create: hipMalloc failed"); }
    *out = p;
    return 0;
}

// keys "0.weight" [out,in], "0.bias", "2.weight" [out,out], "2.bias" (nn.Sequential numbering; an "mm_projector." prefix is accepted)
extern "C" int showo_projector_load(showo_projector* p, const char* key, const float* src, int64_t n, void* stream) {
    if (!p || !key || !src) return set_error_msg(1, "projector_load: null argument");
    hipStream_t s = (hipStream_t)stream;
    std::string k(key);
    if (k.rfind("mm_projector.", 0) == 0) k = k.substr(13);
    const int64_t I = p->in_dim, D = p->out_dim;
    int rc;
    if (k == "0.weight") rc = split_w(p->w0, p->w0_lo, src, n, D * I, s);
    else if (k == "0.bias") rc = copy_f32(p->b0, src, n, D, s);
    else if (k == "2.weight") rc = split_w(p->w1, p->w1_lo, src, n, D * D, s);
    else if (k == "2.bias") rc = copy_f32(p->b1, src, n, D, s);
    else return set_error_msg(3, "projector_load: unknown key");
    if (rc == 0) { p->loaded.insert(k); p->wT_valid = false; }
    if (rc == 0 && p->w0_lo && (k == "0.weight" || k == "2.weight")) p->lo_keys.insert(k);
    return rc;
}

extern "C" int showo_projector_set_precision(showo_projector* p, int precision) {
    if (!p) return set_error_msg(1, "projector_set_precision: null handle");
    if (precision != 0 && precision != 1) return set_error_msg(1, "projector_set_precision: 0 = bf16 operands, 1 = split bf16 (fp32-class)");
    if (precision == 1 && !p->w0_lo) {  // low halves on first use; the weights must be uploaded again (showo_projector_precise_ready)
        const int64_t I = p->in_dim, D = p->out_dim, T = p->max_rows;
        const bool ok = hipMalloc((void**)&p->w0_lo, D * I * 2) == hipSuccess && hipMalloc((void**)&p->w1_lo, D * D * 2) == hipSuccess &&
                        hipMalloc((void**)&p->xb_lo, T * I * 2) == hipSuccess && hipMalloc((void**)&p->act_lo, T * D * 2) == hipSuccess;
        if (!ok) return set_error_msg(7, "projector_set_precision: hipMalloc failed");
        p->lo_keys.clear();
    }
    p->precision = precision;
    return 0;
}
extern "C" int showo_projector_precise_ready(const showo_projector* p) { return p && p->w0_lo && p->lo_keys.size() == 2; }

// x fp32 [T, in] -> out fp32 [T, out] = W1 gelu(W0 x + b0) + b1
extern "C" int showo_projector_forward(showo_projector* p, const float* x, int T, float* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!p || !x || !out) return set_error_msg(1, "projector_forward: null argument");
    if (T <= 0 || T > p->max_rows) return set_error_msg(5, "projector_forward: too many rows for the workspace");
    if (p->loaded.size() != 4) return set_error_msg(4, "projector_forward: weights missing");
    const int I = p->in_dim, D = p->out_dim;
    if (p->precision == 1 && !showo_projector_precise_ready(p))
        return set_error_msg(4, "projector (precision 1): the low halves of the weights are missing (weights were loaded before "
                                "showo_projector_set_precision(p, 1)): upload the weights again");
    if (p->precision == 1) {  // accuracy mode: split-bf16 GEMMs, exact GELU in fp32 -> (hi, lo); inference only (the backward keeps bf16)
        TRY(showo_split_f32_bf16(x, p->xb, p->xb_lo, (int64_t)T * I, s));
        TRY(showo_gemm_bf16x3(p->xb, p->xb_lo, I, p->w0, p->w0_lo, I, p->b0, 0, p->f, D, nullptr, 0, T, D, I, s));
        act_split_kernel<1><<<dim3(1024), dim3(256), 0, s>>>(p->f, p->act, p->act_lo, (int64_t)T * D);
        TRY(showo_gemm_bf16x3(p->act, p->act_lo, D, p->w1, p->w1_lo, D, p->b1, 0, out, D, nullptr, 0, T, D, D, s));
        SHOWO_CHECK_HIP(hipGetLastError());
        p->last_T = T;
        return 0;
    }
    TRY(showo_cast_f32_bf16(x, p->xb, (int64_t)T * I, s));
    TRY(showo_gemm_bf16(p->xb, I, p->w0, I, p->b0, 0, p->f, D, nullptr, 0, T, D, I, SHOWO_EPI_F32, s));
    act_kernel<1><<<dim3(1024), dim3(256), 0, s>>>(p->f, p->act, (int64_t)T * D);
    TRY(showo_gemm_bf16(p->act, D, p->w1, D, p->b1, 0, out, D, nullptr, 0, T, D, D, SHOWO_EPI_F32, s));
    SHOWO_CHECK_HIP(hipGetLastError());
    p->last_T = T;
    return 0;
}

namespace {
// df = d_act * gelu'(f), exact GELU: gelu'(v) = Phi(v) + v phi(v); fp32 pre-activation saved by the forward
__global__ void dgelu_erf_kernel(const bf16_t* __restrict__ da, const float* __restrict__ f, bf16_t* __restrict__ df, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float v = f[i];
        const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
        df[i] = f2bf(bf2f(da[i]) * (cdf + v * pdf));
    }
}
}  // namespace

// Backward of the LAST showo_projector_forward (same x, T): given dout fp32 [T,out] it writes the parameter gradients
// gw0 fp32 [out,in], gb0 [out], gw1 [out,out], gb1 [out] and (optional) dx fp32 [T,in].  Same machinery as the trainer
// (train_engine.hip): everything on the NT GEMM with transposed bf16 images, bias gradients as fixed-order column sums.
extern "C" int showo_projector_backward(showo_projector* p, const float* dout, int T, float* dx, float* gw0, float* gb0,
                                        float* gw1, float* gb1, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!p || !dout || !gw0 || !gb0 || !gw1 || !gb1) return set_error_msg(1, "projector_backward: null argument");
    if (T <= 0 || T != p->last_T) return set_error_msg(1, "projector_backward: run showo_projector_forward on the same rows first");
    const int I = p->in_dim, D = p->out_dim;
    const int Tp = ((T + 63) / 64) * 64, Tm = ((p->max_rows + 63) / 64) * 64;
    if (!p->w0T) {
        const int64_t big = (int64_t)(D > I ? D : I) * Tm;
        bool ok = hipMalloc((void**)&p->w0T, (size_t)I * D * 2) == hipSuccess && hipMalloc((void**)&p->w1T, (size_t)D * D * 2) == hipSuccess &&
                  hipMalloc((void**)&p->dout16, (size_t)Tm * D * 2) == hipSuccess && hipMalloc((void**)&p->dact16, (size_t)Tm * D * 2) == hipSuccess &&
                  hipMalloc((void**)&p->df16, (size_t)Tm * D * 2) == hipSuccess && hipMalloc((void**)&p->tA, (size_t)big * 2) == hipSuccess &&
                  hipMalloc((void**)&p->tB, (size_t)big * 2) == hipSuccess && hipMalloc((void**)&p->dx16, (size_t)Tm * I * 2) == hipSuccess &&
                  hipMalloc((void**)&p->colpart, (size_t)(Tm / 64 + 8) * (D > I ? D : I) * 4) == hipSuccess;
        if (!ok) return set_error_msg(7, "projector_backward: hipMalloc failed");
    }
    if (!p->wT_valid) {  // transposed weight images for the two data-gradient GEMMs (remade when the weights change)
        TRY(showo_transpose_bf16(p->w0, I, p->w0T, D, I, D, 0, nullptr, nullptr, 0, s));   // [out,in] -> [in,out]
        TRY(showo_transpose_bf16(p->w1, D, p->w1T, D, D, D, 0, nullptr, nullptr, 0, s));
        p->wT_valid = true;
    }
    TRY(showo_cast_f32_bf16(dout, p->dout16, (int64_t)T * D, s));
    // second Linear: gb1 = colsum(dout), gw1 = dout^T act, d_act = dout W1
    TRY(showo_transpose_bf16(p->dout16, D, p->tA, T, D, Tp, 0, p->colpart, gb1, 0, s));
    TRY(showo_transpose_bf16(p->act, D, p->tB, T, D, Tp, 0, nullptr, nullptr, 0, s));
    TRY(showo_gemm_bf16(p->tA, Tp, p->tB, Tp, nullptr, 0, gw1, D, nullptr, 0, D, D, Tp, SHOWO_EPI_F32, s));
    TRY(showo_gemm_bf16(p->dout16, D, p->w1T, D, nullptr, 0, p->dact16, D, nullptr, 0, T, D, D, SHOWO_EPI_BF16, s));
    dgelu_erf_kernel<<<dim3(1024), dim3(256), 0, s>>>(p->dact16, p->f, p->df16, (int64_t)T * D);
    // first Linear: gb0 = colsum(df), gw0 = df^T x, dx = df W0
    TRY(showo_transpose_bf16(p->df16, D, p->tA, T, D, Tp, 0, p->colpart, gb0, 0, s));
    TRY(showo_transpose_bf16(p->xb, I, p->tB, T, I, Tp, 0, nullptr, nullptr, 0, s));
    TRY(showo_gemm_bf16(p->tA, Tp, p->tB, Tp, nullptr, 0, gw0, I, nullptr, 0, D, I, Tp, SHOWO_EPI_F32, s));
    if (dx) TRY(showo_gemm_bf16(p->df16, D, p->w0T, D, nullptr, 0, dx, I, nullptr, 0, T, I, D, SHOWO_EPI_F32, s));
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}
