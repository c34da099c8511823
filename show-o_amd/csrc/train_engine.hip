// Training step of the Show-o transformer on gfx950: forward with saved activations, the three cross-entropies of
// Showo.forward and the full backward into fp32 gradient buffers that carry the reference's state-dict names.
// Replaces (reference): Showo.forward with labels (models/modeling_showo.py:59-102) + loss.backward()
// (training/train.py:590-612) for the Phi stack (models/phi.py:774-790, 953-1183).
//
// GEMM forms (all on the one NT kernel C = A W^T, K-contiguous operands):
//   forward   Y[T,N]   = X[T,K]     W[N,K]^T
//   dgrad     dX[T,K]  = dY[T,N]    (W^T)[K,N]^T        -> needs the transposed weight copy W^T (made at load time)
//   wgrad     dW[N,K]  = (dY^T)[N,Tp] (X^T)[K,Tp]^T     -> needs token-contiguous images of dY and X
//             (showo_transpose_bf16; the bias gradient = column sums of dY comes out of the same pass)
// Saved per layer: x (fp32 block input), h = LN(x), raw qkv, Q, K, V^T, lse, attention output, fc1 pre-activation.
#include "engine.h"
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <map>

using namespace showo;

#define TRY(expr)            \
    do {                     \
        int _rc = (expr);    \
        if (_rc) return _rc; \
    } while (0)

namespace {
struct LayerT {
    bf16_t *wqkvT = nullptr, *wdT = nullptr, *w1T = nullptr, *w2T = nullptr;  // [K_in, N_out] images for dgrad
    // saved activations
    float* x = nullptr;
    bf16_t *h = nullptr, *qkv = nullptr, *Q = nullptr, *K = nullptr, *Vt = nullptr, *attn = nullptr, *f = nullptr;
    bf16_t* a = nullptr;  // gelu_new(f) [T, F]: the token-major operand of the fc2 weight gradient (showo_gemm_tn_bf16)
    float* lse = nullptr;
    // gradients (fp32, reference parameter layout)
    float *gwqkv = nullptr, *gbqkv = nullptr, *gwd = nullptr, *gbd = nullptr, *gw1 = nullptr, *gb1 = nullptr, *gw2 = nullptr, *gb2 = nullptr;
    float *gln = nullptr;   // [2,H] (weight, bias)
    float* gqk = nullptr;   // [4,64] (q_ln w, b, k_ln w, b)
};
struct Grad { float* p; int64_t n; };
struct Bound { std::string key; float *p, *m, *v; int64_t n; bool decay; };

// SHOWO_TRAIN_TN (default 1): weight gradients by showo_gemm_tn_bf16 on the token-major tensors the backward already holds (dY, and the
// activations saved by the forward) + showo_colsum_bf16 for the bias gradients; 0 = transpose both operands and run the k-contiguous
// GEMM (the round-2 path: 243 transposes per step).  Read once per process: the per-layer gelu(fc1) buffers exist only in TN mode.
bool train_tn() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SHOWO_TRAIN_TN"); v = e ? (atoi(e) != 0) : 1; }
    return v != 0;
}
}  // namespace

struct showo_trainer {
    showo_engine* e;
    int maxB, maxL, Tmax, Tp, Lp, Vp;
    std::vector<void*> allocs;
    std::vector<LayerT> L;
    bf16_t* wlmT = nullptr;
    std::map<std::string, Grad> grads;
    std::vector<Bound> bound;  // master weights + AdamW moments registered by the host (showo_train_bind_param)
    showo::AdamSeg* adam_segs = nullptr;  // device tables of the multi-tensor optimizer launch (train_kernels.hip, adamw_multi_kernel)
    int* adam_seg_of = nullptr;
    int64_t* adam_start_of = nullptr;
    int adam_chunks = 0;
    bool adam_dirty = true;
    // head
    float *logits = nullptr, *gembed = nullptr, *gfln = nullptr, *gwlm = nullptr, *gblm = nullptr;
    bf16_t *dlogits = nullptr, *bigT = nullptr;  // bigT: [max(Vp, F, 3H), Tp] transposed image of the dY side (SHOWO_TRAIN_TN=0 only)
    // backward scratch
    float *dy = nullptr, *dh = nullptr, *colpart = nullptr, *lnpart = nullptr, *qkpart = nullptr, *D = nullptr, *rowloss = nullptr;
    bf16_t *dy16 = nullptr, *d_o = nullptr, *dff = nullptr, *dqk = nullptr, *dqkv = nullptr, *xT = nullptr, *QT = nullptr, *KT = nullptr, *dOT = nullptr;
    void* ce_rows = nullptr;
    int* counts = nullptr;
    int* order_ws = nullptr;
    float* losses = nullptr;
    int64_t* ids = nullptr;
    // state of the last forward
    int B = 0, Lq = 0;
    bool have_fwd = false;
    bool has_mask = false;
    bool from_embeds = false;  // last forward started from caller-provided embeddings: d(loss)/d(embeddings) = dy
    bool weights_synced = false;
    // loss weights announced before the forward (showo_train_set_loss_weights): the forward's cross-entropy pass then writes the
    // logit gradients as well, and a backward with the same labels / split / weights skips its own pass over the 2.6 GB of logits
    bool lw_set = false;
    float lw[3] = {0.f, 0.f, 0.f};
    bool dl_valid = false;
    float dl_g[3] = {0.f, 0.f, 0.f};
    const int64_t* dl_labels = nullptr;
    int dl_split[4] = {0, 0, 0, 0};

    template <class T>
    int alloc(T** p, int64_t n) {
        void* q = nullptr;
        hipError_t err = hipMalloc(&q, (size_t)(n > 0 ? n : 1) * sizeof(T));
        if (err != hipSuccess) return set_error_hip(err, "hipMalloc(trainer)", __FILE__, __LINE__);
        allocs.push_back(q);
        *p = (T*)q;
        return 0;
    }
    // gradients live in ONE flat fp32 buffer (bucket = contiguous range: embed | layer 0 .. nL-1 | head), so a data-parallel
    // job all-reduces a bucket with one collective while the backward of the next block runs
    float* gflat = nullptr;
    int64_t gflat_n = 0, gcursor = 0;
    std::vector<std::pair<int64_t, int64_t>> buckets;  // (offset, count)
    float* carve(int64_t n) {
        n = (n + 63) & ~(int64_t)63;  // keep every tensor 256-B aligned
        float* p = gflat + gcursor;
        gcursor += n;
        return p;
    }
    int galloc(const std::string& key, float** p, int64_t n) {
        *p = carve(n);
        grads[key] = Grad{*p, n};
        return 0;
    }
};

extern "C" int showo_train_create(showo_engine* e, int max_batch, int max_seq, showo_trainer** out) {
    if (!e || !out) return set_error_msg(1, "train_create: null argument");
    if (max_batch > e->cfg.max_batch || max_seq > e->cfg.max_seq || (int64_t)max_batch * max_seq > e->maxT)
        return set_error_msg(5, "train_create: the engine workspace is smaller than the training batch");
    if (e->cfg.rotary_dim != 32) return set_error_msg(1, "train: rotary_dim 32 only");
    if (e->H % 64 || e->H > 2048) return set_error_msg(1, "train: hidden must be a multiple of 64 and <= 2048");
    showo_trainer* t = new showo_trainer();
    t->e = e;
    t->maxB = max_batch; t->maxL = max_seq;
    t->Tmax = max_batch * max_seq;
    t->Tp = ((t->Tmax + 63) / 64) * 64;
    t->Lp = ((max_seq + 63) / 64) * 64;
    t->Vp = ((e->V + 255) / 256) * 256;  // whole 256-column tiles: the lm_head weight gradient (gemm_tn.hip) fetches its dY columns unchecked
    const int64_t H = e->H, F = e->F, V = e->V, T = t->Tmax, Tp = t->Tp, Vp = t->Vp, nH = e->nH;
    int rc = 0;
    t->L.resize(e->nL);
    char key[160];
    {
        auto al = [](int64_t n) { return (n + 63) & ~(int64_t)63; };
        const int64_t per_layer = al(3 * H * H) + al(3 * H) + al(H * H) + al(H) + al(F * H) + al(F) + al(H * F) + al(H) + al(2 * H) + al(256);
        t->gflat_n = al(V * H) + e->nL * per_layer + al(2 * H) + al(V * H) + al(Vp);
        rc |= t->alloc(&t->gflat, t->gflat_n);
        if (rc) { showo_train_destroy(t); return rc; }
        hipMemset(t->gflat, 0, (size_t)t->gflat_n * sizeof(float));
    }
    { int64_t c0 = t->gcursor; t->galloc("showo.model.embed_tokens.weight", &t->gembed, V * H); t->buckets.push_back({c0, t->gcursor - c0}); }
    for (int i = 0; i < e->nL; ++i) {
        LayerT& l = t->L[i];
        rc |= t->alloc(&l.wqkvT, H * 3 * H); rc |= t->alloc(&l.wdT, H * H); rc |= t->alloc(&l.w1T, H * F); rc |= t->alloc(&l.w2T, F * H);
        // token-major operands of showo_gemm_tn_bf16 (h, attn, a; dy16 / dff / dqkv / dlogits below) are allocated with Tp rows: the
        // kernel reads -- and zeroes in registers -- the rows between T and the next multiple of 64
        rc |= t->alloc(&l.x, T * H); rc |= t->alloc(&l.h, Tp * H); rc |= t->alloc(&l.qkv, T * 3 * H);
        rc |= t->alloc(&l.Q, T * H); rc |= t->alloc(&l.K, T * H); rc |= t->alloc(&l.Vt, (int64_t)max_batch * H * t->Lp);
        rc |= t->alloc(&l.attn, Tp * H); rc |= t->alloc(&l.f, T * F); rc |= t->alloc(&l.lse, (int64_t)max_batch * nH * max_seq);
        if (train_tn()) rc |= t->alloc(&l.a, Tp * F);
        if (rc) break;
        hipMemset(l.Vt, 0, (size_t)max_batch * H * t->Lp * sizeof(bf16_t));
        const char* names[3] = {"q_proj", "k_proj", "v_proj"};
        const int64_t lc0 = t->gcursor;
        l.gwqkv = t->carve(3 * H * H); l.gbqkv = t->carve(3 * H);
        for (int j = 0; j < 3 && !rc; ++j) {
            snprintf(key, sizeof key, "showo.model.layers.%d.self_attn.%s.weight", i, names[j]);
            t->grads[key] = Grad{l.gwqkv + j * H * H, H * H};
            snprintf(key, sizeof key, "showo.model.layers.%d.self_attn.%s.bias", i, names[j]);
            t->grads[key] = Grad{l.gbqkv + j * H, H};
        }
        snprintf(key, sizeof key, "showo.model.layers.%d.self_attn.dense.weight", i); rc |= t->galloc(key, &l.gwd, H * H);
        snprintf(key, sizeof key, "showo.model.layers.%d.self_attn.dense.bias", i); rc |= t->galloc(key, &l.gbd, H);
        snprintf(key, sizeof key, "showo.model.layers.%d.mlp.fc1.weight", i); rc |= t->galloc(key, &l.gw1, F * H);
        snprintf(key, sizeof key, "showo.model.layers.%d.mlp.fc1.bias", i); rc |= t->galloc(key, &l.gb1, F);
        snprintf(key, sizeof key, "showo.model.layers.%d.mlp.fc2.weight", i); rc |= t->galloc(key, &l.gw2, H * F);
        snprintf(key, sizeof key, "showo.model.layers.%d.mlp.fc2.bias", i); rc |= t->galloc(key, &l.gb2, H);
        l.gln = t->carve(2 * H);
        l.gqk = t->carve(256);
        t->buckets.push_back({lc0, t->gcursor - lc0});
        snprintf(key, sizeof key, "showo.model.layers.%d.input_layernorm.weight", i); t->grads[key] = Grad{l.gln, H};
        snprintf(key, sizeof key, "showo.model.layers.%d.input_layernorm.bias", i); t->grads[key] = Grad{l.gln + H, H};
        const char* qk[4] = {"q_layernorm.weight", "q_layernorm.bias", "k_layernorm.weight", "k_layernorm.bias"};
        for (int j = 0; j < 4; ++j) {
            snprintf(key, sizeof key, "showo.model.layers.%d.self_attn.%s", i, qk[j]);
            t->grads[key] = Grad{l.gqk + 64 * j, 64};
        }
    }
    rc |= t->alloc(&t->wlmT, H * Vp);
    const int64_t hc0 = t->gcursor;
    t->gfln = t->carve(2 * H);
    t->galloc("showo.lm_head.weight", &t->gwlm, V * H);
    t->galloc("showo.lm_head.bias", &t->gblm, Vp);
    t->buckets.push_back({hc0, t->gcursor - hc0});
    if (!rc) {
        t->grads["showo.lm_head.bias"].n = V;
        t->grads["showo.model.final_layernorm.weight"] = Grad{t->gfln, H};
        t->grads["showo.model.final_layernorm.bias"] = Grad{t->gfln + H, H};
    }
    rc |= t->alloc(&t->logits, T * V);
    rc |= t->alloc(&t->dlogits, Tp * Vp);
    // bigT holds the transposed dY side of every wgrad GEMM: dlogits^T [Vp, Tp], df^T [F, Tp], dqkv^T [3H, Tp] -- the tallest wins
    // (round 3: sizing it by max(Vp, F) alone overflowed for geometries with 3H > max(Vp, F), found by the SMALL training fixture)
    const int64_t bigrows = std::max<int64_t>(std::max<int64_t>(Vp, F), 3 * H);
    if (!train_tn()) {  // the transposed operand images exist only on the transpose + NT path (1.5 GB at the stage-1 geometry)
        rc |= t->alloc(&t->bigT, bigrows * Tp);
        rc |= t->alloc(&t->xT, std::max<int64_t>(F, H) * Tp);
    }
    rc |= t->alloc(&t->dy, T * H); rc |= t->alloc(&t->dh, T * H); rc |= t->alloc(&t->dy16, Tp * H); rc |= t->alloc(&t->d_o, T * H);
    rc |= t->alloc(&t->dff, Tp * F); rc |= t->alloc(&t->dqk, T * 2 * H); rc |= t->alloc(&t->dqkv, Tp * 3 * H);
    rc |= t->alloc(&t->QT, (int64_t)max_batch * H * t->Lp); rc |= t->alloc(&t->KT, (int64_t)max_batch * H * t->Lp);
    rc |= t->alloc(&t->dOT, (int64_t)max_batch * H * t->Lp);
    rc |= t->alloc(&t->D, (int64_t)max_batch * nH * max_seq);
    rc |= t->alloc(&t->colpart, (Tp / 32 + 8) * bigrows);  // showo_colsum_bf16 writes one partial row per 32 tokens
    rc |= t->alloc(&t->lnpart, (int64_t)showo_ln_bwd_blocks((int)T) * 3 * H);
    rc |= t->alloc(&t->qkpart, (int64_t)showo_qkln_rope_bwd_blocks((int)T, (int)nH) * 256);
    rc |= t->alloc(&t->rowloss, 2 * T);
    rc |= t->alloc((char**)&t->ce_rows, 12 * T);
    rc |= t->alloc(&t->counts, 4);
    rc |= t->alloc(&t->order_ws, 2 * T);
    rc |= t->alloc(&t->losses, 4);
    rc |= t->alloc(&t->ids, T);
    if (rc) { showo_train_destroy(t); return rc; }
    *out = t;
    return 0;
}

extern "C" void showo_train_destroy(showo_trainer* t) {
    if (!t) return;
    for (void* p : t->allocs) hipFree(p);
    if (t->adam_segs) { hipFree(t->adam_segs); hipFree(t->adam_seg_of); hipFree(t->adam_start_of); }
    delete t;
}

extern "C" int showo_train_invalidate_weights(showo_trainer* t) {
    if (!t) return set_error_msg(1, "train: null handle");
    t->weights_synced = false;
    return 0;
}

// transposed bf16 weight images for the dgrad GEMMs (call after the engine's weights changed)
static int sync_weights(showo_trainer* t, hipStream_t s) {
    if (t->weights_synced) return 0;
    showo_engine* e = t->e;
    const int H = e->H, F = e->F, V = e->V;
    for (int i = 0; i < e->nL; ++i) {
        Layer& w = e->layers[i];
        LayerT& l = t->L[i];
        TRY(showo_transpose_bf16(w.wqkv, H, l.wqkvT, 3 * H, H, 3 * H, 0, nullptr, nullptr, 0, s));  // [3H,H] -> [H,3H]
        TRY(showo_transpose_bf16(w.wd, H, l.wdT, H, H, H, 0, nullptr, nullptr, 0, s));
        TRY(showo_transpose_bf16(w.w1, H, l.w1T, F, H, F, 0, nullptr, nullptr, 0, s));                // [F,H] -> [H,F]
        TRY(showo_transpose_bf16(w.w2, F, l.w2T, H, F, H, 0, nullptr, nullptr, 0, s));                // [H,F] -> [F,H]
    }
    TRY(showo_transpose_bf16(e->wlm, H, t->wlmT, V, H, t->Vp, 0, nullptr, nullptr, 0, s));            // [V,H] -> [H,Vp]
    t->weights_synced = true;
    return 0;
}

// Visibility intervals built on the device (showo_mask_predict_next / _mmu) instead of a dense mask: the next
// showo_train_forward call that passes mask == NULL attends (forward and backward) with iv int32 [B,L,4]; NULL restores causal.
// flag (optional, int32[1] written by the interval builders): non-zero = some row needs more than two visibility runs, the
// intervals do not describe the mask; checked on the device, without a host sync: the three losses come back as NaN.
extern "C" int showo_trainer_use_intervals(showo_trainer* t, const int32_t* iv, const int32_t* flag) {
    if (!t) return set_error_msg(1, "trainer: null handle");
    t->e->ext_iv = iv;
    t->e->ext_flag = iv ? flag : nullptr;
    return 0;
}

namespace {
__global__ void poison_losses_kernel(float* __restrict__ losses, const int32_t* __restrict__ flag) {
    if (threadIdx.x < 3 && *flag != 0) losses[threadIdx.x] = __builtin_nanf("");
}
}  // namespace

static int train_forward_impl(showo_trainer* t, const int64_t* ids, const float* embeds, const float* mask, const int64_t* labels,
                              int B, int L, int b_t2i, int b_lm, int b_mmu, int max_seq_len, float* logits_out, float* losses_out,
                              void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!t || ((ids == nullptr) == (embeds == nullptr))) return set_error_msg(1, "train_forward: exactly one of ids / embeds");
    showo_engine* e = t->e;
    if (showo_engine_missing(e) != 0) return set_error_msg(4, "train: weights missing");
    if (B > t->maxB || L > t->maxL || (int64_t)B * L > t->Tmax) return set_error_msg(5, "train: batch exceeds the trainer workspace");
    if ((e->H % 64) || (e->F % 64)) return set_error_msg(1, "train: hidden/ffn must be multiples of 64");
    if (e->img_f16 || e->precision == 2)
        return set_error_msg(4, "train: the engine's weight images hold fp16 (precision 2); training runs on bf16 images -- "
                                "showo_engine_set_precision(e, 0) and upload the weights again");
    TRY(sync_weights(t, s));
    const int H = e->H, F = e->F, V = e->V, nH = e->nH, T = B * L;
    const int Lp = ((L + 63) / 64) * 64;
    if (ids) {
        SHOWO_CHECK_HIP(hipMemcpyAsync(t->ids, ids, (size_t)T * 8, hipMemcpyDeviceToDevice, s));
        TRY(showo_embed_f32(ids, e->embed, t->L[0].x, T, H, V, s));  // layer 0's saved input IS the embedding output (no copy)
    } else {  // inputs_embeds path (modeling_showo.py:77-78, phi.py:1005-1006): the residual stream starts from the caller's rows
        SHOWO_CHECK_HIP(hipMemcpyAsync(t->L[0].x, embeds, (size_t)T * H * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    t->from_embeds = ids == nullptr;
    const int32_t *iv = nullptr, *flag = nullptr;
    if (mask) {
        TRY(showo_mask_compress(mask, e->iv, e->flag, B, L, L, s));
        iv = e->iv; flag = e->flag;
    } else if (e->ext_iv) {
        // caller-built visibility intervals (showo_trainer_use_intervals): kept in the engine's own buffer for the backward
        SHOWO_CHECK_HIP(hipMemcpyAsync(e->iv, e->ext_iv, (size_t)T * 4 * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        if (e->ext_flag) SHOWO_CHECK_HIP(hipMemcpyAsync(e->flag, e->ext_flag, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        else SHOWO_CHECK_HIP(hipMemsetAsync(e->flag, 0, sizeof(int32_t), s));
        iv = e->iv; flag = e->flag;
    }
    // SHOWO_TRAIN_FUSED_PROJ (default 1): the fused projection launch of the inference layer, in its save-for-backward form
    static int fused_env = -1;
    if (fused_env < 0) { const char* env = getenv("SHOWO_TRAIN_FUSED_PROJ"); fused_env = env ? (atoi(env) != 0) : 1; }
    // SHOWO_TRAIN_QKPREP (default 0): 1 = raw-only projection launch + showo_qk_prep (A/B of the epilogue's LayerNorm / RoPE cost)
    static int qkprep_env = -1;
    if (qkprep_env < 0) { const char* env = getenv("SHOWO_TRAIN_QKPREP"); qkprep_env = env ? (atoi(env) != 0) : 0; }
    const bool fused_proj = fused_env && T >= 256 && e->cfg.rotary_dim == 32 && (3 * H) % 256 == 0 && (F % 8) == 0 &&
                            (int64_t)T * F * 2 < ((int64_t)1 << 32);
    for (int i = 0; i < e->nL; ++i) {
        Layer& w = e->layers[i];
        LayerT& l = t->L[i];
        // The layer's input (saved for ln_bwd) lives in l.x already: the previous layer's fc2 epilogue wrote it there.  The residual
        // stream ping-pongs l.x -> e->x (after dense) -> L[i + 1].x (after fc2; e->x for the last layer): no 92 MB copy per layer.
        float* xnext = (i + 1 < e->nL) ? t->L[i + 1].x : e->x;
        bf16_t* ffn = l.a ? l.a : e->ffn;  // TN mode keeps gelu(fc1) per layer for the fc2 weight gradient
        TRY(showo_layernorm_f32_bf16(l.x, w.ln_w, w.ln_b, l.h, nullptr, T, H, e->cfg.ln_eps, s));
        if (fused_proj) {
            // q/k/v_proj + q/k LayerNorm + RoPE + relayout AND fc1 + gelu_new in one launch that also saves qkv and the fc1
            // pre-activation for backward ([Wqkv ; W1] is one allocation, engine.hip); same bits as the four launches below
            if (qkprep_env) {
                // raw-only form: the launch stores qkv / the fc1 pre-activation / gelu and the q/k LayerNorm + RoPE + relayout run as
                // showo_qk_prep on the saved qkv (same bits: both start from the rounded values)
                TRY(showo_gemm_qkv_fc1_save_bf16(l.h, H, w.wqkv, H, w.bqkv, w.qln_w, w.qln_b, w.kln_w, w.kln_b, e->cosT, e->sinT, nullptr, nullptr,
                                                 nullptr, l.qkv, 3 * H, l.f, ffn, F, F, B, L, nH, e->cfg.rotary_dim, e->cfg.ln_eps, 0, L, Lp, 0, s));
                TRY(showo_qk_prep(l.qkv, w.qln_w, w.qln_b, w.kln_w, w.kln_b, e->cosT, e->sinT, l.Q, l.K, l.Vt, B, L, nH, e->cfg.rotary_dim,
                                  e->cfg.ln_eps, 0, L, Lp, s));
            } else
            TRY(showo_gemm_qkv_fc1_save_bf16(l.h, H, w.wqkv, H, w.bqkv, w.qln_w, w.qln_b, w.kln_w, w.kln_b, e->cosT, e->sinT, l.Q, l.K, l.Vt,
                                             l.qkv, 3 * H, l.f, ffn, F, F, B, L, nH, e->cfg.rotary_dim, e->cfg.ln_eps, 0, L, Lp, 0, s));
            TRY(showo_attn_fwd_lse(l.Q, l.K, l.Vt, iv, flag, mask, l.attn, l.lse, B, nH, L, L, L, Lp, H, s));
            TRY(showo_gemm_bf16(l.attn, H, w.wd, H, w.bd, 0, e->x, H, l.x, H, T, H, H, SHOWO_EPI_RESID_F32, s));
        } else {
            TRY(showo_gemm_bf16(l.h, H, w.wqkv, H, w.bqkv, 0, l.qkv, 3 * H, nullptr, 0, T, 3 * H, H, SHOWO_EPI_BF16, s));
            TRY(showo_qk_prep(l.qkv, w.qln_w, w.qln_b, w.kln_w, w.kln_b, e->cosT, e->sinT, l.Q, l.K, l.Vt, B, L, nH, e->cfg.rotary_dim,
                              e->cfg.ln_eps, 0, L, Lp, s));
            TRY(showo_attn_fwd_lse(l.Q, l.K, l.Vt, iv, flag, mask, l.attn, l.lse, B, nH, L, L, L, Lp, H, s));
            TRY(showo_gemm_bf16(l.attn, H, w.wd, H, w.bd, 0, e->x, H, l.x, H, T, H, H, SHOWO_EPI_RESID_F32, s));
            TRY(showo_gemm_bf16(l.h, H, w.w1, H, w.b1, 0, l.f, F, nullptr, 0, T, F, H, SHOWO_EPI_BF16, s));
            TRY(showo_gelu_bf16(l.f, ffn, (int64_t)T * F, s));
        }
        TRY(showo_gemm_bf16(ffn, F, w.w2, F, w.b2, 0, xnext, H, e->x, H, T, H, F, SHOWO_EPI_RESID_F32, s));
    }
    TRY(showo_layernorm_f32_bf16(e->x, e->fln_w, e->fln_b, e->hf, nullptr, T, H, e->cfg.ln_eps, s));
    TRY(showo_gemm_bf16(e->hf, H, e->wlm, H, e->blm, 0, t->logits, V, nullptr, 0, T, V, H, SHOWO_EPI_F32, s));
    if (logits_out) SHOWO_CHECK_HIP(hipMemcpyAsync(logits_out, t->logits, (size_t)T * V * sizeof(float), hipMemcpyDeviceToDevice, s));
    t->B = B; t->Lq = L;
    t->have_fwd = true;
    t->has_mask = iv != nullptr;
    t->dl_valid = false;
    if (labels) {
        if (t->lw_set) {  // one pass: losses and d(sum_g w_g loss_g)/d(logits)
            TRY(showo_ce_loss(t->logits, V, labels, B, L, V, b_t2i, b_lm, b_mmu, max_seq_len, t->lw[0], t->lw[1], t->lw[2], t->ce_rows,
                              t->counts, t->rowloss, t->dlogits, t->Vp, t->losses, s));
            t->dl_valid = true;
            t->dl_labels = labels;
            for (int k = 0; k < 3; ++k) t->dl_g[k] = t->lw[k];
            t->dl_split[0] = b_t2i; t->dl_split[1] = b_lm; t->dl_split[2] = b_mmu; t->dl_split[3] = max_seq_len;
        } else
        TRY(showo_ce_loss(t->logits, V, labels, B, L, V, b_t2i, b_lm, b_mmu, max_seq_len, 0.f, 0.f, 0.f, t->ce_rows, t->counts,
                          t->rowloss, nullptr, 0, t->losses, s));
        // the backward works on the interval form only: a mask with more than two visibility runs in a row (never produced by
        // the reference's builders with contiguous padding) must not train silently wrong -> NaN losses, no host sync
        if (iv) poison_losses_kernel<<<1, 64, 0, s>>>(t->losses, e->flag);
        if (losses_out) SHOWO_CHECK_HIP(hipMemcpyAsync(losses_out, t->losses, 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    return 0;
}

extern "C" int showo_train_forward(showo_trainer* t, const int64_t* ids, const float* mask, const int64_t* labels, int B, int L,
                                   int b_t2i, int b_lm, int b_mmu, int max_seq_len, float* logits_out, float* losses_out,
                                   void* stream) {
    return train_forward_impl(t, ids, nullptr, mask, labels, B, L, b_t2i, b_lm, b_mmu, max_seq_len, logits_out, losses_out, stream);
}
extern "C" int showo_train_forward_embeds(showo_trainer* t, const float* embeds, const float* mask, const int64_t* labels, int B,
                                          int L, int b_t2i, int b_lm, int b_mmu, int max_seq_len, float* logits_out,
                                          float* losses_out, void* stream) {
    return train_forward_impl(t, nullptr, embeds, mask, labels, B, L, b_t2i, b_lm, b_mmu, max_seq_len, logits_out, losses_out, stream);
}
// d(weighted loss) / d(input embeddings) fp32 [B*L, H] of the last backward (the residual-stream gradient at block 0's input)
extern "C" int showo_train_input_grad(showo_trainer* t, float* out, int64_t n, void* stream) {
    if (!t || !out || !t->have_fwd) return set_error_msg(1, "train_input_grad: run forward + backward first");
    if (n != (int64_t)t->B * t->Lq * t->e->H) return set_error_msg(1, "train_input_grad: size mismatch");
    SHOWO_CHECK_HIP(hipMemcpyAsync(out, t->dy, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

// ---- backward, in three phases so that a data-parallel driver can start the gradient exchange of a finished bucket
// while the next block is still running: head (bucket nL+1) -> blocks nL-1 .. 0 (buckets i+1) -> embedding (bucket 0)
#define BW_PROLOGUE                                                                                                   \
    hipStream_t s = (hipStream_t)stream;                                                                              \
    if (!t || !t->have_fwd) return set_error_msg(1, "train_backward: run showo_train_forward first");                  \
    showo_engine* e = t->e;                                                                                           \
    const int H = e->H, F = e->F, V = e->V, nH = e->nH, B = t->B, L = t->Lq, T = B * L, Vp = t->Vp;                    \
    const int Tp = ((T + 63) / 64) * 64, Lp = ((L + 63) / 64) * 64;                                                   \
    const int32_t* iv = t->has_mask ? e->iv : nullptr;                                                                \
    (void)F; (void)V; (void)nH; (void)Vp; (void)Tp; (void)Lp; (void)iv; (void)s;

extern "C" int showo_train_backward_head(showo_trainer* t, const int64_t* labels, int b_t2i, int b_lm, int b_mmu, int max_seq_len,
                                         float g_t2i, float g_lm, float g_mmu, void* stream) {
    BW_PROLOGUE
    if (!labels) return set_error_msg(1, "train_backward: labels required");
    // ---- loss + head
    const bool have_dl = t->dl_valid && t->dl_labels == labels && t->dl_g[0] == g_t2i && t->dl_g[1] == g_lm && t->dl_g[2] == g_mmu &&
                         t->dl_split[0] == b_t2i && t->dl_split[1] == b_lm && t->dl_split[2] == b_mmu && t->dl_split[3] == max_seq_len;
    t->dl_valid = false;  // dlogits^T below reuses nothing of it, but a second backward must not trust a consumed flag blindly
    if (!have_dl)
        TRY(showo_ce_loss(t->logits, V, labels, B, L, V, b_t2i, b_lm, b_mmu, max_seq_len, g_t2i, g_lm, g_mmu, t->ce_rows, t->counts,
                          t->rowloss, t->dlogits, Vp, nullptr, s));
    if (train_tn()) {
        TRY(showo_colsum_bf16(t->dlogits, Vp, T, Vp, t->colpart, t->gblm, 0, s));                                        // lm_head bias grad
        TRY(showo_gemm_tn_bf16(t->dlogits, Vp, e->hf, H, t->gwlm, H, V, H, T, 0, 1, s));                                    // dWlm [V,H]
    } else {
    TRY(showo_transpose_bf16(t->dlogits, Vp, t->bigT, T, Vp, Tp, 0, t->colpart, t->gblm, 0, s));  // dlogits^T + lm_head bias grad
    TRY(showo_transpose_bf16(e->hf, H, t->xT, T, H, Tp, 0, nullptr, nullptr, 0, s));
    TRY(showo_gemm_bf16(t->bigT, Tp, t->xT, Tp, nullptr, 0, t->gwlm, H, nullptr, 0, V, H, Tp, SHOWO_EPI_F32, s));      // dWlm [V,H]
    }
    TRY(showo_gemm_bf16(t->dlogits, Vp, t->wlmT, Vp, nullptr, 0, t->dh, H, nullptr, 0, T, H, Vp, SHOWO_EPI_F32, s));   // d hf
    SHOWO_CHECK_HIP(hipMemsetAsync(t->dy, 0, (size_t)T * H * sizeof(float), s));
    if (train_tn() && e->nL > 0) {  // + column sums of dy16 = the dense / fc2 bias gradients of the top block
        TRY(showo_ln_bwd_colsum(e->x, e->fln_w, t->dh, t->dy, t->dy, t->dy16, t->lnpart, t->gfln, t->L[e->nL - 1].gb2, T, H, e->cfg.ln_eps, s));
    } else
    TRY(showo_ln_bwd(e->x, e->fln_w, t->dh, t->dy, t->dy, t->dy16, t->lnpart, t->gfln, T, H, e->cfg.ln_eps, s));
    return 0;
}

extern "C" int showo_train_backward_layer(showo_trainer* t, int i, void* stream) {
    BW_PROLOGUE
    if (i < 0 || i >= e->nL) return set_error_msg(1, "train_backward_layer: bad layer index");
    {
        Layer& w = e->layers[i];
        LayerT& l = t->L[i];
        if (train_tn() && l.a) {
            // weight gradients straight from the token-major tensors: dW = dY^T X by showo_gemm_tn_bf16, db = column sums of dY
            // db2 = dbd = column sums of dy16: written into l.gb2 by the LayerNorm backward that produced dy16 (showo_ln_bwd_colsum)
            SHOWO_CHECK_HIP(hipMemcpyAsync(l.gbd, l.gb2, (size_t)H * sizeof(float), hipMemcpyDeviceToDevice, s));
            TRY(showo_gemm_tn_bf16(t->dy16, H, l.a, F, l.gw2, F, H, F, T, 0, 1, s));                                               // dW2 [H,F]
            TRY(showo_gemm_tn_bf16(t->dy16, H, l.attn, H, l.gwd, H, H, H, T, 0, 1, s));                                            // dWd [H,H]
            TRY(showo_gemm_bf16(t->dy16, H, l.w2T, H, nullptr, 0, t->dff, F, nullptr, 0, T, F, H, SHOWO_EPI_BF16, s));          // d a
            TRY(showo_dgelu_colsum_bf16(t->dff, l.f, t->dff, F, T, F, t->colpart, l.gb1, s));                                   // d f, db1
            TRY(showo_gemm_tn_bf16(t->dff, F, l.h, H, l.gw1, H, F, H, T, 0, 1, s));                                                // dW1 [F,H]
            TRY(showo_gemm_bf16(t->dff, F, l.w1T, F, nullptr, 0, t->dh, H, nullptr, 0, T, H, F, SHOWO_EPI_F32, s));             // dh (mlp)
            // attention
            TRY(showo_gemm_bf16(t->dy16, H, l.wdT, H, nullptr, 0, t->d_o, H, nullptr, 0, T, H, H, SHOWO_EPI_BF16, s));          // d o
            TRY(showo_head_transpose(l.Q, t->QT, B, nH, L, Lp, (int64_t)nH * L * 64, (int64_t)L * 64, 64, s));
            TRY(showo_head_transpose(l.K, t->KT, B, nH, L, Lp, (int64_t)nH * L * 64, (int64_t)L * 64, 64, s));
            TRY(showo_attn_bwd(l.Q, l.K, t->QT, t->KT, l.qkv + 2 * H, 3 * H, l.attn, t->d_o, H, t->dOT, l.lse, t->D, iv, nullptr, t->dqk, 2 * H,
                               t->dqk + H, 2 * H, t->dqkv + 2 * H, 3 * H, B, nH, L, Lp, s));
            TRY(showo_qkln_rope_bwd(t->dqk, t->dqk + H, 2 * H, l.qkv, w.qln_w, w.kln_w, e->cosT, e->sinT, t->dqkv, t->qkpart, l.gqk, T, L,
                                    nH, e->cfg.rotary_dim, e->cfg.ln_eps, s));
            TRY(showo_colsum_bf16(t->dqkv, 3 * H, T, 3 * H, t->colpart, l.gbqkv, 0, s));                                        // dbqkv
            TRY(showo_gemm_tn_bf16(t->dqkv, 3 * H, l.h, H, l.gwqkv, H, 3 * H, H, T, 0, 1, s));                                     // dWqkv [3H,H]
            TRY(showo_gemm_bf16(t->dqkv, 3 * H, l.wqkvT, 3 * H, nullptr, 0, t->dh, H, t->dh, H, T, H, 3 * H, SHOWO_EPI_RESID_F32, s));  // dh += attn part
        } else {
        // dy^T (+ bias grads of fc2 and dense: both are column sums of dy)
        TRY(showo_transpose_bf16(t->dy16, H, t->bigT, T, H, Tp, 0, t->colpart, l.gb2, 0, s));
        SHOWO_CHECK_HIP(hipMemcpyAsync(l.gbd, l.gb2, (size_t)H * sizeof(float), hipMemcpyDeviceToDevice, s));
        // MLP
        TRY(showo_transpose_bf16(l.f, F, t->xT, T, F, Tp, 1, nullptr, nullptr, 0, s));                                      // gelu(f)^T
        TRY(showo_gemm_bf16(t->bigT, Tp, t->xT, Tp, nullptr, 0, l.gw2, F, nullptr, 0, H, F, Tp, SHOWO_EPI_F32, s));         // dW2 [H,F]
        TRY(showo_transpose_bf16(l.attn, H, t->xT, T, H, Tp, 0, nullptr, nullptr, 0, s));                                   // attn^T
        TRY(showo_gemm_bf16(t->bigT, Tp, t->xT, Tp, nullptr, 0, l.gwd, H, nullptr, 0, H, H, Tp, SHOWO_EPI_F32, s));         // dWd [H,H]
        TRY(showo_gemm_bf16(t->dy16, H, l.w2T, H, nullptr, 0, t->dff, F, nullptr, 0, T, F, H, SHOWO_EPI_BF16, s));          // d a
        TRY(showo_dgelu_bf16(t->dff, l.f, t->dff, (int64_t)T * F, s));                                                      // d f
        TRY(showo_transpose_bf16(t->dff, F, t->bigT, T, F, Tp, 0, t->colpart, l.gb1, 0, s));                                // df^T, db1
        TRY(showo_transpose_bf16(l.h, H, t->xT, T, H, Tp, 0, nullptr, nullptr, 0, s));                                      // h^T
        TRY(showo_gemm_bf16(t->bigT, Tp, t->xT, Tp, nullptr, 0, l.gw1, H, nullptr, 0, F, H, Tp, SHOWO_EPI_F32, s));         // dW1 [F,H]
        TRY(showo_gemm_bf16(t->dff, F, l.w1T, F, nullptr, 0, t->dh, H, nullptr, 0, T, H, F, SHOWO_EPI_F32, s));             // dh (mlp)
        // attention
        TRY(showo_gemm_bf16(t->dy16, H, l.wdT, H, nullptr, 0, t->d_o, H, nullptr, 0, T, H, H, SHOWO_EPI_BF16, s));          // d o
        TRY(showo_head_transpose(l.Q, t->QT, B, nH, L, Lp, (int64_t)nH * L * 64, (int64_t)L * 64, 64, s));
        TRY(showo_head_transpose(l.K, t->KT, B, nH, L, Lp, (int64_t)nH * L * 64, (int64_t)L * 64, 64, s));
        TRY(showo_attn_bwd(l.Q, l.K, t->QT, t->KT, l.qkv + 2 * H, 3 * H, l.attn, t->d_o, H, t->dOT, l.lse, t->D, iv, nullptr, t->dqk, 2 * H,
                           t->dqk + H, 2 * H, t->dqkv + 2 * H, 3 * H, B, nH, L, Lp, s));
        TRY(showo_qkln_rope_bwd(t->dqk, t->dqk + H, 2 * H, l.qkv, w.qln_w, w.kln_w, e->cosT, e->sinT, t->dqkv, t->qkpart, l.gqk, T, L,
                                nH, e->cfg.rotary_dim, e->cfg.ln_eps, s));
        TRY(showo_transpose_bf16(t->dqkv, 3 * H, t->bigT, T, 3 * H, Tp, 0, t->colpart, l.gbqkv, 0, s));                     // dqkv^T, dbqkv
        TRY(showo_gemm_bf16(t->bigT, Tp, t->xT, Tp, nullptr, 0, l.gwqkv, H, nullptr, 0, 3 * H, H, Tp, SHOWO_EPI_F32, s));   // dWqkv (xT = h^T)
        TRY(showo_gemm_bf16(t->dqkv, 3 * H, l.wqkvT, 3 * H, nullptr, 0, t->dh, H, t->dh, H, T, H, 3 * H, SHOWO_EPI_RESID_F32, s));  // dh += attn part
        }
        // LayerNorm + residual
        if (train_tn() && i > 0 && t->L[i - 1].a) {
            TRY(showo_ln_bwd_colsum(l.x, w.ln_w, t->dh, t->dy, t->dy, t->dy16, t->lnpart, l.gln, t->L[i - 1].gb2, T, H, e->cfg.ln_eps, s));
        } else
        TRY(showo_ln_bwd(l.x, w.ln_w, t->dh, t->dy, t->dy, t->dy16, t->lnpart, l.gln, T, H, e->cfg.ln_eps, s));
        }
    return 0;
}

// Announce the weights of the three losses (training/train.py:600: loss = w_t2i loss_t2i + w_lm loss_lm + w_mmu loss_mmu) BEFORE the
// forward: its cross-entropy pass then also writes d(loss)/d(logits), and showo_train_backward[_head] called with the same labels
// pointer, batch split and weights does not read the logits a second time.  enable = 0 restores the two-pass behaviour.
extern "C" int showo_train_set_loss_weights(showo_trainer* t, float w_t2i, float w_lm, float w_mmu, int enable) {
    if (!t) return set_error_msg(1, "train_set_loss_weights: null handle");
    t->lw_set = enable != 0;
    t->lw[0] = w_t2i; t->lw[1] = w_lm; t->lw[2] = w_mmu;
    t->dl_valid = false;
    return 0;
}

extern "C" int showo_train_backward_embed(showo_trainer* t, void* stream) {
    BW_PROLOGUE
    // ---- embedding
    SHOWO_CHECK_HIP(hipMemsetAsync(t->gembed, 0, (size_t)V * H * sizeof(float), s));
    if (t->from_embeds) return 0;  // the table was not read by this forward; its caller owns d/d(embeddings) (showo_train_input_grad)
    TRY(showo_embed_bwd(t->ids, t->dy, t->gembed, t->order_ws, T, H, V, s));
    return 0;
}

// d(g_t2i loss_t2i + g_lm loss_lm + g_mmu loss_mmu) / d(parameters) of the last showo_train_forward
extern "C" int showo_train_backward(showo_trainer* t, const int64_t* labels, int b_t2i, int b_lm, int b_mmu, int max_seq_len,
                                    float g_t2i, float g_lm, float g_mmu, void* stream) {
    TRY(showo_train_backward_head(t, labels, b_t2i, b_lm, b_mmu, max_seq_len, g_t2i, g_lm, g_mmu, stream));
    for (int i = t->e->nL - 1; i >= 0; --i) TRY(showo_train_backward_layer(t, i, stream));
    return showo_train_backward_embed(t, stream);
}

// bucket b of the flat gradient buffer: 0 = embedding, 1 + i = block i, nL + 1 = head (final LayerNorm, lm_head)
extern "C" int showo_train_bucket(showo_trainer* t, int b, float** ptr, int64_t* n) {
    if (!t || !ptr || !n || b < 0 || b >= (int)t->buckets.size()) return set_error_msg(1, "train_bucket: bad argument");
    *ptr = t->gflat + t->buckets[b].first;
    *n = t->buckets[b].second;
    return 0;
}
extern "C" int showo_train_num_buckets(showo_trainer* t) { return t ? (int)t->buckets.size() : -1; }

extern "C" int showo_train_grad(showo_trainer* t, const char* key, float** ptr, int64_t* n) {
    if (!t || !key || !ptr || !n) return set_error_msg(1, "train_grad: null argument");
    auto it = t->grads.find(key);
    if (it == t->grads.end()) return set_error_msg(3, "train_grad: unknown state-dict key");
    *ptr = it->second.p;
    *n = it->second.n;
    return 0;
}

extern "C" int showo_train_losses(showo_trainer* t, float* out3, void* stream) {
    if (!t || !out3) return set_error_msg(1, "train_losses: null argument");
    SHOWO_CHECK_HIP(hipMemcpyAsync(out3, t->losses, 3 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

extern "C" int showo_train_grad_copy(showo_trainer* t, const char* key, float* dst, int64_t n, void* stream) {
    float* p = nullptr;
    int64_t m = 0;
    TRY(showo_train_grad(t, key, &p, &m));
    if (m != n) return set_error_msg(2, "train_grad_copy: element count mismatch");
    SHOWO_CHECK_HIP(hipMemcpyAsync(dst, p, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

// ---- optimizer: the host registers its fp32 master tensors and moment buffers once; one call then updates everything
extern "C" int showo_train_bind_param(showo_trainer* t, const char* key, float* param, float* exp_avg, float* exp_avg_sq, int64_t n) {
    if (!t || !key || !param || !exp_avg || !exp_avg_sq) return set_error_msg(1, "train_bind_param: null argument");
    auto it = t->grads.find(key);
    if (it == t->grads.end()) return set_error_msg(3, "train_bind_param: unknown state-dict key");
    if (it->second.n != n) return set_error_msg(2, "train_bind_param: element count mismatch");
    std::string k(key);
    // reference rule (training/train.py:211): no weight decay for names containing one of these substrings
    const char* nd[4] = {"bias", "layer_norm.weight", "mlm_ln.weight", "embeddings.weight"};
    bool decay = true;
    for (const char* x : nd) decay = decay && k.find(x) == std::string::npos;
    t->adam_dirty = true;  // the optimizer's segment table is rebuilt by the next step
    for (auto& b : t->bound)
        if (b.key == k) { b.p = param; b.m = exp_avg; b.v = exp_avg_sq; b.n = n; b.decay = decay; return 0; }
    t->bound.push_back(Bound{k, param, exp_avg, exp_avg_sq, n, decay});
    return 0;
}

// torch.optim.AdamW.step() over every bound parameter with the gradients of the last backward, followed by the refresh of
// the engine's bf16 weight images (the transposed images are rebuilt lazily by the next forward)
extern "C" int showo_train_adamw_step(showo_trainer* t, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                      void* stream) {
    if (!t) return set_error_msg(1, "train_adamw_step: null handle");
    if (t->bound.empty()) return set_error_msg(1, "train_adamw_step: no parameters bound");
    // SHOWO_TRAIN_ADAMW_MULTI (default 1): one launch over a segment table instead of one adamw + one image refresh per tensor
    static int multi = -1;
    if (multi < 0) { const char* env = getenv("SHOWO_TRAIN_ADAMW_MULTI"); multi = env ? (atoi(env) != 0) : 1; }
    if (multi) {
        hipStream_t s = (hipStream_t)stream;
        if (t->adam_dirty) {  // (re)build the table: segments = bound tensors with their engine destinations, chunks of ADAM_CHUNK elements
            std::vector<showo::AdamSeg> segs;
            std::vector<int> seg_of;
            std::vector<int64_t> start_of;
            for (auto& b : t->bound) {
                showo::AdamSeg sg{b.p, b.m, b.v, t->grads[b.key].p, nullptr, nullptr, b.n, b.decay ? 1 : 0};
                uint16_t* d16 = nullptr;
                TRY(showo_engine_slot(t->e, b.key.c_str(), b.n, &d16, &sg.dst32));
                sg.dst16 = d16;
                for (int64_t st = 0; st < b.n; st += showo::ADAM_CHUNK) { seg_of.push_back((int)segs.size()); start_of.push_back(st); }
                segs.push_back(sg);
            }
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            hipStreamIsCapturing(s, &cs);
            if (cs != hipStreamCaptureStatusNone) return set_error_msg(7, "train_adamw_step: the segment table must be built outside a stream capture");
            if (t->adam_segs) { hipFree(t->adam_segs); hipFree(t->adam_seg_of); hipFree(t->adam_start_of); }
            SHOWO_CHECK_HIP(hipMalloc(&t->adam_segs, segs.size() * sizeof(showo::AdamSeg)));
            SHOWO_CHECK_HIP(hipMalloc(&t->adam_seg_of, seg_of.size() * sizeof(int)));
            SHOWO_CHECK_HIP(hipMalloc(&t->adam_start_of, start_of.size() * sizeof(int64_t)));
            SHOWO_CHECK_HIP(hipMemcpy(t->adam_segs, segs.data(), segs.size() * sizeof(showo::AdamSeg), hipMemcpyHostToDevice));
            SHOWO_CHECK_HIP(hipMemcpy(t->adam_seg_of, seg_of.data(), seg_of.size() * sizeof(int), hipMemcpyHostToDevice));
            SHOWO_CHECK_HIP(hipMemcpy(t->adam_start_of, start_of.data(), start_of.size() * sizeof(int64_t), hipMemcpyHostToDevice));
            t->adam_chunks = (int)seg_of.size();
            t->adam_dirty = false;
        }
        TRY(showo::adamw_multi_launch(t->adam_segs, t->adam_seg_of, t->adam_start_of, t->adam_chunks, lr, beta1, beta2, eps, weight_decay, step, s));
        TRY(showo_engine_weights_touched(t->e));
        t->weights_synced = false;
        return 0;
    }
    for (auto& b : t->bound) {
        const Grad& g = t->grads[b.key];
        TRY(showo_adamw(b.p, g.p, b.m, b.v, b.n, lr, beta1, beta2, eps, b.decay ? weight_decay : 0.f, step, stream));
        TRY(showo_engine_load(t->e, b.key.c_str(), b.p, b.n, stream));
    }
    t->weights_synced = false;
    return 0;
}
