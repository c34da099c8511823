// Show-o transformer engine: packed bf16 weights + workspaces resident in HBM, whole-module entry points.
// Replaces (reference): PhiForCausalLM/PhiModel forward (models/phi.py:953-1183), Showo.forward without
// labels (models/modeling_showo.py:76-79), Showo.t2i_generate (models/modeling_showo.py:104-181) and the
// incremental form of Showo.mmu_generate (models/modeling_showo.py:183-240; the reference re-runs the whole
// sequence per token, here K/V of earlier rows are cached, which is exact because the mask it grows never
// lets an old row see a new column, modeling_showo.py:203-217).
//
// Data layout in HBM (T = B*L tokens, H hidden, F ffn):
//   x      fp32 [T,H]    residual stream (kept fp32; both GEMM epilogues accumulate into it in place)
//   h      bf16 [T,H]    LayerNorm output (shared by the attention and MLP branches, phi.py:776-790)
//   qkv    bf16 [T,3H]   fused q|k|v projection
//   Q,K    bf16 [B,nH,L,64]; Vt bf16 [B,nH,64,Lp]    head-major operands of the attention kernel
//   attn   bf16 [T,H]; ffn bf16 [T,F]
//   weights: Wqkv [3H,H], Wd [H,H], W1 [F,H], W2 [H,F], Wlm [V,H] bf16 row-major (= nn.Linear layout,
//   K-contiguous, exactly what the MFMA GEMM's operand loader wants); biases / LayerNorm params fp32.
#include "engine.h"
#include "decode_common.h"
#include <cstdio>
#include <vector>
#include <cstring>
#include <set>
#include <string>
#include <vector>

using namespace showo;

namespace {

__global__ void init_cur_kernel(const int64_t* ids, int ld, int img_start, int64_t mask_id, int64_t offset, int64_t* cur, int N,
                                int total) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int b = i / N, n = i - b * N;
    int64_t v = ids[(int64_t)b * ld + img_start + n];
    cur[i] = (v == mask_id) ? mask_id : v - offset;
}
// uncond rows = uncond prefix (first `prefix` tokens) + cond tokens from `prefix` on (modeling_showo.py:137-138)
__global__ void build_ids_kernel(const int64_t* cond, const int64_t* uncond, int64_t* all, int B, int L, int prefix) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * L) return;
    int b = i / L, l = i - b * L;
    int64_t c = cond[i];
    all[i] = c;
    if (uncond) all[(int64_t)(B + b) * L + l] = (l < prefix) ? uncond[i] : c;
}
__global__ void rows_index_kernel(int32_t* rows, int nseq, int L, int img_start, int N) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseq * N) return;
    int b = i / N, n = i - b * N;
    rows[i] = b * L + img_start + n;
}
__global__ void set_iv_kernel(int32_t* iv, int a, int b, int c, int d) {
    if (threadIdx.x == 0) { iv[0] = a; iv[1] = b; iv[2] = c; iv[3] = d; }
}
// ---- t2i prefix reuse: rows [0, prefix) of every sequence (pads + text; causal, never see an image column) are step-invariant
__global__ void gather_ids_kernel(const int64_t* __restrict__ all, int64_t* __restrict__ act, int nseq, int L, int prefix) {
    const int La = L - prefix;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseq * La) return;
    int b = i / La, j = i - b * La;
    act[i] = all[(int64_t)b * L + prefix + j];
}
__global__ void gather_iv_kernel(const int32_t* __restrict__ iv, int32_t* __restrict__ act, int nseq, int L, int prefix) {
    const int La = L - prefix;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseq * La) return;
    int b = i / La, j = i - b * La;
    reinterpret_cast<int4*>(act)[i] = reinterpret_cast<const int4*>(iv)[(int64_t)b * L + prefix + j];
}
// out |= 1 if a prefix row can see a column >= prefix (then the prefix is not step-invariant), out |= 2 if *mask_flag is set
__global__ void prefix_check_kernel(const int32_t* __restrict__ iv, const int32_t* __restrict__ mask_flag, int32_t* __restrict__ out,
                                    int nseq, int L, int prefix) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && mask_flag && *mask_flag) atomicOr(out, 2);
    if (i >= nseq * prefix) return;
    int b = i / prefix, r = i - b * prefix;
    const int4 v = reinterpret_cast<const int4*>(iv)[(int64_t)b * L + r];
    int hi = 0;
    if (v.x < v.y) hi = max(hi, v.y);
    if (v.z < v.w) hi = max(hi, v.w);
    if (hi > prefix) atomicOr(out, 1);
}
// mask row of the token at position *pos: the last prompt row extended by the columns [L0, pos] (modeling_showo.py:203-217)
__global__ void decode_iv_kernel(const int32_t* __restrict__ last_iv, int L0, const int* __restrict__ pos, int32_t* __restrict__ iv) {
    if (threadIdx.x != 0) return;
    int a = last_iv[0], b = last_iv[1], c = last_iv[2], d = last_iv[3];
    const int P = *pos;
    if (b == L0 && a < b) b = P + 1;
    else if (d == L0 && c < d) d = P + 1;
    else if (!(c < d)) { c = L0; d = P + 1; }
    else if (!(a < b)) { a = L0; b = P + 1; }
    iv[0] = a; iv[1] = b; iv[2] = c; iv[3] = d;
}
__global__ void store_token_kernel(const int64_t* __restrict__ tok, int64_t* __restrict__ out, const int* __restrict__ pos, int base) {
    if (threadIdx.x == 0) out[*pos - base] = *tok;
}
// per-call constants of the graph-replayed denoise loop, passed BY VALUE as kernel arguments (no host staging buffer to keep alive,
// no stream synchronisation): sched[0..n) = mask_len_f | temperature per step, hdr = step index | pad | seed lo | seed hi
struct SchedArgs { float v[512]; int hdr[4]; int n; };
__global__ void sched_upload_kernel(SchedArgs a, float* __restrict__ sched, int* __restrict__ hdr) {
    for (int i = threadIdx.x; i < a.n; i += blockDim.x) sched[i] = a.v[i];
    if (threadIdx.x < 4) hdr[threadIdx.x] = a.hdr[threadIdx.x];
}
__global__ void copy_i64_kernel(const int64_t* s, int64_t* d, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = s[i];
}

}  // namespace

#define TRY(expr)            \
    do {                     \
        int _rc = (expr);    \
        if (_rc) return _rc; \
    } while (0)

extern "C" int showo_engine_create(const showo_engine_config* c, showo_engine** out) {
    if (!c || !out) return set_error_msg(1, "engine_create: null argument");
    if (c->hidden % 64 || c->hidden != c->heads * 64) return set_error_msg(1, "engine: head_dim must be 64");
    if (c->ffn % 64) return set_error_msg(1, "engine: ffn must be a multiple of 64");
    showo_engine* e = new showo_engine();
    e->cfg = *c;
    e->H = c->hidden; e->nL = c->layers; e->nH = c->heads; e->F = c->ffn; e->V = c->vocab;
    e->maxT = (int64_t)c->max_batch * c->max_seq;
    const int H = e->H, F = e->F, V = e->V;
    int rc = 0;
    rc |= e->alloc(&e->embed, (int64_t)V * H);
    e->layers.resize(e->nL);
    for (auto& l : e->layers) {
        rc |= e->alloc(&l.wqkv, (int64_t)(3 * H + F) * H); rc |= e->alloc(&l.bqkv, 3 * H + F);
        if (!rc) { l.w1 = l.wqkv + (int64_t)3 * H * H; l.b1 = l.bqkv + 3 * H; }
        rc |= e->alloc(&l.wd, (int64_t)H * H); rc |= e->alloc(&l.bd, H);
        rc |= e->alloc(&l.w2, (int64_t)H * F); rc |= e->alloc(&l.b2, H);
        rc |= e->alloc(&l.wd2, showo_gemm_tiled_elems(H, H + F)); rc |= e->alloc(&l.bd2, H);
        rc |= e->alloc(&l.ln_w, H); rc |= e->alloc(&l.ln_b, H);
        rc |= e->alloc(&l.qln_w, 64); rc |= e->alloc(&l.qln_b, 64);
        rc |= e->alloc(&l.kln_w, 64); rc |= e->alloc(&l.kln_b, 64);
    }
    rc |= e->alloc(&e->fln_w, H); rc |= e->alloc(&e->fln_b, H);
    rc |= e->alloc(&e->wlm, (int64_t)V * H); rc |= e->alloc(&e->blm, V);
    rc |= e->alloc(&e->cosT, (int64_t)c->max_pos * c->rotary_dim);
    rc |= e->alloc(&e->sinT, (int64_t)c->max_pos * c->rotary_dim);
    const int64_t T = e->maxT;
    const int Lp = ((c->max_seq + 63) / 64) * 64;
    rc |= e->alloc(&e->x, T * H); rc |= e->alloc(&e->h, T * H); rc |= e->alloc(&e->qkv, T * 3 * H);
    rc |= e->alloc(&e->Q, T * H); rc |= e->alloc(&e->K, T * H);
    rc |= e->alloc(&e->Vt, (int64_t)c->max_batch * H * Lp);
    rc |= e->alloc(&e->attn, T * H); rc |= e->alloc(&e->ffn, T * F); rc |= e->alloc(&e->hf, (T + 64) * H);  // + 64 rows: the trainer's lm_head weight gradient reads hf up to the next multiple of 64 rows
    rc |= e->alloc(&e->iv, T * 4); rc |= e->alloc(&e->flag, 4); rc |= e->alloc(&e->rows, T);
    rc |= e->alloc(&e->ids_all, T); rc |= e->alloc(&e->cur, T); rc |= e->alloc(&e->sampled, T); rc |= e->alloc(&e->sel, T);
    rc |= e->alloc(&e->iv1, 4); rc |= e->alloc(&e->tok1, 1);
    if (rc) { showo_engine_destroy(e); return rc; }
    hipMemset(e->Vt, 0, (size_t)c->max_batch * H * Lp * sizeof(bf16_t));
    hipMemset(e->flag, 0, 16);
    e->expected = 1 + e->nL * 18 + 2 + 2 + 2;
    *out = e;
    return 0;
}

extern "C" void showo_engine_destroy(showo_engine* e) {
    if (!e) return;
    for (auto& ge : e->t2i_graphs)
        if (ge.exec) hipGraphExecDestroy(ge.exec);
    if (e->ev_pfx) hipEventDestroy(e->ev_pfx);
    if (e->pfx_host) hipHostFree(e->pfx_host);
    showo::engine_batch_free(e);
    for (void* p : e->allocs) hipFree(p);
    delete e;
}

// parity hook: the next forward calls copy the fp32 residual stream into buf [layers + 1, B*L, hidden] (slot 0 = embeddings,
// slot i = output of block i - 1); NULL switches it off.  Lets a test hold EVERY block to the oracle on the GPU's own block input.
extern "C" int showo_engine_set_collect(showo_engine* e, float* buf) {
    if (!e) return set_error_msg(1, "engine: null handle");
    e->collect = buf;
    return 0;
}

extern "C" int showo_engine_t2i_captures(const showo_engine* e) { return e ? e->t2i_captures : -1; }

extern "C" int showo_engine_missing(const showo_engine* e) { return e ? e->expected - (int)e->loaded.size() : -1; }

static int copy_f32(float* dst, const float* src, int64_t n, int64_t expect, hipStream_t s) {
    if (n != expect) return set_error_msg(2, "engine_load: element count mismatch");
    hipError_t e = hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return set_error_hip(e, "hipMemcpyAsync", __FILE__, __LINE__);
    return 0;
}
// GEMM weight: bf16 image; in accuracy mode also the low half (hi + lo = w to 2^-17; hi is the same RNE rounding either way)
// Precision 2: the layer weights become IEEE-half images (saturating RNE); the lm_head keeps its (hi, lo) bf16 pair (head_rows_split).
static int cast_w(showo_engine* e, const std::string& key, bf16_t* dst, bf16_t* dst_lo, const float* src, int64_t n, int64_t expect, hipStream_t s,
                  bool is_head = false) {
    if (n != expect) return set_error_msg(2, "engine_load: element count mismatch");
    if ((e->precision == 1 || (e->precision == 2 && is_head)) && dst_lo) {
        int rc = showo_split_f32_bf16(src, dst, dst_lo, n, s);
        if (!rc) e->lo_loaded.insert(key);
        return rc;
    }
    if (!is_head) e->img_f16 = e->precision == 2;
    return showo_cast_f32_op16(src, dst, n, (e->precision == 2 && !is_head) ? SHOWO_OP_F16 : SHOWO_OP_BF16, s);
}

extern "C" int showo_engine_load(showo_engine* e, const char* key, const float* src, int64_t n, void* stream) {
    if (!e || !key || !src) return set_error_msg(1, "engine_load: null argument");
    hipStream_t s = (hipStream_t)stream;
    const int64_t H = e->H, F = e->F, V = e->V;
    std::string k(key);
    int rc = -1;
    int li = -1;
    char sub[128];
    if (k == "showo.model.embed_tokens.weight") rc = copy_f32(e->embed, src, n, V * H, s);
    else if (k == "showo.model.final_layernorm.weight") rc = copy_f32(e->fln_w, src, n, H, s);
    else if (k == "showo.model.final_layernorm.bias") rc = copy_f32(e->fln_b, src, n, H, s);
    else if (k == "showo.lm_head.weight") rc = cast_w(e, k, e->wlm, e->wlm_lo, src, n, V * H, s, true);
    else if (k == "showo.lm_head.bias") rc = copy_f32(e->blm, src, n, V, s);
    else if (k == "rope.cos") rc = copy_f32(e->cosT, src, n, (int64_t)e->cfg.max_pos * e->cfg.rotary_dim, s);
    else if (k == "rope.sin") rc = copy_f32(e->sinT, src, n, (int64_t)e->cfg.max_pos * e->cfg.rotary_dim, s);
    else if (sscanf(key, "showo.model.layers.%d.%127s", &li, sub) == 2 && li >= 0 && li < e->nL) {
        showo::Layer& l = e->layers[li];
        std::string t(sub);
        if (t == "self_attn.q_proj.weight") rc = cast_w(e, k, l.wqkv, l.wqkv_lo, src, n, H * H, s);
        else if (t == "self_attn.k_proj.weight") rc = cast_w(e, k, l.wqkv + H * H, l.wqkv_lo ? l.wqkv_lo + H * H : nullptr, src, n, H * H, s);
        else if (t == "self_attn.v_proj.weight") rc = cast_w(e, k, l.wqkv + 2 * H * H, l.wqkv_lo ? l.wqkv_lo + 2 * H * H : nullptr, src, n, H * H, s);
        else if (t == "self_attn.q_proj.bias") rc = copy_f32(l.bqkv, src, n, H, s);
        else if (t == "self_attn.k_proj.bias") rc = copy_f32(l.bqkv + H, src, n, H, s);
        else if (t == "self_attn.v_proj.bias") rc = copy_f32(l.bqkv + 2 * H, src, n, H, s);
        else if (t == "self_attn.dense.weight") rc = cast_w(e, k, l.wd, l.wd_lo, src, n, H * H, s);
        else if (t == "self_attn.dense.bias") rc = copy_f32(l.bd, src, n, H, s);
        else if (t == "self_attn.q_layernorm.weight") rc = copy_f32(l.qln_w, src, n, 64, s);
        else if (t == "self_attn.q_layernorm.bias") rc = copy_f32(l.qln_b, src, n, 64, s);
        else if (t == "self_attn.k_layernorm.weight") rc = copy_f32(l.kln_w, src, n, 64, s);
        else if (t == "self_attn.k_layernorm.bias") rc = copy_f32(l.kln_b, src, n, 64, s);
        else if (t == "mlp.fc1.weight") rc = cast_w(e, k, l.w1, l.w1_lo, src, n, F * H, s);
        else if (t == "mlp.fc1.bias") rc = copy_f32(l.b1, src, n, F, s);
        else if (t == "mlp.fc2.weight") rc = cast_w(e, k, l.w2, l.w2_lo, src, n, H * F, s);
        else if (t == "mlp.fc2.bias") rc = copy_f32(l.b2, src, n, H, s);
        else if (t == "input_layernorm.weight") rc = copy_f32(l.ln_w, src, n, H, s);
        else if (t == "input_layernorm.bias") rc = copy_f32(l.ln_b, src, n, H, s);
    }
    if (rc == -1) return set_error_msg(3, "engine_load: unknown state-dict key");
    if (rc == 0) e->loaded.insert(k);
    e->fused_valid = false;  // the fused weight images (and with them a cached t2i graph's preconditions) are rebuilt on the next call
    e->px3_valid = false;
    e->head3_valid = false;
    return rc;
}

// Destination of a state-dict tensor inside the engine, without copying: bf16 weight image OR fp32 vector (exactly one is set).
// Used by the trainer's fused optimizer step, which writes the refreshed images itself (train_engine.hip).
extern "C" int showo_engine_slot(showo_engine* e, const char* key, int64_t n, uint16_t** dst_bf16, float** dst_f32) {
    if (!e || !key || !dst_bf16 || !dst_f32) return set_error_msg(1, "engine_slot: null argument");
    const int64_t H = e->H, F = e->F, V = e->V;
    *dst_bf16 = nullptr; *dst_f32 = nullptr;
    std::string k(key);
    int64_t expect = -1;
    int li = -1;
    char sub[128];
    if (k == "showo.model.embed_tokens.weight") { *dst_f32 = e->embed; expect = V * H; }
    else if (k == "showo.model.final_layernorm.weight") { *dst_f32 = e->fln_w; expect = H; }
    else if (k == "showo.model.final_layernorm.bias") { *dst_f32 = e->fln_b; expect = H; }
    else if (k == "showo.lm_head.weight") { *dst_bf16 = e->wlm; expect = V * H; }
    else if (k == "showo.lm_head.bias") { *dst_f32 = e->blm; expect = V; }
    else if (sscanf(key, "showo.model.layers.%d.%127s", &li, sub) == 2 && li >= 0 && li < e->nL) {
        showo::Layer& l = e->layers[li];
        std::string t(sub);
        if (t == "self_attn.q_proj.weight") { *dst_bf16 = l.wqkv; expect = H * H; }
        else if (t == "self_attn.k_proj.weight") { *dst_bf16 = l.wqkv + H * H; expect = H * H; }
        else if (t == "self_attn.v_proj.weight") { *dst_bf16 = l.wqkv + 2 * H * H; expect = H * H; }
        else if (t == "self_attn.q_proj.bias") { *dst_f32 = l.bqkv; expect = H; }
        else if (t == "self_attn.k_proj.bias") { *dst_f32 = l.bqkv + H; expect = H; }
        else if (t == "self_attn.v_proj.bias") { *dst_f32 = l.bqkv + 2 * H; expect = H; }
        else if (t == "self_attn.dense.weight") { *dst_bf16 = l.wd; expect = H * H; }
        else if (t == "self_attn.dense.bias") { *dst_f32 = l.bd; expect = H; }
        else if (t == "self_attn.q_layernorm.weight") { *dst_f32 = l.qln_w; expect = 64; }
        else if (t == "self_attn.q_layernorm.bias") { *dst_f32 = l.qln_b; expect = 64; }
        else if (t == "self_attn.k_layernorm.weight") { *dst_f32 = l.kln_w; expect = 64; }
        else if (t == "self_attn.k_layernorm.bias") { *dst_f32 = l.kln_b; expect = 64; }
        else if (t == "mlp.fc1.weight") { *dst_bf16 = l.w1; expect = F * H; }
        else if (t == "mlp.fc1.bias") { *dst_f32 = l.b1; expect = F; }
        else if (t == "mlp.fc2.weight") { *dst_bf16 = l.w2; expect = H * F; }
        else if (t == "mlp.fc2.bias") { *dst_f32 = l.b2; expect = H; }
        else if (t == "input_layernorm.weight") { *dst_f32 = l.ln_w; expect = H; }
        else if (t == "input_layernorm.bias") { *dst_f32 = l.ln_b; expect = H; }
    }
    if (expect < 0) return set_error_msg(3, "engine_slot: unknown state-dict key");
    if (expect != n) return set_error_msg(2, "engine_slot: element count mismatch");
    return 0;
}
// the caller rewrote weight images through showo_engine_slot pointers: same bookkeeping as showo_engine_load
extern "C" int showo_engine_weights_touched(showo_engine* e) {
    if (!e) return set_error_msg(1, "engine: null handle");
    e->fused_valid = false;
    e->px3_valid = false;
    e->head3_valid = false;
    e->lo_loaded.clear();  // the hi images were rewritten without their low halves: accuracy mode needs a re-upload
    e->img_f16 = false;    // the trainer writes bf16 images: a precision-2 engine needs a re-upload too (run_layers checks)
    return 0;
}

namespace {
__global__ void add2_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] + b[i];
}
}  // namespace

// SHOWO_W_TILED=0 keeps the fused projections on row-major weights (A/B runs)
static bool w_tiled_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* env = getenv("SHOWO_W_TILED");
        v = env ? (atoi(env) != 0) : 1;
    }
    return v != 0;
}

// Weight images of the two fused projections, rebuilt after any weight load:
//   wd2  = [Wd | W2] ([H, H + F], K-concatenated) for showo_gemm_kcat_bf16, bd2 = bd + b2;
//   wq1t = tiled copy of [Wqkv ; W1] (showo_gemm_tile_weight) -- with SHOWO_W_TILED (default) wd2 is stored tiled as well
static int fused_sync(showo_engine* e, hipStream_t s) {
    if (e->fused_valid) return 0;
    const int H = e->H, F = e->F;
    const bool tiled = w_tiled_enabled();
    if (tiled && !e->wtmp) TRY(e->alloc(&e->wtmp, (int64_t)H * (H + F)));
    for (auto& l : e->layers) {
        bf16_t* cat = tiled ? e->wtmp : l.wd2;
        SHOWO_CHECK_HIP(hipMemcpy2DAsync(cat, (size_t)(H + F) * 2, l.wd, (size_t)H * 2, (size_t)H * 2, H, hipMemcpyDeviceToDevice, s));
        SHOWO_CHECK_HIP(hipMemcpy2DAsync(cat + H, (size_t)(H + F) * 2, l.w2, (size_t)F * 2, (size_t)F * 2, H, hipMemcpyDeviceToDevice, s));
        if (tiled) {
            TRY(showo_gemm_tile_weight(cat, H + F, H, H + F, l.wd2, s));
            if (!l.wq1t) TRY(e->alloc(&l.wq1t, showo_gemm_tiled_elems(3 * H + F, H)));
            TRY(showo_gemm_tile_weight(l.wqkv, H, 3 * H + F, H, l.wq1t, s));
        }
        add2_f32_kernel<<<dim3((H + 255) / 256), dim3(256), 0, s>>>(l.bd, l.b2, l.bd2, H);
    }
    SHOWO_CHECK_HIP(hipGetLastError());
    e->fused_tiled = tiled;
    e->fused_valid = true;
    return 0;
}

// SHOWO_FUSED_LAYER=0 restores the four-GEMM layer (A/B runs)
static bool fused_layer_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* env = getenv("SHOWO_FUSED_LAYER");
        v = env ? (atoi(env) != 0) : 1;
    }
    return v != 0;
}

// ---- the 24-layer stack.  K / V^T go to the per-call workspace (layer stride 0) or to a per-layer cache:
//      the AR decode cache (one sequence) or the t2i prefix-reuse cache (whole batch).
struct KVDest {
    bf16_t *k, *vt;
    int64_t k_lstride, v_lstride;  // elements between consecutive layers
    int Lcap, Lp;                  // key rows per head in k, columns per row in vt
    bf16_t *k_lo = nullptr, *vt_lo = nullptr;  // accuracy mode: low halves (same layout and strides)
};
static KVDest kv_workspace(showo_engine* e, int L) { return KVDest{e->K, e->Vt, 0, 0, L, ((L + 63) / 64) * 64, e->p_Klo, e->p_Vtlo}; }
namespace showo { int g_decode_impl = 0; }
static int g_decode_chain = 0;  // 1: the fused layer as a plain three-launch chain (no co-scheduled fc2 role)
extern "C" int showo_decode_set_impl(int impl) {
    if (impl < 0 || impl > 2) return set_error_msg(1, "decode_set_impl: 0 = fused layer (default), 1 = unfused, 2 = fused chain without the co-scheduled fc2 role");
    showo::g_decode_impl = impl == 1;
    g_decode_chain = impl == 2;
    return 0;
}

static KVDest kv_decode_cache(showo_engine* e) {
    return KVDest{e->kcache, e->vtcache, (int64_t)e->nH * e->cache_cap * 64, (int64_t)e->nH * 64 * e->cache_cap, e->cache_cap, e->cache_cap,
                  e->kcache_lo, e->vtcache_lo};
}

// parity hook (showo_engine_set_collect): copy the fp32 residual stream after layer `slot - 1` (slot 0 = the embeddings)
static int collect_x(showo_engine* e, int slot, int T, hipStream_t s) {
    if (!e->collect) return 0;
    SHOWO_CHECK_HIP(hipMemcpyAsync(e->collect + (int64_t)slot * T * e->H, e->x, (size_t)T * e->H * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}

// ---- Infinity-Cache prefetch plan of the decode layers (decode_common.h, prefetch_role) -------------------------------------------
static int g_pf_mb = -1, g_pf_dense = 0, g_pf_blocks = 64;  // off by default (measured negative: profiles/r5_decode_dot2_sweep.txt)
static void decode_prefetch_env() {
    if (g_pf_mb >= 0) return;
    const char* m = getenv("SHOWO_DECODE_PF_MB");
    g_pf_mb = m ? atoi(m) : 0;
    if (g_pf_mb < 0) g_pf_mb = 0;
    const char* d = getenv("SHOWO_DECODE_PF_DENSE");
    if (d) g_pf_dense = atoi(d) != 0;
    const char* b = getenv("SHOWO_DECODE_PF_BLOCKS");
    if (b && atoi(b) > 0) g_pf_blocks = atoi(b);
}
extern "C" int showo_decode_set_prefetch(int next_mb, int dense, int blocks) {
    if (next_mb < 0 || blocks < 0 || blocks > 1024) return set_error_msg(1, "showo_decode_set_prefetch: next_mb >= 0, 0 <= blocks <= 1024");
    decode_prefetch_env();
    g_pf_mb = next_mb;
    g_pf_dense = dense != 0;
    g_pf_blocks = blocks;
    return 0;
}
namespace showo {
DecodeTuning& decode_tuning() {
    static DecodeTuning t;
    static bool init = false;
    if (!init) {
        init = true;
        auto env = [](const char* name, int* v) { const char* e = getenv(name); if (e && atoi(e) > 0) *v = atoi(e); };
        env("SHOWO_DECODE_CO_BLOCKS", &t.co_blocks);
        env("SHOWO_DECODE_BATCH_CO_BLOCKS", &t.batch_co_blocks);
        env("SHOWO_DECODE_BATCH_LN_BLOCKS", &t.batch_ln_blocks);
        env("SHOWO_DECODE_LN_BLOCKS", &t.ln_blocks);
        env("SHOWO_DECODE_OUT_BLOCKS", &t.out_blocks);
    }
    return t;
}
}  // namespace showo
extern "C" int showo_decode_set_tuning(const char* name, int value) {
    if (!name || value < 1 || value > 65535) return set_error_msg(1, "showo_decode_set_tuning: name and 1 <= value <= 65535");
    showo::DecodeTuning& t = showo::decode_tuning();
    const std::string n(name);
    if (n == "co_blocks") t.co_blocks = value;
    else if (n == "batch_co_blocks") t.batch_co_blocks = value;
    else if (n == "batch_ln_blocks") t.batch_ln_blocks = value;
    else if (n == "ln_blocks") t.ln_blocks = value;
    else if (n == "out_blocks") t.out_blocks = value;
    else return set_error_msg(1, "showo_decode_set_tuning: unknown knob");
    return 0;
}
namespace showo {
void decode_prefetch_plan(const ::showo_engine* e, int li, DecodePrefetch* pf) {
    decode_prefetch_env();
    *pf = DecodePrefetch{};
    if ((g_pf_mb == 0 && !g_pf_dense) || g_pf_blocks == 0) return;
    const int64_t H = e->H, F = e->F;
    int n = 0;
    if (g_pf_dense) { pf->p[n] = e->layers[li].wd; pf->bytes[n] = H * H * 2; ++n; }
    int64_t left = (int64_t)g_pf_mb << 20;
    auto add = [&](const void* p, int64_t bytes) {
        if (left <= 0 || !p) return;
        const int64_t b = (bytes < left ? bytes : left) & ~(int64_t)15;
        pf->p[n] = p; pf->bytes[n] = b; ++n;
        left -= b;
    };
    if (li + 1 < e->nL) {
        add(e->layers[li + 1].wqkv, 3 * H * H * 2);
        if (n < 3) add(e->layers[li + 1].w1, F * H * 2);
    } else {
        add(e->wlm, (int64_t)e->V * H * 2);
    }
    pf->blocks = n ? g_pf_blocks : 0;
}
}  // namespace showo


// ---- accuracy mode ----------------------------------------------------------------------------------------------------------------
// Two implementations.  (1) precise.hip (run_layers_precise): split-bf16 GEMMs on the 128^2 kernel, everything between them in fp32 on
// the vector ALU: any shape, slow.  (2) run_layers_precise_fast (round 5): the PRODUCTION kernels on K-concatenated split images.  A
// split-precision product a w = a_hi w_hi + a_lo w_hi + a_hi w_lo is ONE bf16 GEMM over K' = 3K between A' = [a_hi | a_lo | a_hi] and
// W' = [w_hi | w_hi | w_lo] with the fp32 accumulation the MFMA does anyway, so gemm2p / gemm3w run unchanged (tile tuner, tiled
// weights, split-K, the K-concatenated residual form); only the fused projection epilogue (epilogue_qkv_split: outputs as (hi, lo)
// pairs, IEEE gelu) and the attention (attn_lds_body<SPLIT>: three MFMAs per fragment pair) have accuracy-mode forms.  A layer is
// the same two GEMM launches + attention as in precision 0, so prefix reuse, hipGraph replay and the KV cache all apply.
static int g_precise_fast = -1;  // -1: read SHOWO_PRECISE_FAST once (default on); 0 / 1: showo_precise_set_fast
extern "C" int showo_precise_set_fast(int on) { g_precise_fast = on ? 1 : 0; return 0; }
static bool precise_fast_shape_ok(const showo_engine* e) {
    if (g_precise_fast < 0) { const char* env = getenv("SHOWO_PRECISE_FAST"); g_precise_fast = env ? (atoi(env) != 0) : 1; }
    return g_precise_fast && e->cfg.rotary_dim == 32 && (3 * e->H) % 256 == 0 && e->H % 64 == 0 && e->F % 64 == 0 && e->H % 4 == 0;
}
// ... and its images / workspaces exist (showo_engine_set_precision(e, 1) made them)
static bool precise_fast_ok(const showo_engine* e) { return precise_fast_shape_ok(e) && e->p_h3 != nullptr; }
extern "C" int showo_engine_set_precision(showo_engine* e, int precision) {
    if (!e) return set_error_msg(1, "engine: null handle");
    if (precision < 0 || precision > 2) return set_error_msg(1, "engine_set_precision: 0 = bf16 operands, 1 = split bf16 (fp32-class), 2 = fp16 operands");
    if (precision == 2 && !e->wlm3) {  // the lm_head of precision 2 is a split-bf16 product: its low half, [hi | hi | lo] image and LayerNorm workspace
        const int64_t H = e->H, V = e->V, T = e->maxT;
        int rc = 0;
        if (!e->wlm_lo) rc |= e->alloc(&e->wlm_lo, V * H);
        if (!e->p_hf3) rc |= e->alloc(&e->p_hf3, (T + 256) * 3 * H);
        if (!rc) rc |= e->alloc(&e->wlm3, V * 3 * H);
        if (rc) { e->release(&e->wlm3); return rc; }
        e->head3_valid = false;
    }
    if ((precision == 2) != (e->precision == 2)) {
        // the layer weight images change their element type: un-load every GEMM weight so that the host uploads them again
        // (showo_engine_missing() > 0 until it has); cached t2i graphs are keyed on the precision and simply miss
        for (auto it = e->loaded.begin(); it != e->loaded.end();) {
            const std::string& k = *it;
            const bool gemm_w = k.size() > 7 && k.compare(k.size() - 7, 7, ".weight") == 0 &&
                                (k.find("_proj.") != std::string::npos || k.find(".dense.") != std::string::npos || k.find(".fc1.") != std::string::npos ||
                                 k.find(".fc2.") != std::string::npos || k == "showo.lm_head.weight");
            if (gemm_w) it = e->loaded.erase(it); else ++it;
        }
        e->lo_loaded.clear();
        e->fused_valid = false; e->px3_valid = false; e->head3_valid = false;
    }
    if (precision == 1 && !e->p_hlo) {  // low halves of every GEMM weight + fp32 workspaces, allocated on first use
        const int64_t H = e->H, F = e->F, V = e->V, T = e->maxT;
        int rc = 0;
        for (auto& l : e->layers) {
            rc |= e->alloc(&l.wqkv_lo, (3 * H + F) * H);
            if (!rc) l.w1_lo = l.wqkv_lo + 3 * H * H;
            rc |= e->alloc(&l.wd_lo, H * H);
            rc |= e->alloc(&l.w2_lo, H * F);
        }
        rc |= e->alloc(&e->p_hlo, T * H); rc |= e->alloc(&e->p_actlo, T * F);
        rc |= e->alloc(&e->p_qkv, T * 3 * H); rc |= e->alloc(&e->p_f, T * F);
        rc |= e->alloc(&e->p_Q, T * H); rc |= e->alloc(&e->p_K, T * H); rc |= e->alloc(&e->p_V, T * H); rc |= e->alloc(&e->p_a, T * H);
        if (!e->wlm_lo) rc |= e->alloc(&e->wlm_lo, V * H);
        if (rc) return rc;
    }
    if (precision == 1 && precise_fast_shape_ok(e) && !e->p_h3) {  // K-concatenated split images + workspaces of the production-kernel path
        const int64_t H = e->H, F = e->F, V = e->V, T = e->maxT, HF = H + F;
        const int64_t Lp = ((e->cfg.max_seq + 63) / 64) * 64;
        int rc = 0;
        for (auto& l : e->layers) {
            rc |= e->alloc(&l.wq1x3, showo_gemm_tiled_elems((int)(3 * H + F), (int)(3 * H)));
            rc |= e->alloc(&l.wd2x3, showo_gemm_tiled_elems((int)H, (int)(3 * HF)));
        }
        const int64_t t1 = (3 * H + F) * 3 * H, t2 = H * 3 * HF;
        rc |= e->alloc(&e->wtmp3, t1 > t2 ? t1 : t2);
        rc |= e->alloc(&e->p_h3, (T + 256) * 3 * H);
        if (!e->p_hf3) rc |= e->alloc(&e->p_hf3, (T + 256) * 3 * H);
        rc |= e->alloc(&e->p_act, (T + 256) * 2 * HF);
        rc |= e->alloc(&e->p_Qlo, T * H); rc |= e->alloc(&e->p_Klo, T * H);
        rc |= e->alloc(&e->p_Vtlo, (int64_t)e->cfg.max_batch * H * Lp);
        if (!e->wlm3) rc |= e->alloc(&e->wlm3, V * 3 * H);
        if (rc) {  // a partial set must not survive (wlm3 == NULL would re-allocate everything at the next call)
            for (auto& l : e->layers) { e->release(&l.wq1x3); e->release(&l.wd2x3); }
            e->release(&e->wtmp3); e->release(&e->p_h3); e->release(&e->p_act); e->release(&e->p_Qlo); e->release(&e->p_Klo);
            e->release(&e->p_Vtlo);
            return rc;
        }
        hipMemset(e->p_Vtlo, 0, (size_t)e->cfg.max_batch * H * Lp * sizeof(bf16_t));
        e->px3_valid = false;
    }
    e->precision = precision;
    return 0;
}
// 1 when accuracy mode runs on the production kernels (K-concatenated split images): prefix reuse, hipGraph replay and the KV-cached
// decode are then available in precision 1 as well; 0: the fp32 reference kernels of precise.hip (small / odd shapes, SHOWO_PRECISE_FAST=0)
extern "C" int showo_engine_precise_fast(const showo_engine* e) { return e && precise_fast_ok(e) ? 1 : 0; }
// 1 when every GEMM weight has a current low half (q, k, v, dense, fc1, fc2 per layer + lm_head), i.e. precision 1 can run
extern "C" int showo_engine_precise_ready(const showo_engine* e) {
    return e && e->wlm_lo && e->p_hlo && (int)e->lo_loaded.size() == e->nL * 6 + 1;
}
extern "C" int showo_engine_get_precision(const showo_engine* e) { return e ? e->precision : -1; }

static int precise_check(showo_engine* e) {
    if (!showo_engine_precise_ready(e))
        return set_error_msg(4, "engine (precision 1): the low halves of the weights are missing or stale (weights were loaded before "
                                "showo_engine_set_precision(e, 1), or rewritten by the trainer): upload the weights again");
    return 0;
}

// One pass over the layer stack in accuracy mode.  Same operation order as models/phi.py:774-790; the two residual adds land as
// (x + dense(attn)) + fc2(gelu(fc1(h))) (fp32 either way).  K / V of the call live in the fp32 workspace: no KV cache in this mode.
static int run_layers_precise(showo_engine* e, int B, int L, const int32_t* iv, const int32_t* flag, const float* dense, hipStream_t s) {
    const int H = e->H, F = e->F, nH = e->nH, T = B * L;
    TRY(precise_check(e));
    if (e->cfg.rotary_dim != 32) return set_error_msg(1, "engine (precision 1): rotary_dim 32 only");
    TRY(collect_x(e, 0, T, s));
    for (int li = 0; li < e->nL; ++li) {
        showo::Layer& l = e->layers[li];
        TRY(showo::precise_ln_split(e->x, l.ln_w, l.ln_b, nullptr, e->h, e->p_hlo, T, H, e->cfg.ln_eps, s));
        TRY(showo_gemm_bf16x3(e->h, e->p_hlo, H, l.wqkv, l.wqkv_lo, H, l.bqkv, 0, e->p_qkv, 3 * H, nullptr, 0, T, 3 * H, H, s));
        TRY(showo::precise_qk_prep(e->p_qkv, l.qln_w, l.qln_b, l.kln_w, l.kln_b, e->cosT, e->sinT, e->p_Q, e->p_K, e->p_V, B, L, nH,
                                   e->cfg.ln_eps, 0, L, s));
        TRY(showo::precise_attention(e->p_Q, e->p_K, e->p_V, iv, flag, dense, e->p_a, B, nH, L, L, L, H, s));
        TRY(showo_split_f32_bf16(e->p_a, e->attn, e->p_actlo, (int64_t)T * H, s));
        TRY(showo_gemm_bf16x3(e->attn, e->p_actlo, H, l.wd, l.wd_lo, H, l.bd, 0, e->x, H, e->x, H, T, H, H, s));
        TRY(showo_gemm_bf16x3(e->h, e->p_hlo, H, l.w1, l.w1_lo, H, l.b1, 0, e->p_f, F, nullptr, 0, T, F, H, s));
        TRY(showo::precise_gelu_split(e->p_f, e->ffn, e->p_actlo, (int64_t)T * F, s));
        TRY(showo_gemm_bf16x3(e->ffn, e->p_actlo, F, l.w2, l.w2_lo, F, l.b2, 0, e->x, H, e->x, H, T, H, F, s));
        TRY(collect_x(e, li + 1, T, s));
    }
    return 0;
}
static int head_rows_precise(showo_engine* e, const int32_t* rows, int nrows, int col0, int ncols, float* logits, hipStream_t s) {
    TRY(precise_check(e));
    TRY(showo::precise_ln_split(e->x, e->fln_w, e->fln_b, rows, e->hf, e->p_hlo, nrows, e->H, e->cfg.ln_eps, s));
    return showo_gemm_bf16x3(e->hf, e->p_hlo, e->H, e->wlm + (int64_t)col0 * e->H, e->wlm_lo + (int64_t)col0 * e->H, e->H, e->blm + col0, 0,
                             logits, ncols, nullptr, 0, nrows, ncols, e->H, s);
}

static int fused_sync(showo_engine* e, hipStream_t s);
// [hi | hi | lo] rows of the lm_head (precision 1 and 2: the head is a split-bf16 product in both); never inside a capture
static int head3_sync(showo_engine* e, hipStream_t s) {
    if (e->head3_valid) return 0;
    if (!e->wlm3 || !e->wlm_lo || !e->lo_loaded.count("showo.lm_head.weight"))
        return set_error_msg(4, "engine: the low half of the lm_head is missing or stale (the weights were loaded before "
                                "showo_engine_set_precision(e, 1 | 2), or rewritten by the trainer): upload the weights again");
    const int64_t H = e->H, V = e->V;
    SHOWO_CHECK_HIP(hipMemcpy2DAsync(e->wlm3, (size_t)3 * H * 2, e->wlm, (size_t)H * 2, (size_t)H * 2, (size_t)V, hipMemcpyDeviceToDevice, s));
    SHOWO_CHECK_HIP(hipMemcpy2DAsync(e->wlm3 + H, (size_t)3 * H * 2, e->wlm, (size_t)H * 2, (size_t)H * 2, (size_t)V, hipMemcpyDeviceToDevice, s));
    SHOWO_CHECK_HIP(hipMemcpy2DAsync(e->wlm3 + 2 * H, (size_t)3 * H * 2, e->wlm_lo, (size_t)H * 2, (size_t)H * 2, (size_t)V, hipMemcpyDeviceToDevice, s));
    e->head3_valid = true;
    return 0;
}
// K-concatenated split images of every GEMM weight (see "accuracy mode" above), rebuilt after any weight load; never inside a capture
static int precise_sync(showo_engine* e, hipStream_t s) {
    if (e->px3_valid) return 0;
    TRY(precise_check(e));
    TRY(fused_sync(e, s));  // bd2 = bd + b2
    const int64_t H = e->H, F = e->F, V = e->V, HF = H + F;
    auto cp = [&](bf16_t* dst, int64_t ldd, const bf16_t* src, int64_t lds, int64_t cols, int64_t rows) -> int {
        SHOWO_CHECK_HIP(hipMemcpy2DAsync(dst, (size_t)ldd * 2, src, (size_t)lds * 2, (size_t)cols * 2, (size_t)rows, hipMemcpyDeviceToDevice, s));
        return 0;
    };
    for (auto& l : e->layers) {
        bf16_t* cat = e->wtmp3;
        // [Wqkv ; W1] rows -> [w_hi | w_hi | w_lo]  (wqkv / w1 and their low halves are one allocation each)
        TRY(cp(cat, 3 * H, l.wqkv, H, H, 3 * H + F));
        TRY(cp(cat + H, 3 * H, l.wqkv, H, H, 3 * H + F));
        TRY(cp(cat + 2 * H, 3 * H, l.wqkv_lo, H, H, 3 * H + F));
        TRY(showo_gemm_tile_weight(cat, (int)(3 * H), (int)(3 * H + F), (int)(3 * H), l.wq1x3, s));
        // [Wd | W2] rows -> [Wd_hi | W2_hi | Wd_hi | W2_hi | Wd_lo | W2_lo] against act = [attn_hi | ffn_hi | attn_lo | ffn_lo | attn_hi | ffn_hi]
        for (int rep = 0; rep < 3; ++rep) {
            TRY(cp(cat + rep * HF, 3 * HF, rep == 2 ? l.wd_lo : l.wd, H, H, H));
            TRY(cp(cat + rep * HF + H, 3 * HF, rep == 2 ? l.w2_lo : l.w2, F, F, H));
        }
        TRY(showo_gemm_tile_weight(cat, (int)(3 * HF), (int)H, (int)(3 * HF), l.wd2x3, s));
    }
    TRY(head3_sync(e, s));
    e->px3_valid = true;
    return 0;
}

// One pass over the layer stack in accuracy mode on the production kernels: LayerNorm -> [hi | lo | hi]; ONE [Wqkv ; W1] projection
// (K' = 3H) whose epilogue writes Q / K / V^T and gelu(fc1) as (hi, lo) pairs; split attention; ONE K-concatenated residual GEMM
// (K' = 3 (H + F)) over act = [attn_hi | ffn_hi | attn_lo | ffn_lo] read as [act | act[:, :H + F]].  Any T >= 1, any KV destination.
static int run_layers_precise_fast(showo_engine* e, int B, int L, int pos0, const KVDest& kv, const int32_t* iv, const int32_t* flag,
                                   const float* dense, hipStream_t s) {
    const int H = e->H, F = e->F, nH = e->nH, T = B * L, HF = H + F, lda = 2 * HF;
    if (!kv.k_lo || !kv.vt_lo) return set_error_msg(7, "engine (precision 1): the low-half K / V^T destination is missing");
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    hipStreamIsCapturing(s, &cs);
    if (!e->px3_valid && cs != hipStreamCaptureStatusNone) return set_error_msg(7, "engine (precision 1): split weight images must be built before a stream capture");
    TRY(precise_sync(e, s));
    TRY(collect_x(e, 0, T, s));
    const int Lk = pos0 + L;
    for (int li = 0; li < e->nL; ++li) {
        showo::Layer& l = e->layers[li];
        bf16_t* Kd = kv.k + li * kv.k_lstride;
        bf16_t* Vd = kv.vt + li * kv.v_lstride;
        bf16_t* Kl = kv.k_lo + li * kv.k_lstride;
        bf16_t* Vl = kv.vt_lo + li * kv.v_lstride;
        TRY(showo::precise_ln_split3(e->x, l.ln_w, l.ln_b, nullptr, e->p_h3, T, H, e->cfg.ln_eps, s));
        TRY(showo_gemm_qkv_fc1_split(e->p_h3, 3 * H, l.wq1x3, 3 * H, 3 * H, l.bqkv, l.qln_w, l.qln_b, l.kln_w, l.kln_b, e->cosT, e->sinT,
                                     e->Q, e->p_Qlo, Kd, Kl, Vd, Vl, e->p_act + H, e->p_act + HF + H, lda, F, B, L, nH, e->cfg.rotary_dim,
                                     e->cfg.ln_eps, pos0, kv.Lcap, kv.Lp, 1, s));
        TRY(showo_attn_fwd_split(e->Q, e->p_Qlo, Kd, Kl, Vd, Vl, iv, flag, dense, e->p_act, e->p_act + HF, B, nH, L, Lk, kv.Lcap, kv.Lp, lda, s));
        TRY(showo_gemm_kcat_bf16(e->p_act, lda, 2 * HF, e->p_act, lda, HF, l.wd2x3, 3 * HF, l.bd2, e->x, H, e->x, H, T, H,
                                 SHOWO_EPI_RESID_F32, 1, s));
        TRY(collect_x(e, li + 1, T, s));
    }
    return 0;
}
// final LayerNorm -> [hi | lo | hi], lm_head rows [hi | hi | lo]: ONE bf16 GEMM over 3H (precision 1 on the production kernels, and the
// head of precision 2: final hidden state and lm_head weight are then not rounding points at all, oracle/predict_rounding.py)
static int head_rows_precise_fast(showo_engine* e, const int32_t* rows, int nrows, int col0, int ncols, float* logits, hipStream_t s) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    hipStreamIsCapturing(s, &cs);
    if (e->precision == 2) {
        if (!e->head3_valid && cs != hipStreamCaptureStatusNone) return set_error_msg(7, "engine (precision 2): the lm_head image must be built before a stream capture");
        TRY(head3_sync(e, s));
    } else {
        if (!e->px3_valid && cs != hipStreamCaptureStatusNone) return set_error_msg(7, "engine (precision 1): split weight images must be built before a stream capture");
        TRY(precise_sync(e, s));
    }
    const int H = e->H;
    TRY(showo::precise_ln_split3(e->x, e->fln_w, e->fln_b, rows, e->p_hf3, nrows, H, e->cfg.ln_eps, s));
    return showo_gemm_bf16(e->p_hf3, 3 * H, e->wlm3 + (int64_t)col0 * 3 * H, 3 * H, e->blm + col0, 0, logits, ncols, nullptr, 0, nrows, ncols,
                           3 * H, SHOWO_EPI_F32, s);
}

// precision 2 range check (showo_engine_set_range_check): count the elements of a freshly written 16-bit activation that sit at the
// fp16 saturation value (or are inf / NaN).  Off (one pointer test) unless a counter is registered; never inside the timed path.
static int range_check(showo_engine* e, const bf16_t* x, int64_t n, hipStream_t s) {
    if (!e->range_count || e->precision != 2) return 0;
    return showo_count_f16_saturated(x, n, e->range_count, s);
}
extern "C" int showo_engine_set_range_check(showo_engine* e, int64_t* count) {
    if (!e) return set_error_msg(1, "engine: null handle");
    e->range_count = count;
    return 0;
}

static int run_layers(showo_engine* e, int B, int L, int pos0, const KVDest& kv, const int32_t* iv, const int32_t* flag,
                      const float* dense, hipStream_t s) {
    const int H = e->H, F = e->F, nH = e->nH;
    const int T = B * L;
    if (e->precision == 1) {
        if (precise_fast_ok(e)) return run_layers_precise_fast(e, B, L, pos0, kv, iv, flag, dense, s);
        if (pos0 != 0 || kv.k != e->K) return set_error_msg(1, "engine (precision 1): KV-cached calls need the production-kernel form of accuracy mode (showo_engine_precise_fast)");
        return run_layers_precise(e, B, L, iv, flag, dense, s);
    }
    // precision 2: the same launches with IEEE-half operands (op = SHOWO_OP_F16 on every 16-bit operand, weight image and activation)
    const int op = e->precision == 2 ? SHOWO_OP_F16 : SHOWO_OP_BF16;
    if ((op == SHOWO_OP_F16) != e->img_f16)
        return set_error_msg(4, "engine: the weight images hold the other 16-bit type (loaded under another precision, or rewritten by the "
                                "trainer): upload the weights again");
    TRY(collect_x(e, 0, T, s));
    const int Lk = pos0 + L;
    const int Lcap = kv.Lcap, Lp = kv.Lp;
    // (the fused three-launch decode layer of decode.hip has instances for both operand types since round 6)
    if (T == 1 && showo::g_decode_impl == 0 && !dense && showo::decode_fused_shapes_ok(H, F) && (size_t)Lcap * 4 + 2048 <= 60000) {
        // AR decode step (decode.hip): three launches per layer.  Co-scheduled layer (default): the attention launch carries extra
        // blocks that stream the fc2 weights into y2 (attention.hip, attn_decode_co_kernel) -- fc2 does not depend on the attention
        // (Phi's block is parallel-residual) -- and the third launch adds dense(attn) and y2:
        //   LN + qkv GEMV + fc1 GEMV  ->  [ attention (32 blocks) || fc2 GEMV -> y2 (co_blocks) ]  ->  dense GEMV + both residual adds
        // Same arithmetic and parenthesisation as the plain chain (x = (x + (dense + bd)) + (fc2 + b2)): bit-identical results.
        // Measured (gpurun_out/bench_mmu_r2p_*, one box, cfg4): chain 857-861 tokens/s; co-scheduled 922-928 with 224 fc2 blocks, 965
        // with 128, 969 with 96.  (A stream fork / join of the same overlap was SLOWER than the chain, 705-711 vs 843-846 tokens/s --
        // every fork / join edge of the per-token graph costs more than the overlap buys -- and left the library in round 6; so did
        // folding the third launch into the second behind an in-launch hand-off, profiles/r6_decode_merge_ab.txt.)
        // SHOWO_DECODE_FORK=0 / showo_decode_set_impl(2): the plain chain.
        static int co_on = -1;
        if (co_on < 0) {
            const char* env = getenv("SHOWO_DECODE_FORK");
            co_on = env ? (atoi(env) != 0) : 1;
        }
        const int co_blocks = showo::decode_tuning().co_blocks;
        const bool co = co_on && F == 8192 && !g_decode_chain;
        if (co && !e->y2) TRY(e->alloc(&e->y2, H));
        for (int li = 0; li < e->nL; ++li) {
            showo::Layer& l = e->layers[li];
            bf16_t* Kd = kv.k + li * kv.k_lstride;
            bf16_t* Vd = kv.vt + li * kv.v_lstride;
            if (co) {
                TRY(showo::decode_ln_gemv2(e->x, l.ln_w, l.ln_b, e->cfg.ln_eps, H, l.wqkv, l.bqkv, e->qkv, nullptr, 3 * H, l.w1, l.b1,
                                           e->ffn, F, s, op));
                showo::DecodePrefetch pf;
                showo::decode_prefetch_plan(e, li, &pf);
                TRY(showo::attn_decode_fused(e->qkv, l.qln_w, l.qln_b, l.kln_w, l.kln_b, e->cosT, e->sinT, Kd, Vd, iv, e->attn, nH,
                                             e->cfg.rotary_dim, e->cfg.ln_eps, pos0, Lcap, Lp, s, l.w2, e->ffn, l.b2, F, H, e->y2, co_blocks, &pf, op));
                TRY(showo::decode_out_gemv2(e->x, l.wd, e->attn, l.bd, H, l.w2, e->ffn, l.b2, F, H, s, 2, e->y2, op));
                continue;
            }
            TRY(showo::decode_ln_gemv2(e->x, l.ln_w, l.ln_b, e->cfg.ln_eps, H, l.wqkv, l.bqkv, e->qkv, nullptr, 3 * H, l.w1, l.b1,
                                       e->ffn, F, s, op));
            TRY(showo::attn_decode_fused(e->qkv, l.qln_w, l.qln_b, l.kln_w, l.kln_b, e->cosT, e->sinT, Kd, Vd, iv, e->attn, nH,
                                         e->cfg.rotary_dim, e->cfg.ln_eps, pos0, Lcap, Lp, s, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, nullptr, op));
            TRY(showo::decode_out_gemv2(e->x, l.wd, e->attn, l.bd, H, l.w2, e->ffn, l.b2, F, H, s, 0, nullptr, op));
        }
        return 0;
    }
    // Prefill / t2i layers: TWO GEMM launches.  Phi's block is parallel-residual (phi.py:774-790): q/k/v_proj and fc1 read the
    // same LayerNorm output -> one [Wqkv ; W1] projection with a column-split epilogue; dense and fc2 add into the same residual
    // row -> one K-concatenated GEMM over [attn | ffn] with a single read-modify-write of x.
    const bool fused = T >= 256 && e->cfg.rotary_dim == 32 && fused_layer_enabled() && (3 * H) % 256 == 0 &&
                       (int64_t)T * F * 2 < ((int64_t)1 << 32);
    if (fused) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        hipStreamIsCapturing(s, &cs);
        if (!e->fused_valid && cs != hipStreamCaptureStatusNone) return set_error_msg(7, "engine: fused weight images must be built before a stream capture");
        TRY(fused_sync(e, s));
        for (int li = 0; li < e->nL; ++li) {
            showo::Layer& l = e->layers[li];
            bf16_t* Kd = kv.k + li * kv.k_lstride;
            bf16_t* Vd = kv.vt + li * kv.v_lstride;
            TRY(showo_layernorm_f32_op16(e->x, l.ln_w, l.ln_b, e->h, nullptr, T, H, e->cfg.ln_eps, op, s));
            TRY(range_check(e, e->h, (int64_t)T * H, s));
            TRY(showo_gemm_qkv_fc1_op16(e->h, H, e->fused_tiled ? l.wq1t : l.wqkv, H, l.bqkv, l.qln_w, l.qln_b, l.kln_w, l.kln_b, e->cosT,
                                        e->sinT, e->Q, Kd, Vd, e->ffn, F, F, B, L, nH, e->cfg.rotary_dim, e->cfg.ln_eps, pos0, Lcap, Lp,
                                        e->fused_tiled ? 1 : 0, op, s));
            TRY(range_check(e, e->Q, (int64_t)T * H, s));
            TRY(range_check(e, e->ffn, (int64_t)T * F, s));
            TRY(showo_attn_fwd_op16(e->Q, Kd, Vd, iv, flag, dense, e->attn, B, nH, L, Lk, Lcap, Lp, H, op, s));
            TRY(range_check(e, e->attn, (int64_t)T * H, s));
            TRY(showo_gemm_kcat_op16(e->attn, H, H, e->ffn, F, F, l.wd2, H + F, l.bd2, e->x, H, e->x, H, T, H, SHOWO_EPI_RESID_F32,
                                     e->fused_tiled ? 1 : 0, op, s));
            TRY(collect_x(e, li + 1, T, s));
        }
        return 0;
    }
    for (int li = 0; li < e->nL; ++li) {
        showo::Layer& l = e->layers[li];
        bf16_t* Kd = kv.k + li * kv.k_lstride;
        bf16_t* Vd = kv.vt + li * kv.v_lstride;
        TRY(showo_layernorm_f32_op16(e->x, l.ln_w, l.ln_b, e->h, nullptr, T, H, e->cfg.ln_eps, op, s));
        TRY(range_check(e, e->h, (int64_t)T * H, s));
        if (T >= 256 && e->cfg.rotary_dim == 32) {
            // prefill / t2i: one kernel (the projection's epilogue normalises, rotates and relayouts the fp32 accumulators)
            TRY(showo_gemm_qkv_fc1_op16(e->h, H, l.wqkv, H, l.bqkv, l.qln_w, l.qln_b, l.kln_w, l.kln_b, e->cosT, e->sinT, e->Q, Kd, Vd,
                                        nullptr, 0, 0, B, L, nH, e->cfg.rotary_dim, e->cfg.ln_eps, pos0, Lcap, Lp, 0, op, s));
        } else {
            TRY(showo_gemm_op16(e->h, H, l.wqkv, H, l.bqkv, 0, e->qkv, 3 * H, nullptr, 0, T, 3 * H, H, SHOWO_EPI_BF16, op, s));
            TRY(range_check(e, e->qkv, (int64_t)T * 3 * H, s));
            TRY(showo_qk_prep_op16(e->qkv, l.qln_w, l.qln_b, l.kln_w, l.kln_b, e->cosT, e->sinT, e->Q, Kd, Vd, B, L, nH,
                                   e->cfg.rotary_dim, e->cfg.ln_eps, pos0, Lcap, Lp, op, s));
        }
        TRY(showo_attn_fwd_op16(e->Q, Kd, Vd, iv, flag, dense, e->attn, B, nH, L, Lk, Lcap, Lp, H, op, s));
        TRY(range_check(e, e->attn, (int64_t)T * H, s));
        TRY(showo_gemm_op16(e->attn, H, l.wd, H, l.bd, 0, e->x, H, e->x, H, T, H, H, SHOWO_EPI_RESID_F32, op, s));
        TRY(showo_gemm_op16(e->h, H, l.w1, H, l.b1, 0, e->ffn, F, nullptr, 0, T, F, H, SHOWO_EPI_GELU_BF16, op, s));
        TRY(range_check(e, e->ffn, (int64_t)T * F, s));
        TRY(showo_gemm_op16(e->ffn, F, l.w2, F, l.b2, 0, e->x, H, e->x, H, T, H, F, SHOWO_EPI_RESID_F32, op, s));
        TRY(collect_x(e, li + 1, T, s));
    }
    return 0;
}

static int check_ready(showo_engine* e, int B, int L) {
    if (!e) return set_error_msg(1, "engine: null handle");
    if (showo_engine_missing(e) != 0) return set_error_msg(4, "engine: weights missing (showo_engine_missing() != 0)");
    if ((int64_t)B * L > e->maxT || B > e->cfg.max_batch || L > e->cfg.max_seq || L > e->cfg.max_pos)
        return set_error_msg(5, "engine: batch/sequence exceeds the configured workspace");
    return 0;
}

static int embed_in(showo_engine* e, const int64_t* ids, const float* embeds, int T, hipStream_t s) {
    if ((ids == nullptr) == (embeds == nullptr)) return set_error_msg(1, "engine: exactly one of ids / embeds must be given");
    if (ids) return showo_embed_f32(ids, e->embed, e->x, T, e->H, e->V, s);
    SHOWO_CHECK_HIP(hipMemcpyAsync(e->x, embeds, (size_t)T * e->H * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}

static int hidden(showo_engine* e, const int64_t* ids, const float* embeds, const float* mask, int B, int L, hipStream_t s) {
    TRY(check_ready(e, B, L));
    TRY(embed_in(e, ids, embeds, B * L, s));
    const int32_t *iv = nullptr, *flag = nullptr;
    if (mask) {
        TRY(showo_mask_compress(mask, e->iv, e->flag, B, L, L, s));
        iv = e->iv; flag = e->flag;
    } else if (e->ext_iv) {
        iv = e->ext_iv; flag = e->ext_flag;
    }
    return run_layers(e, B, L, 0, kv_workspace(e, L), iv, flag, mask, s);
}

static int head_rows(showo_engine* e, const int32_t* rows, int nrows, int col0, int ncols, float* logits, hipStream_t s) {
    if (col0 < 0 || ncols <= 0 || col0 + ncols > e->V) return set_error_msg(1, "engine: bad vocabulary slice");
    if (e->precision == 1) return precise_fast_ok(e) ? head_rows_precise_fast(e, rows, nrows, col0, ncols, logits, s)
                                                       : head_rows_precise(e, rows, nrows, col0, ncols, logits, s);
    if (e->precision == 2) {
        // the split-bf16 head; a decode step (one row) takes the fused LayerNorm + (hi, lo) GEMV: 2 x 240 MB of weights in one launch
        // instead of the LayerNorm launch + a GEMV over the 719 MB [hi | hi | lo] image
        if (nrows == 1 && !rows && showo::g_decode_impl == 0 && showo::decode_fused_shapes_ok(e->H, e->F) && e->wlm_lo &&
            e->lo_loaded.count("showo.lm_head.weight"))
            return showo::decode_split_head(e->x, e->fln_w, e->fln_b, e->cfg.ln_eps, e->H, e->wlm + (int64_t)col0 * e->H,
                                            e->wlm_lo + (int64_t)col0 * e->H, e->blm + col0, logits, ncols, ncols, 1, s);
        return head_rows_precise_fast(e, rows, nrows, col0, ncols, logits, s);  // (any shape)
    }
    if (nrows == 1 && !rows && showo::g_decode_impl == 0 && showo::decode_fused_shapes_ok(e->H, e->F))  // decode step: LN + lm_head in one launch
        return showo::decode_ln_gemv2(e->x, e->fln_w, e->fln_b, e->cfg.ln_eps, e->H, e->wlm + (int64_t)col0 * e->H, e->blm + col0,
                                      nullptr, logits, ncols, nullptr, nullptr, nullptr, 0, s);
    TRY(showo_layernorm_f32_bf16(e->x, e->fln_w, e->fln_b, e->hf, rows, nrows, e->H, e->cfg.ln_eps, s));
    return showo_gemm_bf16(e->hf, e->H, e->wlm + (int64_t)col0 * e->H, e->H, e->blm + col0, 0, logits, ncols, nullptr, 0,
                           nrows, ncols, e->H, SHOWO_EPI_F32, s);
}

extern "C" int showo_engine_forward(showo_engine* e, const int64_t* ids, const float* embeds, const float* mask, int B, int L,
                                    float* logits, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (B == 0 || L == 0) return 0;  // empty batch: nothing to compute (the reference returns an empty logits tensor)
    TRY(hidden(e, ids, embeds, mask, B, L, s));
    return head_rows(e, nullptr, B * L, 0, e->V, logits, s);
}

extern "C" int showo_engine_forward_rows(showo_engine* e, const int64_t* ids, const float* embeds, const float* mask, int B,
                                         int L, const int32_t* rows, int nrows, int col0, int ncols, float* logits,
                                         void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (nrows > e->maxT) return set_error_msg(5, "engine: too many rows");
    TRY(hidden(e, ids, embeds, mask, B, L, s));
    return head_rows(e, rows, nrows, col0, ncols, logits, s);
}

extern "C" int showo_engine_t2i_generate(showo_engine* e, int64_t* ids_cond, int64_t* ids_uncond, const float* mask, int B,
                                         int L, int N, int text_len, int64_t mask_id, int id_offset, int codebook,
                                         float guidance, int steps, const float* mask_len_host, const float* temps_host,
                                         uint64_t seed, const float* exp_noise, const float* uniform, int use_graph,
                                         int64_t* sampled_out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const bool cfg = (ids_uncond != nullptr) && guidance > 0.f;  // modeling_showo.py:136
    const int nseq = cfg ? 2 * B : B;
    TRY(check_ready(e, nseq, L));
    if (N + 2 > L || steps < 1 || !mask_len_host || !temps_host) return set_error_msg(1, "t2i: bad arguments");
    if (id_offset + codebook > e->V) return set_error_msg(1, "t2i: codebook slice exceeds the vocabulary");
    const int img_start = L - (N + 1);  // input_ids[:, -(N+1):-1]
    const int nrows = nseq * N;
    if ((int64_t)nrows * codebook > e->row_logits_cap) {
        e->row_logits_cap = (int64_t)nrows * codebook;
        TRY(e->alloc(&e->row_logits, e->row_logits_cap));
    }
    const int thr = 256;
    const int32_t *iv = nullptr, *flag = nullptr;
    if (mask) {
        TRY(showo_mask_compress(mask, e->iv, e->flag, nseq, L, L, s));  // the mask is step-invariant: compress once
        iv = e->iv; flag = e->flag;
    } else if (e->ext_iv) {
        iv = e->ext_iv; flag = e->ext_flag;
    }
    // ---- prefix reuse.  Rows [0, prefix) of every sequence (pads + text, prefix = position of <soi>) are causal: they never see
    // an image column, and their ids never change, so their hidden states -- hence their K / V in every layer -- are the same
    // in all steps.  Step 0 runs the whole sequence and leaves K / V^T of every layer in a batch-wide cache; the later steps run
    // only the rows [prefix, L) (soi, image tokens, eoi: 258 of 387 at 256x256) against the cached text keys.  Exact up to the
    // grouping of rows into tiles (same arithmetic per row); checked on the device: interval masks only, no prefix row may see
    // a column >= prefix.  use_graph bit 1 disables it.
    // The check is DEFERRED: its flag travels to pinned host memory behind an event and is read after every step of the call has
    // been queued (the GPU never waits for the host); in the case that it fires the call is repeated without reuse -- ids_cond /
    // sampled_out are only written at the very end, so nothing of the optimistic attempt is visible to the caller.
    const int prefix = text_len + 1;
    const int La = L - prefix;
    const int LpC = ((L + 63) / 64) * 64;
    const bool pfast = e->precision == 1 && precise_fast_ok(e);
    const bool reuse_ok = (e->precision == 0 || e->precision == 2 || pfast) && !(use_graph & 2) && steps > 1 && prefix >= 1 && La >= N + 1 && nseq * La >= 1;
    hipStreamCaptureStatus cs0 = hipStreamCaptureStatusNone;
    hipStreamIsCapturing(s, &cs0);
    if (cs0 != hipStreamCaptureStatusNone) return set_error_msg(7, "t2i_generate: the call captures its own graph; do not call it inside a stream capture");
    TRY(fused_sync(e, s));  // weight images of the fused launches: rebuilt here (never inside a capture), so cached graphs stay valid
    if (pfast) TRY(precise_sync(e, s));
    if (e->precision == 2) TRY(head3_sync(e, s));
    if (reuse_ok && !e->pfx_flag) {
        TRY(e->alloc(&e->pfx_flag, 4));
        SHOWO_CHECK_HIP(hipHostMalloc((void**)&e->pfx_host, 64, hipHostMallocDefault));
        SHOWO_CHECK_HIP(hipEventCreateWithFlags(&e->ev_pfx, hipEventDisableTiming));
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
    const bool reuse = reuse_ok && attempt == 0;
    bool check_pending = false;
    build_ids_kernel<<<dim3((B * L + thr - 1) / thr), dim3(thr), 0, s>>>(ids_cond, cfg ? ids_uncond : nullptr, e->ids_all, B, L,
                                                                       text_len + 1);
    init_cur_kernel<<<dim3((B * N + thr - 1) / thr), dim3(thr), 0, s>>>(ids_cond, L, img_start, mask_id, id_offset, e->cur, N, B * N);
    rows_index_kernel<<<dim3((nrows + thr - 1) / thr), dim3(thr), 0, s>>>(e->rows, nseq, L, img_start, N);
    SHOWO_CHECK_HIP(hipGetLastError());
    if (reuse && iv) {
        SHOWO_CHECK_HIP(hipMemsetAsync(e->pfx_flag, 0, sizeof(int32_t), s));
        prefix_check_kernel<<<dim3((nseq * prefix + thr - 1) / thr), dim3(thr), 0, s>>>(iv, flag, e->pfx_flag, nseq, L, prefix);
        SHOWO_CHECK_HIP(hipMemcpyAsync(e->pfx_host, e->pfx_flag, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        SHOWO_CHECK_HIP(hipEventRecord(e->ev_pfx, s));
        check_pending = true;
    }
    KVDest kvc = kv_workspace(e, L);
    if (reuse) {
        const int64_t kn = (int64_t)e->nL * nseq * e->nH * L * 64, vn = (int64_t)e->nL * nseq * e->nH * 64 * LpC;
        if (kn > e->tk_cap) { TRY(e->alloc(&e->tk, kn)); e->tk_cap = kn; }
        if (vn > e->tvt_cap) { TRY(e->alloc(&e->tvt, vn)); e->tvt_cap = vn; }
        SHOWO_CHECK_HIP(hipMemsetAsync(e->tvt, 0, (size_t)vn * sizeof(bf16_t), s));  // pad key columns must stay finite
        if (!e->ids_act) { TRY(e->alloc(&e->ids_act, e->maxT)); TRY(e->alloc(&e->iv_act, e->maxT * 4)); TRY(e->alloc(&e->rows_act, e->maxT)); }
        if (pfast) {
            if (kn > e->tk_lo_cap) { TRY(e->alloc(&e->tk_lo, kn)); e->tk_lo_cap = kn; }
            if (vn > e->tvt_lo_cap) { TRY(e->alloc(&e->tvt_lo, vn)); e->tvt_lo_cap = vn; }
            SHOWO_CHECK_HIP(hipMemsetAsync(e->tvt_lo, 0, (size_t)vn * sizeof(bf16_t), s));
        }
        kvc = KVDest{e->tk, e->tvt, (int64_t)nseq * e->nH * L * 64, (int64_t)nseq * e->nH * 64 * LpC, L, LpC, pfast ? e->tk_lo : nullptr,
                     pfast ? e->tvt_lo : nullptr};
        if (iv) gather_iv_kernel<<<dim3((nseq * La + thr - 1) / thr), dim3(thr), 0, s>>>(iv, e->iv_act, nseq, L, prefix);
        rows_index_kernel<<<dim3((nrows + thr - 1) / thr), dim3(thr), 0, s>>>(e->rows_act, nseq, La, img_start - prefix, N);
        SHOWO_CHECK_HIP(hipGetLastError());
    }
    // one denoise step (modeling_showo.py:135-179).  step < 0: the step index / schedule constants are read on the device.
    // full: run every row (always in step 0); otherwise only the rows [prefix, L) against the cached prefix keys.
    auto denoise_step = [&](int step, bool full) -> int {
        if (full) {
            TRY(showo_embed_f32(e->ids_all, e->embed, e->x, nseq * L, e->H, e->V, s));
            TRY(run_layers(e, nseq, L, 0, kvc, iv, flag, mask, s));
            TRY(head_rows(e, e->rows, nrows, id_offset, codebook, e->row_logits, s));
        } else {
            gather_ids_kernel<<<dim3((nseq * La + thr - 1) / thr), dim3(thr), 0, s>>>(e->ids_all, e->ids_act, nseq, L, prefix);
            TRY(showo_embed_f32(e->ids_act, e->embed, e->x, nseq * La, e->H, e->V, s));
            TRY(run_layers(e, nseq, La, prefix, kvc, iv ? e->iv_act : nullptr, nullptr, nullptr, s));
            TRY(head_rows(e, e->rows_act, nrows, id_offset, codebook, e->row_logits, s));
        }
        const float* lu = cfg ? e->row_logits + (int64_t)B * N * codebook : nullptr;
        const bool dev = step < 0;  // device-step mode: noise offsets are applied inside the kernels
        TRY(showo_cfg_softmax_sample(e->row_logits, lu, codebook, guidance, e->cur, mask_id,
                                     exp_noise ? exp_noise + (dev ? 0 : (int64_t)step * B * N * codebook) : nullptr, seed,
                                     (uint32_t)(dev ? 0 : step), e->sampled, e->sel, B, N, codebook, s));
        TRY(showo_mask_by_topk(e->sel, e->sampled, e->cur, e->ids_all, cfg ? e->ids_all + (int64_t)B * L : nullptr, L, img_start,
                               mask_id, id_offset, dev ? 0.f : mask_len_host[step], dev ? 0.f : temps_host[step],
                               uniform ? uniform + (dev ? 0 : (int64_t)step * B * N) : nullptr, seed, (uint32_t)(dev ? 0 : step), nullptr,
                               B, N, s));
        return 0;
    };
    // hipGraph path (the default of Showo.t2i_generate): ONE denoise step -- every row from <soi> on, ~80 kernels -- is captured with
    // everything that changes from step to step or call to call in device memory (step index, schedule constants, seed, the gathered
    // visibility intervals) and the instantiated graph is CACHED on the engine (LRU of T2I_GRAPH_SLOTS), keyed by every launch
    // argument baked into it; later calls with the same key replay it without capturing again and without a host synchronisation.
    // The eager steps before a capture (2 with prefix reuse: the full step 0 and one active-rows step) launch every kernel variant
    // once (first-use attributes, GEMM tile tuning) outside the capture; step 0 always runs eagerly.  Not combined with per-launch
    // event timing.
    const int n_eager = reuse ? 2 : 1;
    const bool graph = (use_graph & 1) && steps > n_eager && !showo::g_prof_on_query() && (e->precision == 0 || e->precision == 2 || pfast);
    if (!graph) {
        for (int step = 0; step < steps; ++step) TRY(denoise_step(step, step == 0 || !reuse));
    } else {
        if (!e->step_dev) { TRY(e->alloc(&e->step_dev, 4)); }
        if (e->sched_cap < 2 * steps) {
            // (re)allocation changes a pointer baked into cached graphs: they are keyed on it (p[8]) and simply miss afterwards
            TRY(e->alloc(&e->sched_dev, 2 * (steps > 256 ? steps : 256))); e->sched_cap = 2 * (steps > 256 ? steps : 256);
        }
        if (2 * steps <= 512) {
            SchedArgs a;
            for (int i = 0; i < steps; ++i) { a.v[i] = mask_len_host[i]; a.v[steps + i] = temps_host[i]; }
            a.hdr[0] = 0; a.hdr[1] = 0; a.hdr[2] = (int)(uint32_t)(seed & 0xffffffffu); a.hdr[3] = (int)(uint32_t)(seed >> 32);
            a.n = 2 * steps;
            sched_upload_kernel<<<1, 256, 0, s>>>(a, e->sched_dev, e->step_dev);
            SHOWO_CHECK_HIP(hipGetLastError());
        } else {
            std::vector<float> sched(2 * steps);
            for (int i = 0; i < steps; ++i) { sched[i] = mask_len_host[i]; sched[steps + i] = temps_host[i]; }
            const int hdr[4] = {0, 0, (int)(uint32_t)(seed & 0xffffffffu), (int)(uint32_t)(seed >> 32)};  // step index | pad | seed
            SHOWO_CHECK_HIP(hipMemcpyAsync(e->sched_dev, sched.data(), sizeof(float) * 2 * steps, hipMemcpyHostToDevice, s));
            SHOWO_CHECK_HIP(hipMemcpyAsync(e->step_dev, hdr, sizeof(hdr), hipMemcpyHostToDevice, s));
            SHOWO_CHECK_HIP(hipStreamSynchronize(s));  // `sched` / `hdr` are host temporaries
        }
        showo::sampler_set_device_step(e->step_dev, e->sched_dev, steps);
        showo_engine::T2IGraphKey key;
        memset(&key, 0, sizeof(key));  // padding bytes take part in the memcmp below
        key.B = B; key.nseq = nseq; key.L = L; key.N = N; key.prefix = prefix; key.steps = steps; key.id_offset = id_offset;
        key.codebook = codebook; key.reuse = reuse ? 1 : 0; key.cfg = cfg ? 1 : 0; key.has_iv = iv ? 1 : 0;
        key.mask_id = mask_id; key.guidance = guidance; key.prec = e->precision;
        key.p[11] = (reuse && pfast) ? e->tk_lo : nullptr; key.p[12] = (reuse && pfast) ? e->tvt_lo : nullptr;
        if (!reuse) { key.p[0] = iv; key.p[1] = flag; key.p[2] = mask; }  // with reuse the captured step reads e->iv_act only
        key.p[3] = exp_noise; key.p[4] = uniform; key.p[5] = e->row_logits;
        key.p[6] = reuse ? e->tk : nullptr; key.p[7] = reuse ? e->tvt : nullptr; key.p[8] = e->sched_dev; key.p[9] = e->step_dev; key.p[10] = s;
        showo_engine::T2IGraphEntry* ent = nullptr;
        for (auto& ge : e->t2i_graphs)
            if (memcmp(&key, &ge.key, sizeof(key)) == 0) { ent = &ge; break; }
        const bool hit = ent != nullptr;
        int rc = 0;
        const int first_replay = hit ? 1 : n_eager;
        for (int i = 0; i < first_replay && !rc; ++i) {
            rc = denoise_step(-1, i == 0 || !reuse);
            if (!rc) rc = showo::sampler_step_inc(e->step_dev, s);
        }
        if (!rc && !hit) {
            hipGraph_t g = nullptr;
            hipGraphExec_t exec = nullptr;
            hipError_t he = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
            if (he == hipSuccess) {
                rc = denoise_step(-1, !reuse);
                if (!rc) rc = showo::sampler_step_inc(e->step_dev, s);
                hipError_t he2 = hipStreamEndCapture(s, &g);
                if (!rc && he2 != hipSuccess) rc = set_error_hip(he2, "hipStreamEndCapture", __FILE__, __LINE__);
            } else {
                rc = set_error_hip(he, "hipStreamBeginCapture", __FILE__, __LINE__);
            }
            if (!rc) {
                hipError_t he3 = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
                if (he3 != hipSuccess) { exec = nullptr; rc = set_error_hip(he3, "hipGraphInstantiate", __FILE__, __LINE__); }
            }
            if (g) hipGraphDestroy(g);
            if (!rc) {
                if ((int)e->t2i_graphs.size() >= showo_engine::T2I_GRAPH_SLOTS) {  // evict the least recently used graph
                    size_t lru = 0;
                    for (size_t i = 1; i < e->t2i_graphs.size(); ++i)
                        if (e->t2i_graphs[i].last_use < e->t2i_graphs[lru].last_use) lru = i;
                    // the evicted graph may still be executing a previous call's replays on another stream
                    hipDeviceSynchronize();
                    hipGraphExecDestroy(e->t2i_graphs[lru].exec);
                    e->t2i_graphs.erase(e->t2i_graphs.begin() + lru);
                }
                e->t2i_graphs.push_back(showo_engine::T2IGraphEntry{key, exec, 0});
                ent = &e->t2i_graphs.back();
                e->t2i_captures++;
            }
        }
        if (!rc) ent->last_use = ++e->t2i_tick;
        for (int step = first_replay; !rc && step < steps; ++step) {
            hipError_t he = hipGraphLaunch(ent->exec, s);
            if (he != hipSuccess) rc = set_error_hip(he, "hipGraphLaunch", __FILE__, __LINE__);
        }
        showo::sampler_set_device_step(nullptr, nullptr, 0);
        if (rc) return rc;
    }
    if (check_pending) {
        SHOWO_CHECK_HIP(hipEventSynchronize(e->ev_pfx));  // recorded before the first step: long since complete
        if (*(volatile int32_t*)e->pfx_host) continue;  // a text row sees an image column (or the mask is not an interval mask): repeat without reuse
    }
    break;
    }  // attempt
    copy_i64_kernel<<<dim3((B * L + thr - 1) / thr), dim3(thr), 0, s>>>(e->ids_all, ids_cond, B * L);  // in-place update like the reference
    copy_i64_kernel<<<dim3((B * N + thr - 1) / thr), dim3(thr), 0, s>>>(e->sampled, sampled_out, B * N);
    SHOWO_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- KV-cached decode ---------------------------------------------------------------------------------
static int ensure_cache(showo_engine* e, int need) {
    if (need <= e->cache_cap) return 0;
    int cap = ((need + 63) / 64) * 64;
    if (cap > e->cfg.max_pos) cap = ((e->cfg.max_pos + 63) / 64) * 64;
    if (need > cap) return set_error_msg(5, "decode: sequence exceeds max_position_embeddings");
    if (e->cache_cap != 0) return set_error_msg(5, "decode: cache capacity exceeded");
    // allocate once at the maximum the position table allows (2048 tokens: 2 x 201 MB at full size)
    cap = ((e->cfg.max_pos + 63) / 64) * 64;
    int64_t n = (int64_t)e->nL * e->nH * cap * 64;
    TRY(e->alloc(&e->kcache, n));
    TRY(e->alloc(&e->vtcache, n));
    hipMemset(e->kcache, 0, (size_t)n * sizeof(bf16_t));
    hipMemset(e->vtcache, 0, (size_t)n * sizeof(bf16_t));
    e->cache_cap = cap;
    return 0;
}
// accuracy mode: low halves of the decode cache (same capacity), allocated when the first precision-1 prefill arrives
static int ensure_cache_lo(showo_engine* e) {
    if (e->kcache_lo || e->cache_cap == 0) return 0;
    const int64_t n = (int64_t)e->nL * e->nH * e->cache_cap * 64;
    TRY(e->alloc(&e->kcache_lo, n));
    TRY(e->alloc(&e->vtcache_lo, n));
    hipMemset(e->kcache_lo, 0, (size_t)n * sizeof(bf16_t));
    hipMemset(e->vtcache_lo, 0, (size_t)n * sizeof(bf16_t));
    return 0;
}

// prefill of ONE sequence into a given per-layer K / V^T destination (the engine's own decode cache, or a slot of a decode batch):
// runs the prompt, leaves K / V^T of every layer there, returns the last prompt row's visibility intervals and its logits
namespace showo {
int engine_prefill_into(showo_engine* e, const int64_t* ids, const float* embeds, const float* mask, int L, bf16_t* k, bf16_t* vt,
                        int64_t k_lstride, int64_t v_lstride, int cap, int* last_iv_out, float* logits_last, hipStream_t s,
                        bf16_t* k_lo, bf16_t* vt_lo) {
    TRY(embed_in(e, ids, embeds, L, s));
    const int32_t *iv = nullptr, *flag = nullptr;
    if (mask) {
        TRY(showo_mask_compress(mask, e->iv, e->flag, 1, L, L, s));
        iv = e->iv; flag = e->flag;
    } else if (e->ext_iv) {  // caller-built intervals (showo_engine_use_intervals, e.g. from showo_mask_mmu_vit)
        iv = e->ext_iv; flag = e->ext_flag;
    }
    TRY(run_layers(e, 1, L, 0, KVDest{k, vt, k_lstride, v_lstride, cap, cap, k_lo, vt_lo}, iv, flag, mask, s));
    if (iv) {
        int32_t f = 0;
        SHOWO_CHECK_HIP(hipMemcpyAsync(last_iv_out, iv + (int64_t)(L - 1) * 4, 16, hipMemcpyDeviceToHost, s));
        if (flag) SHOWO_CHECK_HIP(hipMemcpyAsync(&f, flag, 4, hipMemcpyDeviceToHost, s));
        SHOWO_CHECK_HIP(hipStreamSynchronize(s));
        if (f) return set_error_msg(6, "decode: prompt mask is not interval-representable; KV-cached decode unsupported");
    } else {
        last_iv_out[0] = 0; last_iv_out[1] = L; last_iv_out[2] = 0; last_iv_out[3] = 0;
    }
    set_iv_kernel<<<1, 64, 0, s>>>(e->rows, L - 1, 0, 0, 0);  // rows[0] = L-1
    return head_rows(e, e->rows, 1, 0, e->V, logits_last, s);
}
}  // namespace showo

extern "C" int showo_engine_prefill(showo_engine* e, const int64_t* ids, const float* embeds, const float* mask, int L,
                                    float* logits_last, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    TRY(check_ready(e, 1, L));
    if (e->precision == 1 && !precise_fast_ok(e))
        return set_error_msg(1, "prefill: in accuracy mode the KV-cached decode needs the production-kernel form (showo_engine_precise_fast)");
    TRY(ensure_cache(e, L + 1));
    if (e->precision == 1) TRY(ensure_cache_lo(e));
    const KVDest kv = kv_decode_cache(e);
    TRY(showo::engine_prefill_into(e, ids, embeds, mask, L, kv.k, kv.vt, kv.k_lstride, kv.v_lstride, kv.Lcap, e->last_iv, logits_last, s,
                                   kv.k_lo, kv.vt_lo));
    e->prompt_len = L;
    e->cache_len = L;
    e->cache_precision = e->precision;
    return 0;
}

// a cache prefilled under one precision must not be decoded under another -- precision 1 would attend with zero low halves
// (bf16-grade results while get_precision() says 1), precision 2 would read bf16 keys as fp16
static int check_cache_precision(const showo_engine* e) {
    if (e->cache_precision != e->precision)
        return set_error_msg(1, "decode: the KV cache was prefilled under another precision (showo_engine_set_precision changed since "
                                "showo_engine_prefill): prefill again");
    return 0;
}

extern "C" int showo_engine_decode_step(showo_engine* e, const int64_t* id, const float* embed, float* logits_last, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!e || e->cache_len <= 0) return set_error_msg(1, "decode_step: prefill first");
    TRY(check_cache_precision(e));
    const int P = e->cache_len;  // position of the new token
    if (P + 1 > e->cache_cap || P + 1 > e->cfg.max_pos) return set_error_msg(5, "decode_step: cache full");
    TRY(embed_in(e, id, embed, 1, s));
    // mask row of the new token = last prompt row + [prompt_len, P] (modeling_showo.py:203-217)
    int a = e->last_iv[0], b = e->last_iv[1], c = e->last_iv[2], d = e->last_iv[3];
    const int L0 = e->prompt_len;
    if (b == L0 && a < b) b = P + 1;
    else if (d == L0 && c < d) d = P + 1;
    else if (!(c < d)) { c = L0; d = P + 1; }
    else if (!(a < b)) { a = L0; b = P + 1; }
    else return set_error_msg(6, "decode_step: mask row needs more than two intervals");
    set_iv_kernel<<<1, 64, 0, s>>>(e->iv1, a, b, c, d);
    hipMemsetAsync(e->flag, 0, 4, s);
    if (e->precision == 1) TRY(ensure_cache_lo(e));
    TRY(run_layers(e, 1, 1, P, kv_decode_cache(e), e->iv1, e->flag, nullptr, s));
    e->cache_len = P + 1;
    return head_rows(e, nullptr, 1, 0, e->V, logits_last, s);
}

// Visibility intervals built on the device (showo_mask_predict_next / _mmu / _mmu_vit) instead of a dense mask: the next
// forward / forward_rows / t2i_generate calls that pass mask == NULL attend with iv int32 [B,L,4] (NULL restores causal).
extern "C" int showo_engine_use_intervals(showo_engine* e, const int32_t* iv, const int32_t* flag) {
    if (!e) return set_error_msg(1, "engine: null handle");
    e->ext_iv = iv;
    e->ext_flag = iv ? flag : nullptr;
    return 0;
}

// Continuation loop: n_steps times { embed(tok) -> 24 layers against the KV cache -> lm_head -> next token -> tok }, the
// position and the mask row living in device memory so that ONE step can be captured into a hipGraph and replayed.
// tok int64[1] (device): in = the token to feed first, out = the last token produced; out_tokens int64 [n_steps] (device).
// top_k == 1: arg-max (the reference caller's setting); otherwise temperature / top-k / multinomial on the device
// (showo_sample_topk semantics; draw j of this call uses noise row / Philox stream step0 + j).
static int decode_loop(showo_engine* e, int64_t* tok, int n_steps, int64_t* out_tokens, float* logits_ws, int top_k,
                       float temperature, const float* exp_noise, uint64_t seed, int step0, int use_graph, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!e || e->cache_len <= 0) return set_error_msg(1, "decode_greedy: prefill first");
    if (!tok || !out_tokens || !logits_ws || n_steps < 1) return set_error_msg(1, "decode_greedy: bad arguments");
    TRY(check_cache_precision(e));
    const int P0 = e->cache_len;
    if (P0 + n_steps > e->cache_cap || P0 + n_steps > e->cfg.max_pos) return set_error_msg(5, "decode_greedy: cache full");
    {   // the decode_step precondition: the mask row must stay a two-interval row
        int a = e->last_iv[0], b = e->last_iv[1], c = e->last_iv[2], d = e->last_iv[3];
        const int L0 = e->prompt_len;
        if (!((b == L0 && a < b) || (d == L0 && c < d) || !(c < d) || !(a < b)))
            return set_error_msg(6, "decode_greedy: mask row needs more than two intervals");
    }
    if (!e->pos_dev) { TRY(e->alloc(&e->pos_dev, 4)); TRY(e->alloc(&e->last_iv_dev, 4)); }
    SHOWO_CHECK_HIP(hipMemcpyAsync(e->pos_dev, &P0, sizeof(int), hipMemcpyHostToDevice, s));
    SHOWO_CHECK_HIP(hipMemcpyAsync(e->last_iv_dev, e->last_iv, 4 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    SHOWO_CHECK_HIP(hipMemsetAsync(e->flag, 0, 4, s));
    SHOWO_CHECK_HIP(hipStreamSynchronize(s));  // P0 / last_iv are host temporaries of this call
    // accuracy mode: the production-kernel layers take the position from the host (no device-side position in those launches), so the
    // steps run eagerly with pos = P0 + i; the token boundary (seam / sampler) keeps pos_dev in step with it
    const bool prec = e->precision == 1;
    if (prec) TRY(ensure_cache_lo(e));
    int step_i = 0;
    if (!prec) showo::attn_set_decode_pos(e->pos_dev, P0 + n_steps);
    // Greedy loop (top_k == 1, the reference caller's setting): the token boundary -- arg-max merge, token store, position increment, next
    // embedding row, next mask row -- is one launch (basic.hip, greedy_token_seam_kernel) at the END of a step, so a step is
    // [layers, lm_head, arg-max partials, boundary] and only the first step embeds its token up front.  SHOWO_DECODE_TOKSEAM=0: the
    // five separate launches.
    static int tokseam = -1;
    if (tokseam < 0) { const char* env = getenv("SHOWO_DECODE_TOKSEAM"); tokseam = env ? (atoi(env) != 0) : 1; }
    const bool seam_tok = tokseam && top_k == 1 && e->V >= 16384;
    auto pre = [&]() -> int {
        TRY(showo_embed_f32(tok, e->embed, e->x, 1, e->H, e->V, s));
        decode_iv_kernel<<<1, 64, 0, s>>>(e->last_iv_dev, e->prompt_len, e->pos_dev, e->iv1);
        return 0;
    };
    if (seam_tok) TRY(pre());
    auto one = [&]() -> int {
        if (!seam_tok) TRY(pre());
        // host-side P only sizes nothing here (grids depend on L = 1); the kernels read the position from pos_dev
        TRY(run_layers(e, 1, 1, prec ? P0 + step_i : P0, kv_decode_cache(e), e->iv1, e->flag, nullptr, s));
        ++step_i;
        TRY(head_rows(e, nullptr, 1, 0, e->V, logits_ws, s));
        if (seam_tok)
            return showo::greedy_token_seam(logits_ws, e->V, tok, out_tokens, e->pos_dev, P0, e->embed, e->x, e->H, e->V, e->last_iv_dev,
                                            e->prompt_len, e->iv1, s);
        if (top_k == 1) TRY(showo_argmax_f32(logits_ws, e->V, tok, s));
        else TRY(showo::sample_topk_launch(logits_ws, e->V, top_k, temperature, exp_noise, (int64_t)e->V, seed, step0, e->pos_dev, P0,
                                           tok, s));
        store_token_kernel<<<1, 64, 0, s>>>(tok, out_tokens, e->pos_dev, P0);
        return showo::sampler_step_inc(e->pos_dev, s);
    };
    int rc = one();  // eager first step (kernel attributes, GEMV variants)
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    const bool graph = use_graph && n_steps > 1 && !showo::g_prof_on_query() && !prec;
    if (!rc && graph) {
        hipError_t he = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        if (he == hipSuccess) {
            rc = one();
            hipError_t he2 = hipStreamEndCapture(s, &g);
            if (!rc && he2 != hipSuccess) rc = set_error_hip(he2, "hipStreamEndCapture", __FILE__, __LINE__);
        } else {
            rc = set_error_hip(he, "hipStreamBeginCapture", __FILE__, __LINE__);
        }
        if (!rc) {
            hipError_t he3 = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            if (he3 != hipSuccess) rc = set_error_hip(he3, "hipGraphInstantiate", __FILE__, __LINE__);
        }
        for (int i = 1; !rc && i < n_steps; ++i) {
            hipError_t he4 = hipGraphLaunch(ge, s);
            if (he4 != hipSuccess) rc = set_error_hip(he4, "hipGraphLaunch", __FILE__, __LINE__);
        }
    } else {
        for (int i = 1; !rc && i < n_steps; ++i) rc = one();
    }
    showo::attn_set_decode_pos(nullptr);
    if (ge) { hipStreamSynchronize(s); hipGraphExecDestroy(ge); }
    if (g) hipGraphDestroy(g);
    if (rc) return rc;
    e->cache_len = P0 + n_steps;
    return 0;
}

extern "C" int showo_engine_decode_greedy(showo_engine* e, int64_t* tok, int n_steps, int64_t* out_tokens, float* logits_ws,
                                          int use_graph, void* stream) {
    return decode_loop(e, tok, n_steps, out_tokens, logits_ws, 1, 1.0f, nullptr, 0, 0, use_graph, stream);
}

extern "C" int showo_engine_decode_sample(showo_engine* e, int64_t* tok, int n_steps, int64_t* out_tokens, float* logits_ws,
                                          int top_k, float temperature, const float* exp_noise, uint64_t seed, int step0,
                                          int use_graph, void* stream) {
    if (!(temperature > 0.f)) return set_error_msg(1, "decode_sample: temperature must be > 0");
    return decode_loop(e, tok, n_steps, out_tokens, logits_ws, top_k, temperature, exp_noise, seed, step0, use_graph, stream);
}
