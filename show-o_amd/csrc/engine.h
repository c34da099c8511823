// Internal layout of the transformer engine (shared by engine.hip and train_engine.hip).
#pragma once
#include "common.h"
#include "../../include/showo_hip.h"
#include <set>
#include <string>
#include <vector>

namespace showo {
struct Layer {
    // wqkv and w1 are ONE allocation ([3H + F, H] rows q|k|v|fc1, biases likewise): the fused [Wqkv ; W1] projection reads it as is
    bf16_t *wqkv = nullptr, *wd = nullptr, *w1 = nullptr, *w2 = nullptr;
    float *bqkv = nullptr, *bd = nullptr, *b1 = nullptr, *b2 = nullptr;
    // K-concatenated image of the two residual projections: wd2 [H, H + F] rows [Wd[n,:] | W2[n,:]], bd2 = bd + b2
    // (rebuilt lazily after a weight load, see fused_sync in engine.hip)
    bf16_t* wd2 = nullptr;
    float* bd2 = nullptr;
    bf16_t* wq1t = nullptr;  // tiled copy of [Wqkv ; W1] (showo_gemm_tile_weight), allocated on first use
    float *ln_w = nullptr, *ln_b = nullptr, *qln_w = nullptr, *qln_b = nullptr, *kln_w = nullptr, *kln_b = nullptr;
    // accuracy mode (showo_engine_set_precision 1): low halves of the weights, w = hi + lo to 2^-17 (same layouts as the hi images;
    // wqkv_lo and w1_lo are one allocation like wqkv / w1)
    bf16_t *wqkv_lo = nullptr, *wd_lo = nullptr, *w1_lo = nullptr, *w2_lo = nullptr;
    // accuracy mode on the production kernels: K-concatenated, tiled split images (precise_sync in engine.hip)
    //   wq1x3 rows [w_hi | w_hi | w_lo] of [Wqkv ; W1]  ([3H + F, 3H]);  wd2x3 rows [Wd_hi | W2_hi | Wd_hi | W2_hi | Wd_lo | W2_lo] ([H, 3 (H + F)])
    bf16_t *wq1x3 = nullptr, *wd2x3 = nullptr;
};
}  // namespace showo

struct showo_engine;
namespace showo {
void sampler_set_device_step(const int* step_dev, const float* sched, int steps);
int sampler_step_inc(int* step_dev, hipStream_t s);
void attn_set_decode_pos(const int* p, int lk_max = 0);
int sample_topk_launch(const float* logits, int V, int top_k, float temperature, const float* exp_noise, int64_t noise_stride,
                       uint64_t seed, int step, const int* pos_dev, int pos_base, int64_t* tok, hipStream_t s);
// fused decode layer (decode.hip, attention.hip)
struct DecodePrefetch;
bool decode_fused_shapes_ok(int H, int F);
int decode_ln_gemv2(const float* x, const float* lnw, const float* lnb, float eps, int H, const bf16_t* W0, const float* b0,
                    bf16_t* out0, float* outf, int N0, const bf16_t* W1, const float* b1, bf16_t* out1, int N1, hipStream_t s,
                    int op = 0);  // op: SHOWO_OP_BF16 | SHOWO_OP_F16 -- the element type of weights and 16-bit activations (all decode entry points)
int decode_split_head(const float* x, const float* lnw, const float* lnb, float eps, int H, const bf16_t* Whi, const bf16_t* Wlo,
                      const float* bias, float* logits, int N, int ld, int nb, hipStream_t s);  // decode_batch.hip: precision 2's head on a decode step
int decode_out_gemv2(float* x, const bf16_t* W0, const bf16_t* a0, const float* b0, int K0, const bf16_t* W1, const bf16_t* a1,
                     const float* b1, int K1, int N, hipStream_t s, int mode = 0, float* y2 = nullptr, int op = 0);
int attn_decode_fused(const bf16_t* qkv, const float* qw, const float* qb, const float* kw, const float* kb, const float* cosT,
                      const float* sinT, bf16_t* K, bf16_t* Vt, const int32_t* iv, bf16_t* O, int nH, int rot, float eps, int pos,
                      int Lcap, int Lp, hipStream_t s, const bf16_t* W2 = nullptr, const bf16_t* ffn = nullptr, const float* b2 = nullptr,
                      int F = 0, int Hout = 0, float* y2 = nullptr, int co_blocks = 0, const DecodePrefetch* pf = nullptr, int op = 0);
// Infinity-Cache prefetch role of the co-scheduled decode launches (decode_common.h): what layer li's attention launch reads ahead.
// next_mb: MB of the next launches' weights ([Wqkv ; W1] of layer li + 1, the lm_head after the last layer), dense: this layer's Wd
// first, blocks: prefetch blocks per launch (0: off).  Defaults from SHOWO_DECODE_PF_MB / _DENSE / _BLOCKS; showo_decode_set_prefetch.
void decode_prefetch_plan(const ::showo_engine* e, int li, DecodePrefetch* pf);
// grid knobs of the decode launches: defaults <- environment (read once) <- showo_decode_set_tuning (sweeps in one process)
struct DecodeTuning {
    // defaults = the best of the one-process sweeps of profiles/r5_decode_dot2_sweep.txt (cfg4 shape, after the v_dot2c rewrite)
    int co_blocks = 128;        // fc2 role blocks of the batch-1 attention launch (one column per wave) SHOWO_DECODE_CO_BLOCKS
    int batch_co_blocks = 128;  // fc2 role blocks of the batched attention launch                       SHOWO_DECODE_BATCH_CO_BLOCKS
    int batch_ln_blocks = 1024; // grid cap of ln_gemvB_kernel<NB <= 4> (register activations)           SHOWO_DECODE_BATCH_LN_BLOCKS
    int ln_blocks = 1024;       // grid cap of ln_gemv2_kernel<2>                                        SHOWO_DECODE_LN_BLOCKS
    int out_blocks = 256;       // grid cap of out_gemv2_kernel / out_gemvB / out_dense_y2B              SHOWO_DECODE_OUT_BLOCKS
};
DecodeTuning& decode_tuning();
int greedy_token_seam(const float* logits, int n, int64_t* tok, int64_t* out_tokens, int* pos, int base, const float* table, float* x, int H,
                      int V, const int32_t* last_iv, int L0, int32_t* iv, hipStream_t s);
// accuracy-mode kernels (precise.hip)
int precise_ln_split(const float* x, const float* w, const float* b, const int32_t* row_index, bf16_t* hi, bf16_t* lo, int rows, int H,
                     float eps, hipStream_t s);
int precise_ln_split3(const float* x, const float* w, const float* b, const int32_t* row_index, bf16_t* out, int rows, int H, float eps,
                      hipStream_t s);
int precise_qk_prep(const float* qkv, const float* qw, const float* qb, const float* kw, const float* kb, const float* cosT,
                    const float* sinT, float* Q, float* K, float* V, int B, int L, int nH, float eps, int pos0, int Lcap, hipStream_t s);
int precise_attention(const float* Q, const float* K, const float* V, const int32_t* iv, const int32_t* flag, const float* dense, float* O,
                      int B, int nH, int Lq, int Lk, int Lcap, int ldo, hipStream_t s);
int precise_gelu_split(const float* f, bf16_t* hi, bf16_t* lo, int64_t n, hipStream_t s);
void engine_batch_free(showo_engine* e);  // decode_batch.hip
extern int g_decode_impl;  // 0 = fused decode layer (default), 1 = the seven-launch path (showo_decode_set_impl)
extern bool g_prof_on_query();
}  // namespace showo

using showo::bf16_t;
using showo::set_error_hip;
using showo::set_error_msg;

struct showo_engine {
    showo_engine_config cfg;
    int H, nL, nH, F, V;
    int64_t maxT;
    std::vector<void*> allocs;
    std::set<std::string> loaded;
    int expected = 0;
    bool fused_valid = false;  // wd2 / bd2 / wq1t images match the weights
    bool fused_tiled = false;  // layout of wd2 (and use of wq1t): tiled (default) or row-major (SHOWO_W_TILED=0)
    bf16_t* wtmp = nullptr;    // staging of one layer's [Wd | W2] rows while its tiled image is built
    // weights
    float* embed = nullptr;
    std::vector<showo::Layer> layers;
    float *fln_w = nullptr, *fln_b = nullptr, *blm = nullptr, *cosT = nullptr, *sinT = nullptr;
    bf16_t* wlm = nullptr;
    // workspace
    float* x = nullptr;
    bf16_t *h = nullptr, *qkv = nullptr, *Q = nullptr, *K = nullptr, *Vt = nullptr, *attn = nullptr, *ffn = nullptr, *hf = nullptr;
    int32_t *iv = nullptr, *flag = nullptr, *rows = nullptr;
    // t2i state
    int64_t *ids_all = nullptr, *cur = nullptr, *sampled = nullptr;
    float *sel = nullptr, *row_logits = nullptr;
    int64_t row_logits_cap = 0;
    // decode (KV cache) state: per-layer caches, capacity cap tokens
    bf16_t *kcache = nullptr, *vtcache = nullptr;
    int cache_cap = 0, cache_len = 0, prompt_len = 0;
    int cache_precision = 0;  // the precision the decode cache was prefilled under (its element type and which halves exist): decode steps must match
    int last_iv[4] = {0, 0, 0, 0};
    int32_t* iv1 = nullptr;
    int64_t* tok1 = nullptr;
    // t2i prefix reuse: per-layer K / V^T of the whole batch (text rows are written once, image rows every step)
    bf16_t *tk = nullptr, *tvt = nullptr;
    int64_t tk_cap = 0, tvt_cap = 0;
    int64_t* ids_act = nullptr;
    int32_t *iv_act = nullptr, *rows_act = nullptr, *pfx_flag = nullptr;
    // caller-provided visibility intervals (showo_engine_use_intervals): used when a call passes no dense mask
    const int32_t* ext_iv = nullptr;
    const int32_t* ext_flag = nullptr;
    // hipGraph replay of the decode step: position and last-prompt-row intervals in device memory
    int* pos_dev = nullptr;
    int32_t* last_iv_dev = nullptr;
    // second stream for the two independent branches of a Phi block (attention branch | fc1): see run_layers
    float* y2 = nullptr;  // forked decode layer: fc2 + b2 of the current layer
    // hipGraph replay of the denoise step: instantiated graphs of one active-rows step are cached (small LRU), keyed by everything
    // BAKED into their launches: shapes, scalars, and every pointer that is neither engine-owned-and-fixed nor refreshed per call.
    // With prefix reuse the captured step reads its visibility intervals from the engine's own gathered copy (iv_act, rewritten by
    // every call), so the caller's mask / interval pointers are NOT part of the key: a caller that builds a fresh, equal mask tensor
    // per batch (reference inference_t2i.py:290-318) replays the same graph.  Without reuse the step reads the caller's intervals /
    // dense mask directly and their pointers stay in the key.
    struct T2IGraphKey {
        int B, nseq, L, N, prefix, steps, id_offset, codebook, reuse, cfg, has_iv;
        int64_t mask_id;
        float guidance;
        int prec;
        const void* p[13];
    };
    struct T2IGraphEntry {
        T2IGraphKey key;
        hipGraphExec_t exec;
        uint64_t last_use;
    };
    static constexpr int T2I_GRAPH_SLOTS = 6;  // full / ragged last batch x CFG on / off x reuse on / off never thrash
    std::vector<T2IGraphEntry> t2i_graphs;
    uint64_t t2i_tick = 0;
    // deferred prefix check of t2i_generate: the device flag lands in pinned host memory behind an event; the host looks at it only
    // after every step of the call is queued (no pipeline bubble), and repeats the call without prefix reuse in the (never yet seen)
    // case that a text row can see an image column
    int32_t* pfx_host = nullptr;
    hipEvent_t ev_pfx = nullptr;
    // accuracy mode (showo_engine_set_precision): 0 = bf16 operands (default, the timed path), 1 = split-bf16 GEMMs + fp32 attention.
    // lo_loaded: GEMM weights whose low halves are current (cleared when somebody rewrites the hi images behind the loader's back).
    // precision 2 (round 6): IEEE-half ("fp16") operands on the same kernels -- the layer weight images (wqkv / w1 / wd / w2 and the fused
    // wq1t / wd2) and every 16-bit activation (h, Q, K, V^T, attn, ffn, the KV caches) then hold fp16 bits; img_f16 says which type the
    // layer images hold RIGHT NOW (set by the loader; a switch between precision 2 and 0 / 1 un-loads the GEMM weights so that the host
    // uploads them again).  The lm_head stays a split-bf16 product in precision 2 (wlm / wlm_lo / wlm3 + p_hf3, as in precision 1).
    int precision = 0;
    bool img_f16 = false;
    bool head3_valid = false;  // wlm3 = [hi | hi | lo] rows of the lm_head matches wlm / wlm_lo
    int64_t* range_count = nullptr;  // precision 2 range check (showo_engine_set_range_check): device counter of saturated fp16 elements
    std::set<std::string> lo_loaded;
    bf16_t* wlm_lo = nullptr;
    bf16_t *p_hlo = nullptr, *p_actlo = nullptr;                            // low halves of h / hf and of attn | gelu(fc1)
    float *p_qkv = nullptr, *p_f = nullptr, *p_Q = nullptr, *p_K = nullptr, *p_V = nullptr, *p_a = nullptr;  // fp32 intermediates
    // accuracy mode on the production kernels (run_layers_precise_fast): split images of the lm_head ([V, 3H] rows [hi | hi | lo]),
    // LayerNorm outputs [T, 3H] = [hi | lo | hi], the attention / gelu(fc1) operand of the residual GEMM act [T, 2 (H + F)] =
    // [attn_hi | ffn_hi | attn_lo | ffn_lo], low halves of Q / K / V^T and of the KV caches
    bool px3_valid = false;
    bf16_t *wlm3 = nullptr, *p_h3 = nullptr, *p_hf3 = nullptr, *p_act = nullptr, *p_Qlo = nullptr, *p_Klo = nullptr, *p_Vtlo = nullptr;
    bf16_t *kcache_lo = nullptr, *vtcache_lo = nullptr, *tk_lo = nullptr, *tvt_lo = nullptr;
    int64_t tk_lo_cap = 0, tvt_lo_cap = 0;
    bf16_t* wtmp3 = nullptr;
    // batched AR decode (decode_batch.hip: showo_engine_batch_begin / _batch_prefill / _batch_decode_greedy)
    struct BatchDecode;
    BatchDecode* bd = nullptr;
    float* collect = nullptr;  // parity hook (showo_engine_set_collect)
    int t2i_captures = 0;  // how often a denoise step was captured (tests: a second identical call must not capture again)
    int* step_dev = nullptr;
    float* sched_dev = nullptr;
    int sched_cap = 0;

    // hipFree a buffer obtained from alloc() and forget it (buffers that are re-sized during the engine's life)
    template <class T>
    void release(T** p) {
        if (!*p) return;
        for (size_t i = 0; i < allocs.size(); ++i)
            if (allocs[i] == (void*)*p) { allocs.erase(allocs.begin() + i); break; }
        hipFree((void*)*p);
        *p = nullptr;
    }
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

