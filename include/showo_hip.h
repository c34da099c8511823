/*
 * showo_hip.h — C ABI of libshowo_hip.so: the MI355X (gfx950) native Show-o hot path.
 *
 * The reference (showlab/Show-o) is pure Python/PyTorch and has no FFI of its own; its boundary for this
 * path is the Python class API `models.Showo` / `models.MAGVITv2` (reference models/__init__.py:1-4).
 * This header is the plain-C surface those classes' replacements (show-o_amd/modeling_*.py) bind through
 * ctypes.  Every entry point names the reference code it replaces (paths relative to the reference
 * root).  Conventions:
 *   - all pointers are DEVICE pointers unless the name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous on it;
 *   - return value: 0 on success, non-zero on error; showo_last_error() returns a static description;
 *   - no allocation ownership crosses the boundary except handles created/destroyed here;
 *   - bf16 tensors are raw uint16 bit patterns; "ids" are int64 as in the reference.
 */
#ifndef SHOWO_HIP_H
#define SHOWO_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SHOWO_ABI_VERSION 1

const char* showo_last_error(void);
int showo_abi_version(void);
/* number of compute units / wave size of the current device; used by bench/tests for sanity. */
int showo_device_info(int* cu_count, int* wave_size, char* arch_name, int arch_name_len);

/* A HIP stream whose kernels never run on `reserve` of the device's CUs (a multiple of 8: reserve / 8 CUs of EVERY XCD, since blocks
 * are dealt to the XCDs round-robin); 0 = a plain non-blocking stream.  The data-parallel trainer runs its compute on such a stream while gradient buckets are being
 * all-reduced, so that RCCL's channel kernels (training/train.py:449,612 through accelerate / DeepSpeed in the reference) find idle
 * CUs instead of queueing behind full-chip GEMM grids.  Destroy with showo_stream_destroy. */
int showo_stream_create_cu_mask(int reserve, void** stream_out);
int showo_stream_destroy(void* stream);
int showo_cu_reserved_max(void);  /* largest `reserve` among the LIVE masked streams (0 = none) */
/* CUs a launch on `stream` may use (device CUs minus the stream's reserve): the one number every grid-sizing rule reads (split-K
 * targets, tile-height model, cooperative-residency test), so launches on a masked stream are sized for what they can occupy */
int showo_cu_usable(void* stream);
/* which CUs does a launch on `stream` run on?  Launches `blocks` one-wave blocks and records per block (xcc_id << 16) | hw_id bits
 * (se, sh, cu) into ids int32 [blocks] (device): tests count the distinct CUs per XCD under a mask. */
int showo_cu_census(int32_t* ids, int blocks, int spin, void* stream);
/* test probe of the wave reductions (common.h): wave w reduces in[64 w .. 64 w + 63]; out[4 w + {0,1,2,3}] = the ds_bpermute butterfly
 * sum, the VALU-only sum (v_permlane32/16_swap + DPP) the decode kernels use, the butterfly max, the VALU-only max.  Same bits. */
int showo_wave_reduce_probe(const float* in, float* out, int n_waves, void* stream);

/* per-launch HIP-event timing of the hot kernels (bench.py roofline leg).  kind: 0 = GEMM, 1 = attention,
 * 2 = conv.  read() synchronises the device and returns summed elapsed ms, launch count, summed algorithmic flops. */
int showo_prof_enable(int on);
int showo_prof_reset(void);
int showo_prof_read(int kind, double* total_ms, int64_t* launches, double* work);
/* time only every stride-th launch of a kind (systematic sample: keeps the instrumentation overhead of a timed region small) */
int showo_prof_set_stride(int stride);
/* launches seen and work submitted per kind since the last reset, timed or not */
int showo_prof_totals(int kind, int64_t* launches, double* work);

/* ---------------------------------------------------------------------------------------------
 * MAGVIT-v2 lookup-free quantizer (reference models/modeling_magvitv2.py:201-206, 208-221, 239-241)
 * ------------------------------------------------------------------------------------------- */
/* z: fp32 [B, C, hw] (NCHW flattened) -> ids int64 [B, hw]; bit for channel c (MSB = channel 0) is (z_c > 0). */
int showo_lfq_pack_nchw(const float* z, int64_t* ids, int B, int C, int hw, void* stream);
/* same, channel-last input z: fp32 [B, hw, ldz] (first C channels used). */
int showo_lfq_pack_nhwc(const float* z, int64_t* ids, int B, int C, int hw, int ldz, void* stream);
/* ids int64 [B, hw] -> z_q fp32 [B, C, hw] of +-1 (get_codebook_entry, NCHW as the reference returns it). */
int showo_lfq_unpack_nchw(const int64_t* ids, float* zq, int B, int C, int hw, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Transformer building blocks (reference models/phi.py)
 * ------------------------------------------------------------------------------------------- */
/* nn.LayerNorm over the last dim (phi.py:744,776,937,1065): x fp32 [rows_in, H] -> y bf16 [rows, H].
 * row_index (optional, int32 [rows]) gathers input rows (used to run the final LN on image rows only). */
int showo_layernorm_f32_bf16(const float* x, const float* w, const float* b, uint16_t* y, const int32_t* row_index,
                             int rows, int H, float eps, void* stream);

/* 16-bit OPERAND TYPE of the `_op16` entry points (round 6): every `uint16_t*` operand, weight image and 16-bit output of such a call
 * holds raw bits of this type.  SHOWO_OP_BF16 = bfloat16 (what the `_bf16` entry points use; they are the op = SHOWO_OP_BF16 case of
 * their `_op16` twin).  SHOWO_OP_F16 = IEEE binary16: the same MFMA / dot2 rate with 11 significand bits instead of 8 -- the operands
 * of showo_engine_set_precision(e, 2), which brings the logits within 1e-3 of the reference's fp32 inference (inference_t2i.py:67,
 * models/phi.py:1182-1183) at the speed of the bf16 path.  Conversions to fp16 saturate at +-65504; subnormals are kept. */
enum { SHOWO_OP_BF16 = 0, SHOWO_OP_F16 = 1 };
/* nn.LayerNorm with the 16-bit output of either type (showo_layernorm_f32_bf16 = op SHOWO_OP_BF16). */
int showo_layernorm_f32_op16(const float* x, const float* w, const float* b, uint16_t* y, const int32_t* row_index,
                             int rows, int H, float eps, int op, void* stream);

/* epilogues of showo_gemm_bf16 */
enum {
    SHOWO_EPI_BF16 = 0,       /* out bf16 = acc + bias                       (q/k/v proj, phi.py:657-659)        */
    SHOWO_EPI_GELU_BF16 = 1,  /* out bf16 = gelu_new(acc + bias)              (fc1 + act, phi.py:208-210)         */
    SHOWO_EPI_F32 = 2,        /* out fp32 = acc + bias                        (lm_head + .float(), phi.py:1182-3) */
    SHOWO_EPI_RESID_F32 = 3   /* out fp32 = resid + acc + bias (resid may alias out) (dense/fc2 + residual, phi.py:727,790) */
};
/* C[M,N] = A[M,K] * W[N,K]^T (+bias[N]); A, W bf16 row-major with leading dims lda, ldw (elements);
 * K % 64 == 0.  `bias` fp32 or NULL.  `bias_per_row` != 0 adds bias[m] instead of bias[n]. */
int showo_gemm_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, int bias_per_row,
                    void* out, int ldo, const float* resid, int ldr, int M, int N, int K, int epilogue, void* stream);
/* The same GEMM on either operand type: A, W and the 16-bit output of SHOWO_EPI_BF16 / SHOWO_EPI_GELU_BF16 are `op` values
 * (q/k/v/dense/fc1/fc2/lm_head of models/phi.py:657-659,727,208-212,1182-1183 in precision 2). */
int showo_gemm_op16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, int bias_per_row,
                    void* out, int ldo, const float* resid, int ldr, int M, int N, int K, int epilogue, int op, void* stream);

/* kernel selection for tests/benchmarks: 0 = by shape (default), 1 = 128x128 register-staged, 2 = 256x256 global_load_lds,
 * 3 = 256x256 phase-split (4 phases per k-tile), 4 = 256x256 phase-split (2 phases per k-tile),
 * 5 = production 2-phase kernel with a per-launch tile height (256 / 208 rows; default for M >= 1024) */
int showo_gemm_set_impl(int impl);
/* development knobs of the 256^2 phase-split kernel (impl 3), used by tools/gemm_bench.cpp: gn = weight panels per
 * tile group of the block->tile map, flags bit0 = run the two wave groups without the one-barrier stagger,
 * dbg = device buffer of 512 uint64 for per-barrier timestamps (NULL = off).  flags >> 8 forces a tile variant of the production
 * kernel (0 = tuned per shape); bits 1/2 weight-panel prefetch on/off, 3/4 direct / LDS-staged bf16 epilogue stores, 6/7 split-K of
 * launches with few tiles off / on (default on: tiles x splits ~ 256 blocks, partials summed in split order by the last block). */
int showo_gemm_tune(int gn, int flags, unsigned long long* dbg);

/* Weight-gradient GEMM on token-major operands (training/train.py:612 loss.backward(): autograd of F.linear): out fp32 [M, N] (ldo)
 * (+)= A^T B, A = dY bf16 [T, M] (lda), B = X bf16 [T, N] (ldb), contraction over the T token rows -- dW = dY^T X without transposing
 * either operand (gfx950's transposing LDS read feeds the MFMA; gemm_tn.hip).  lda, ldb multiples of 8 that cover M, N rounded up
 * to 8; operands 16-byte aligned; any T >= 1.  accumulate != 0 adds into out.  rows_padded != 0: both buffers are readable up to
 * row roundup(T, 64) - 1 (contents arbitrary; zeroed in registers) -- with lda, ldb covering whole 256-column tiles this selects the
 * fast form (unchecked DMAs, 3-deep operand ring); otherwise every fetch is checked.  Split-K by the rule of the production GEMM
 * (a function of (M, N, T) alone: run-to-run identical bits).  Bias gradients: showo_colsum_bf16 on the same dY. */
int showo_gemm_tn_bf16(const uint16_t* A, int lda, const uint16_t* B, int ldb, float* out, int ldo, int M, int N, int T,
                       int accumulate, int rows_padded, void* stream);
/* colsum[c] (+)= sum_t x[t][c], x bf16 [T, C] (row stride ld, 16-byte aligned, ld % 8 == 0); colpart: fp32 scratch of
 * (ceil(T / 32) + 8) * C floats.  Deterministic (per-64-row partials in row order, then a fixed two-level sum). */
int showo_colsum_bf16(const uint16_t* x, int ld, int T, int C, float* colpart, float* colsum, int accumulate, void* stream);

/* Launch counters of the production GEMM family (gemm2p / gemm3w): out3[0] = launches, out3[1] = of those the fused [Wqkv ; W1]
 * save-for-backward form (showo_gemm_qkv_fc1_save_bf16), out3[2] = launches that split K; reset != 0 zeroes them after reading.
 * Parity tests use it to assert that a training batch ran the T >= 256 kernels that the benchmark times. */
int showo_gemm_counters(int64_t* out3, int reset);
/* Cooperative split-K reduction (small-M launches whose tiles x splits blocks are all resident): a block waits for its tile's siblings at
 * most `polls` polls (default 32 768, ~2 ms; negative restores it), then the tile falls back to the last-arriver sum -- the same bits,
 * no abort.  0 makes every early block give up at once (test hook). */
int showo_gemm_set_coop_polls(int polls);

/* Split-precision forms (VQGAN path): every operand is a (hi, lo) bf16 pair, x = hi + lo to ~2^-17; the MFMA
 * accumulates hi*hi + hi*lo + lo*hi in fp32.  fp32 output, optional residual.  Same layouts as the plain calls. */
int showo_gemm_bf16x3(const uint16_t* A, const uint16_t* Alo, int lda, const uint16_t* W, const uint16_t* Wlo, int ldw,
                      const float* bias, int bias_per_row, float* out, int ldo, const float* resid, int ldr, int M, int N,
                      int K, void* stream);
int showo_conv3x3_bf16x3(const uint16_t* x, const uint16_t* xlo, const uint16_t* w, const uint16_t* wlo, const float* bias,
                         const float* resid, float* out, int B, int Hin, int Win, int Cin, int Cout, int mode, void* stream);
/* The same convolution plus the GroupNorm(32) statistics of its OUTPUT -- what showo_gn_stats(out, stats, B, Hout*Wout, Cout)
 * returns, equal to the last bits of the double sums.  In a VQGAN block every 3x3 conv is followed by GroupNorm of its
 * result (common_modules.py:337-357); the conv epilogue produces the per-tile (sum, sumsq) partials (deterministic, no
 * atomics) whenever a 256-pixel tile cannot straddle two images (Hout*Wout % 256 == 0, B*Hout*Wout >= 2048), which saves
 * the read-back of the tensor; otherwise the separate reduction runs.  Cout % 128 == 0; stats: showo_gn_stats_doubles(). */
int showo_conv3x3_bf16x3_gn(const uint16_t* x, const uint16_t* xlo, const uint16_t* w, const uint16_t* wlo, const float* bias,
                            const float* resid, float* out, double* stats, int B, int Hin, int Win, int Cin, int Cout, int mode,
                            void* stream);
/* Launches so far of the split conv kernel that stages the activation operand once per (ky, channel chunk) for all three kx taps
 * (gemm.hip conv3t_split_kernel: modes 0 / 1, Wout a multiple of 16 that divides or is a multiple of 256, Hout * Wout % 256 == 0, launches
 * that do not split K; SHOWO_CONV_3TAP=0 turns it off).  Tests assert their coverage with it. */
int64_t showo_conv3t_launches(void);
/* fp32 -> (hi, lo) bf16 pair */
int showo_split_f32_bf16(const float* src, uint16_t* hi, uint16_t* lo, int64_t n, void* stream);

/* float4 grid-stride device copy (nbytes, src, dst 16-byte aligned): the measured HBM ceiling printed next to the 8 TB/s
 * spec by tools/ceiling.py and bench.py ("re-measure with a copy kernel"). */
int showo_copy_b128(const void* src, void* dst, int64_t nbytes, void* stream);

/* fp32 -> bf16 cast (weight packing) */
int showo_cast_f32_bf16(const float* src, uint16_t* dst, int64_t n, void* stream);
/* fp32 -> 16-bit cast of either operand type (round to nearest even; SHOWO_OP_F16 saturates at +-65504) */
int showo_cast_f32_op16(const float* src, uint16_t* dst, int64_t n, int op, void* stream);
/* number of elements of a 16-bit image whose magnitude is >= 65504 or NaN when read as IEEE half: the range check of precision 2
 * (a saturated convert leaves exactly 65504).  count: device int64, ACCUMULATED into (zero it first). */
int showo_count_f16_saturated(const uint16_t* x, int64_t n, int64_t* count, void* stream);

/* embedding gather (phi.py:1006): ids int64 [T] -> x fp32 [T, H] from table fp32 [V, H]. */
int showo_embed_f32(const int64_t* ids, const float* table, float* x, int T, int H, int V, void* stream);

/* q/k per-head LayerNorm + partial RoPE + head-major relayout (phi.py:661-694):
 * qkv bf16 [B*L, 3*nH*64] (q|k|v) -> Q bf16 [B,nH,L,64] (pre-scaled by 1/sqrt(64)), K bf16 [B,nH,L,64],
 * Vt bf16 [B,nH,64,Lp] (transposed V, Lp % 64 == 0, pad columns zero).  The L new tokens sit at positions
 * pos0 .. pos0+L-1 (pos0 > 0 = KV-cache append): K rows / Vt columns pos0.. are written, K has Lcap rows per head.
 * cos/sin: fp32 [max_pos, rot] tables built exactly as PhiRotaryEmbedding does (phi.py:86-102). */
int showo_qk_prep(const uint16_t* qkv, const float* qln_w, const float* qln_b, const float* kln_w, const float* kln_b,
                  const float* cos_tab, const float* sin_tab, uint16_t* Q, uint16_t* K, uint16_t* Vt,
                  int B, int L, int nH, int rot, float eps, int pos0, int Lcap, int Lp, void* stream);
/* the same on either operand type (qkv in, Q / K / Vt out) */
int showo_qk_prep_op16(const uint16_t* qkv, const float* qln_w, const float* qln_b, const float* kln_w, const float* kln_b,
                       const float* cos_tab, const float* sin_tab, uint16_t* Q, uint16_t* K, uint16_t* Vt,
                       int B, int L, int nH, int rot, float eps, int pos0, int Lcap, int Lp, int op, void* stream);

/* The same result as showo_gemm_bf16(h, Wqkv) + showo_qk_prep in ONE kernel: the QKV projection whose epilogue applies
 * bias, the per-head LayerNorm(64) of q and k, the partial rotary embedding (rot must be 32) and writes Q (pre-scaled),
 * K and V^T head-major -- the fp32 accumulators are normalised directly (no bf16 round trip of qkv).
 * A bf16 [B*L, lda] (LayerNorm output h), Wqkv bf16 [3*nH*64, ldw] rows q|k|v, bias fp32 [3*nH*64]. */
int showo_gemm_qkv_bf16(const uint16_t* A, int lda, const uint16_t* Wqkv, int ldw, const float* bias, const float* qln_w,
                        const float* qln_b, const float* kln_w, const float* kln_b, const float* cos_tab,
                        const float* sin_tab, uint16_t* Q, uint16_t* K, uint16_t* Vt, int B, int L, int nH, int rot,
                        float eps, int pos0, int Lcap, int Lp, void* stream);

/* showo_gemm_qkv_bf16 and fc1 + gelu_new in ONE launch: PhiDecoderLayer feeds the same LayerNorm output to q/k/v_proj and to
 * mlp.fc1 (models/phi.py:776-790, 208-212), so the weight is the row concatenation [Wqkv ; W1] bf16 [3*nH*64 + F, ldw] (bias
 * fp32 [3*nH*64 + F]); output columns below 3*nH*64 take the QKV epilogue above, the others
 * ffn_out bf16 [B*L, ldf] = gelu_new(A W1^T + b1).  Results are bit-identical to the two separate launches whenever both sides
 * use the same k-partition: always with SHOWO_GEMM_SPLITK=0, and otherwise unless the narrower separate launch is a split-K shape
 * (few tiles, K >= 2048: the split count is a function of (M, N, K) only, never of the tile variant the tuner picks).
 * w_tiled = 1: Wqkv_fc1 is the tiled copy made by showo_gemm_tile_weight (ldw must equal K = nH*64). */
int showo_gemm_qkv_fc1_bf16(const uint16_t* A, int lda, const uint16_t* Wqkv_fc1, int ldw, const float* bias,
                            const float* qln_w, const float* qln_b, const float* kln_w, const float* kln_b,
                            const float* cos_tab, const float* sin_tab, uint16_t* Q, uint16_t* K, uint16_t* Vt,
                            uint16_t* ffn_out, int ldf, int F, int B, int L, int nH, int rot, float eps, int pos0, int Lcap,
                            int Lp, int w_tiled, void* stream);
/* showo_gemm_qkv_fc1_bf16 (ffn_out != NULL) / showo_gemm_qkv_bf16 (ffn_out == NULL: F, ldf, w_tiled ignored) on either operand
 * type: A, the weight image and Q / K / V^T / ffn_out are `op` values; LayerNorm(64), RoPE and gelu_new see the fp32 accumulators. */
int showo_gemm_qkv_fc1_op16(const uint16_t* A, int lda, const uint16_t* Wqkv_fc1, int ldw, const float* bias,
                            const float* qln_w, const float* qln_b, const float* kln_w, const float* kln_b,
                            const float* cos_tab, const float* sin_tab, uint16_t* Q, uint16_t* K, uint16_t* Vt,
                            uint16_t* ffn_out, int ldf, int F, int B, int L, int nH, int rot, float eps, int pos0, int Lcap,
                            int Lp, int w_tiled, int op, void* stream);

/* Training forward of showo_gemm_qkv_fc1_bf16 (training/train.py:510-628 through models/phi.py:657-694, 208-212): the same launch
 * also saves what backward reads -- raw_qkv bf16 [B*L, ldraw] = A Wqkv^T + b (pre-LayerNorm q, k and v: the input of
 * showo_qkln_rope_bwd / showo_attn_bwd) and ffn_pre bf16 [B*L, ldf] = A W1^T + b1 (the input of showo_dgelu_bf16) -- and derives
 * Q / K / V^T and ffn_out = gelu_new(ffn_pre) from those ROUNDED values: one launch instead of showo_gemm_bf16 + showo_qk_prep +
 * showo_gemm_bf16 + showo_gelu_bf16, with the forward and the recomputation in backward seeing the same numbers.
 * Q = K = Vt = NULL selects the raw-only form: raw_qkv / ffn_pre / ffn_out are written and the caller runs showo_qk_prep on
 * raw_qkv (same Q / K / V^T bits; A/B switch of the trainer: SHOWO_TRAIN_QKPREP=1). */
int showo_gemm_qkv_fc1_save_bf16(const uint16_t* A, int lda, const uint16_t* Wqkv_fc1, int ldw, const float* bias,
                                 const float* qln_w, const float* qln_b, const float* kln_w, const float* kln_b,
                                 const float* cos_tab, const float* sin_tab, uint16_t* Q, uint16_t* K, uint16_t* Vt,
                                 uint16_t* raw_qkv, int ldraw, uint16_t* ffn_pre, uint16_t* ffn_out, int ldf, int F, int B, int L,
                                 int nH, int rot, float eps, int pos0, int Lcap, int Lp, int w_tiled, void* stream);

/* Accuracy-mode form of showo_gemm_qkv_fc1_bf16 on the same kernels.  Operands are K-concatenated split images: A3 bf16 [M, Kcat]
 * = [a_hi | a_lo | a_hi] (Kcat = 3 K), W3 rows [w_hi | w_hi | w_lo] ([3 nH 64 + F, Kcat], tiled when w_tiled) -- one bf16 GEMM whose
 * fp32 accumulators receive hi*hi + lo*hi + hi*lo.  Every output is a (hi, lo) bf16 pair with hi = RNE(v), lo = RNE(v - hi):
 * Q / Qlo, K / Klo, Vt / Vtlo (layouts of showo_qk_prep), ffn_out / ffn_lo = gelu_new(A W1^T + b1) with IEEE exp and division
 * (both halves with leading dimension ldf).  Reference: models/phi.py:657-694, 208-212 in fp32 (inference_t2i.py:67). */
int showo_gemm_qkv_fc1_split(const uint16_t* A3, int lda, const uint16_t* W3, int ldw, int Kcat, const float* bias,
                             const float* qln_w, const float* qln_b, const float* kln_w, const float* kln_b,
                             const float* cos_tab, const float* sin_tab, uint16_t* Q, uint16_t* Qlo, uint16_t* K, uint16_t* Klo,
                             uint16_t* Vt, uint16_t* Vtlo, uint16_t* ffn_out, uint16_t* ffn_lo, int ldf, int F, int B, int L,
                             int nH, int rot, float eps, int pos0, int Lcap, int Lp, int w_tiled, void* stream);

/* K-concatenated GEMM: out[M,N] = epilogue([A0 | A1] [W0 | W1]^T + bias) with A0 bf16 [M,K0] (lda0), A1 bf16 [M,K1] (lda1) and
 * weight rows [W0[n,:] | W1[n,:]] bf16 [N, ldw] (ldw >= K0 + K1; K0, K1 multiples of 64).  epilogue must be SHOWO_EPI_RESID_F32
 * (out fp32 = acc + bias + resid, in place allowed).
 * Phi's block is parallel-residual (models/phi.py:774-790): x + dense(attn) + fc2(ffn) = x + [attn | ffn] [Wd | W2]^T + (bd + b2)
 * is one launch with ONE read-modify-write of the fp32 residual stream instead of two. */
int showo_gemm_kcat_bf16(const uint16_t* A0, int lda0, int K0, const uint16_t* A1, int lda1, int K1, const uint16_t* W, int ldw,
                         const float* bias, void* out, int ldo, const float* resid, int ldr, int M, int N, int epilogue,
                         int w_tiled, void* stream);
/* the same on either operand type (A0, A1, W) */
int showo_gemm_kcat_op16(const uint16_t* A0, int lda0, int K0, const uint16_t* A1, int lda1, int K1, const uint16_t* W, int ldw,
                         const float* bias, void* out, int ldo, const float* resid, int ldr, int M, int N, int epilogue,
                         int w_tiled, int op, void* stream);

/* Tiled weight layout accepted by the two entry points above (w_tiled = 1): [ceil(N/256)][K/64][256][64] bf16, rows beyond N zero,
 * the eight 16-byte chunks of a row stored at position chunk ^ (row & 7).  One (panel, k-tile) block = 32 KiB = the LDS image the
 * kernel's DMA fills, so a k-loop streams ONE contiguous region per weight panel (DRAM-page friendly; 1 KiB contiguous per
 * wave-instruction).  showo_gemm_tiled_elems = number of bf16 elements of the tiled copy.  W bf16 [N, ldw] row-major (for the
 * K-concatenated GEMM: the [W0 | W1] rows, K = K0 + K1). */
int64_t showo_gemm_tiled_elems(int N, int K);
int showo_gemm_tile_weight(const uint16_t* W, int ldw, int N, int K, uint16_t* out, void* stream);

/* Compress an additive attention mask [B,1,Lq,Lk] fp32 (values 0 / very negative, as built by
 * training/prompting_utils.py:466-511, 591-624) into per-row visibility intervals
 * iv int32 [B, Lq, 4] = (lo1, hi1, lo2, hi2): key c is visible iff lo1<=c<hi1 or lo2<=c<hi2.
 * flag int32[1] is set to 1 if some row is not representable (more than two runs, or a value that is
 * neither 0 nor <= -1e9); the attention kernel then adds the dense mask instead. */
int showo_mask_compress(const float* mask, int32_t* iv, int32_t* flag, int B, int Lq, int Lk, void* stream);

/* On-device mask construction from token ids (reference training/prompting_utils.py:466-511, 591-624).
 * Each call writes the per-row visibility intervals iv int32 [B,L,4] the attention kernels consume and/or the dense
 * additive mask fp32 [B,1,L,L] with the reference's values (0 / float(iinfo(int64).min)); either pointer may be NULL
 * (mmu forms always need iv).  predict_next: flag int32[1] is set if a row needs more than two runs (non-contiguous pads). */
int showo_mask_predict_next(const int64_t* ids, int B, int L, int64_t pad_id, int64_t soi_id, int64_t eoi_id, int rm_pad_in_image,
                            int32_t* iv, int32_t* flag, float* dense, void* stream);
int showo_mask_mmu(const int64_t* ids, int B, int L, int64_t eoi_id, int32_t* iv, float* dense, void* stream);
int showo_mask_mmu_vit(int B, int L, int system_prompt_len, int num_image_tokens, int32_t* iv, float* dense, void* stream);

/* MLM corruption of a training batch's image tokens on the device (reference training/utils.py:77-154
 * mask_or_random_replace_tokens).  Random form: position j of row b is masked iff argsort(noise[b])[j] < num_masked[b]
 * (the reference's `batch_randperm < num_token_masked`, :101-102).  Contiguous form (rect != NULL, :104-129): rows
 * [y0,y1) x columns [x0,x1) of the res x res token grid, rect int32 [B,4] = y0,y1,x0,x1.
 * input_ids = mask_id where masked else the token (:133); labels = the token where masked else ignore_id, or the token
 * everywhere when predict_all != 0 (:143-151); mask uint8 [B,N] is optional. */
int showo_mask_tokens(const int64_t* tokens, const float* noise, const int32_t* num_masked, const int32_t* rect, int res, int B, int N,
                      int64_t mask_id, int64_t ignore_id, int predict_all, int64_t* input_ids, int64_t* labels, uint8_t* mask,
                      void* stream);

/* Fused omni-attention forward (replaces SDPA + dense additive mask, phi.py:715-722):
 * O[b, l, h*64 + d] bf16 = softmax(Q K^T + M) V.  Q,K,Vt as produced by showo_qk_prep (Q already scaled).
 * iv/flag from showo_mask_compress; dense_mask may be NULL iff *flag is known to be 0.
 * Lq query rows (row r attends with mask row r); Lk keys; K holds Lcap rows per head, Vt rows are Lp long.
 * iv == NULL and flag == NULL: causal (query r sees keys <= r + Lk - Lq). */
/* kernel selection for tests/benchmarks: 0 = by shape (default: LDS-tiled for Lq >= 64), 1 = gather form (one wave
 * per 32 query rows, operands straight from L2), 2 = LDS-tiled form (4 waves share 64-key K / V^T tiles staged by
 * global_load_lds; 4 waves/SIMD, no register spill) */
int showo_attn_set_impl(int impl);
/* AR decode step (one new token against the KV cache): 0 = fused layer, three launches (LN + qkv/fc1 GEMV; prep +
 * single-query attention with the fc2 GEMV co-scheduled on the CUs the 32 attention blocks leave idle (F = 8192 only);
 * dense GEMV + both residual adds; default), 1 = the general seven-launch layer, 2 = the fused layer as a plain chain
 * (fc2 in the third launch).  All three give the same bits; 1 and 2 exist so that tests and profiles can compare them. */
int showo_decode_set_impl(int impl);
/* Infinity-Cache prefetch role of the co-scheduled decode launches (batch 1 and batched): `blocks` extra blocks of every layer's
 * attention launch read, by LDS-DMA, weights the NEXT launches will stream -- this layer's dense matrix when dense != 0, then
 * next_mb MB of the next layer's [Wqkv ; W1] (the lm_head after the last layer) -- so that those launches find them in the 256 MiB
 * memory-side cache.  Nothing is computed from the prefetched bytes: results are bit-identical for every setting
 * (tests/test_modules_gpu.py).  blocks = 0 or (next_mb = 0 and dense = 0): role absent.  Defaults: SHOWO_DECODE_PF_MB /
 * SHOWO_DECODE_PF_DENSE / SHOWO_DECODE_PF_BLOCKS.  Takes effect at the next decode call (its graph is captured per call). */
int showo_decode_set_prefetch(int next_mb, int dense, int blocks);
/* Grid knobs of the decode launches for sweeps in one process (defaults <- environment): name in
 * { "co_blocks", "batch_co_blocks", "batch_ln_blocks", "ln_blocks", "out_blocks" }.  Results never depend on them (every grid walks
 * the output columns with a stride; each column is reduced by one wave in one fixed order). */
int showo_decode_set_tuning(const char* name, int value);
int showo_attn_fwd(const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, const int32_t* iv, const int32_t* flag,
                   const float* dense_mask, uint16_t* O, int B, int nH, int Lq, int Lk, int Lcap, int Lp, int ldo,
                   void* stream);
/* the same attention on either operand type: Q, K, V^T, the soft-max numerator P inside the kernel and O are `op` values */
int showo_attn_fwd_op16(const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, const int32_t* iv, const int32_t* flag,
                        const float* dense_mask, uint16_t* O, int B, int nH, int Lq, int Lk, int Lcap, int Lp, int ldo,
                        int op, void* stream);

/* Accuracy-mode attention: Q, K, V^T and the output as (hi, lo) bf16 pairs (same layouts as showo_attn_fwd; O / Olo share ldo);
 * S = Khi Qhi + Khi Qlo + Klo Qhi, fp32 soft-max, O = Vhi Phi + Vhi Plo + Vlo Phi with P split in registers: the fp32 SDPA of
 * models/phi.py:715-722 to ~1e-5 on the MFMA.  Any Lq >= 1. */
int showo_attn_fwd_split(const uint16_t* Q, const uint16_t* Qlo, const uint16_t* K, const uint16_t* Klo, const uint16_t* Vt,
                         const uint16_t* Vtlo, const int32_t* iv, const int32_t* flag, const float* dense_mask, uint16_t* O,
                         uint16_t* Olo, int B, int nH, int Lq, int Lk, int Lcap, int Lp, int ldo, void* stream);

/* Training forward: showo_attn_fwd that also writes lse fp32 [B, nH, Lq] = log sum_k exp(score) of every (masked) row. */
int showo_attn_fwd_lse(const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, const int32_t* iv, const int32_t* flag,
                       const float* dense_mask, uint16_t* O, float* lse, int B, int nH, int Lq, int Lk, int Lcap, int Lp,
                       int ldo, void* stream);

/* X rows [.., L, 64] (row r of (b, h) at x + b*batch_stride + h*head_stride + r*row_stride) -> XT bf16 [B, nH, 64, Lp]
 * (columns >= L zero).  Makes the k-contiguous operand images (Q^T, K^T) of showo_attn_bwd. */
int showo_head_transpose(const uint16_t* x, uint16_t* xt, int B, int nH, int L, int Lp, int64_t batch_stride,
                         int64_t head_stride, int row_stride, void* stream);

/* Backward of the fused attention (autograd of SDPA + mask, phi.py:715-722).  Self-attention, Lq = Lk = L, interval
 * masks (iv from showo_mask_compress, NULL = causal; *flag must be 0).
 * in : Q (pre-scaled), K bf16 [B,nH,L,64]; QT, KT bf16 [B,nH,64,Lp] (showo_head_transpose); V rows token-major (row
 *      stride ldv, head h at column 64h: the v section of the raw qkv projection); O, dO bf16 token-major [B*L, lddo];
 *      lse fp32 [B,nH,L] from showo_attn_fwd_lse.
 * scratch: dOT bf16 [B,nH,64,Lp], D fp32 [B,nH,L].
 * out: dQ (w.r.t. the pre-scaled Q), dK, dV bf16 token-major (row strides ldq/ldk/ldvo, head h at column 64h). */
int showo_attn_bwd(const uint16_t* Q, const uint16_t* K, const uint16_t* QT, const uint16_t* KT, const uint16_t* V, int ldv,
                   const uint16_t* O, const uint16_t* dO, int lddo, uint16_t* dOT, const float* lse, float* D,
                   const int32_t* iv, const int32_t* flag, uint16_t* dQ, int ldq, uint16_t* dK, int ldk, uint16_t* dV,
                   int ldvo, int B, int nH, int L, int Lp, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training kernels (autograd of the block + loss + optimizer; reference training/train.py:590-628).
 * Every reduction is fixed-order (block partials + finalize): bit-reproducible gradients.
 * ------------------------------------------------------------------------------------------- */
/* X bf16 [T, C] (row stride ld) -> XT bf16 [C, Tp] (Tp % 64 == 0, columns >= T zero): the k-contiguous operand of the
 * weight-gradient GEMMs dW = dY^T X.  mode 1 writes gelu_new(X)^T (the MLP activation from the saved pre-activation).
 * colsum (optional, fp32 [C]): column sums of X = bias gradient (accumulate != 0 adds); colpart: scratch fp32 [Tp/64 + 8, C]. */
int showo_transpose_bf16(const uint16_t* x, int ld, uint16_t* xt, int T, int C, int Tp, int mode, float* colpart, float* colsum,
                         int accumulate, void* stream);
/* LayerNorm backward fused with the residual add of the parallel block (phi.py:774-790): dx = dy + dLN(x)^T dh.
 * x, dh, dy fp32 [T,H]; dx32 fp32 (may alias dy), dx16 bf16 copy (may be NULL); dgb fp32 [2,H] = (dgamma, dbeta);
 * part: scratch fp32 [showo_ln_bwd_blocks(T), 2, H]. */
int showo_ln_bwd(const float* x, const float* gamma, const float* dh, const float* dy, float* dx32, uint16_t* dx16, float* part,
                 float* dgb, int T, int H, float eps, void* stream);
int showo_ln_bwd_blocks(int T);
/* the same plus dxsum[H] = column sums of dx16: dx is the output gradient of the block below, whose dense / fc2 bias gradients are
 * exactly these sums (no second pass over dx16).  part: scratch fp32 [showo_ln_bwd_blocks(T), 3, H]. */
int showo_ln_bwd_colsum(const float* x, const float* gamma, const float* dh, const float* dy, float* dx32, uint16_t* dx16, float* part,
                        float* dgb, float* dxsum, int T, int H, float eps, void* stream);
/* Backward of q/k LayerNorm(64) + partial rotary + the 1/8 fold (phi.py:661-694): dq, dk bf16 [T, ldg] (w.r.t. the stored Q,
 * K), raw qkv bf16 [T, 3*nH*64] -> dqkv bf16 [T, 3*nH*64] (q and k sections), dparams fp32 [4,64] = (dq_ln_w, dq_ln_b,
 * dk_ln_w, dk_ln_b); part: scratch fp32 [showo_qkln_rope_bwd_blocks(T, nH), 4, 64]. */
int showo_qkln_rope_bwd(const uint16_t* dq, const uint16_t* dk, int ldg, const uint16_t* qkv, const float* qw, const float* kw,
                        const float* cos_tab, const float* sin_tab, uint16_t* dqkv, float* part, float* dparams, int T, int L,
                        int nH, int rot, float eps, void* stream);
int showo_qkln_rope_bwd_blocks(int T, int nH);
/* The three mean cross-entropies of Showo.forward (modeling_showo.py:80-98) and their gradient.
 * logits fp32 [B*L, ldl]; labels int64 [B,L] (-100 = ignore).  losses fp32 [3] = (t2i, lm, mmu);
 * dlogits (optional) bf16 [B*L, ldd] = d(g_t2i*loss_t2i + g_lm*loss_lm + g_mmu*loss_mmu)/dlogits, pad columns zero.
 * scratch: rows_ws 12*B*L bytes, counts int[3], rowloss fp32 [2*B*L]. */
int showo_ce_loss(const float* logits, int ldl, const int64_t* labels, int B, int L, int V, int b_t2i, int b_lm, int b_mmu,
                  int max_seq_len, float g_t2i, float g_lm, float g_mmu, void* rows_ws, int* counts, float* rowloss,
                  uint16_t* dlogits, int ldd, float* losses, void* stream);
/* Embedding backward (deterministic): dE[ids[t]] = sum of dx[t] over equal ids, in position order.  dE must be
 * zero-filled by the caller; order_ws: scratch int[2*T]. */
int showo_embed_bwd(const int64_t* ids, const float* dx, float* dE, int* order_ws, int T, int H, int V, void* stream);
/* torch.optim.AdamW step on fp32 tensors (step counts from 1). */
int showo_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                float weight_decay, int step, void* stream);
int showo_scale_f32(float* x, int64_t n, float s, void* stream);

/* bf16 gradient wire of the data-parallel exchange (reference: mixed_precision bf16 + DeepSpeed ZeRO-2 reduce,
 * configs/showo_pretraining_stage1.yaml:87, accelerate_configs/8_gpus_deepspeed_zero2.yaml:2-16; SURVEY.md 8e: 2.90 GB per rank
 * per step).  pack: wire[i] = bf16(grad[i] * scale), scale = 1 / world_size applied before rounding, so that the all-reduce SUM
 * of the wire is the mean; unpack: grad[i] = float(wire[i]).  grad 16-byte, wire 8-byte aligned. */
int showo_grad_wire_pack(const float* grad, uint16_t* wire, int64_t n, float scale, void* stream);
int showo_grad_wire_unpack(const uint16_t* wire, float* grad, int64_t n, void* stream);
/* df = da * gelu_new'(f) (bf16, elementwise) */
int showo_dgelu_bf16(const uint16_t* da, const uint16_t* f, uint16_t* df, int64_t n, void* stream);
/* showo_dgelu_bf16 on a [T, C] matrix (row stride ld) fused with the column sums of its result (the fc1 bias gradient, like
 * showo_colsum_bf16 on df); colpart: fp32 scratch of (ceil(T / 64) + 8) * C floats; df may alias da. */
int showo_dgelu_colsum_bf16(const uint16_t* da, const uint16_t* f, uint16_t* df, int ld, int T, int C, float* colpart, float* colsum,
                            void* stream);

/* ---------------------------------------------------------------------------------------------
 * t2i sampler (reference models/modeling_showo.py:140-179, models/sampling.py:14-36)
 * ------------------------------------------------------------------------------------------- */
/* One categorical draw per image token from softmax((1+w)*cond - w*uncond) using the algorithm
 * torch.multinomial uses for one draw (argmax_i p_i / E_i, E~Exp(1)).
 * logits_c/logits_u: fp32 [B*N, ld] (uncond NULL -> no CFG).  cur: int64 [B,N] current code ids or mask_id.
 * exp_noise: optional fp32 [B*N, V] injected noise (tests); else Philox(seed, step).
 * out: sampled int64 [B,N] (known tokens keep their id), sel_prob fp32 [B,N] (FLT_MAX for known tokens). */
int showo_cfg_softmax_sample(const float* logits_c, const float* logits_u, int ld, float guidance, const int64_t* cur,
                             int64_t mask_id, const float* exp_noise, uint64_t seed, uint32_t step,
                             int64_t* sampled, float* sel_prob, int B, int N, int V, void* stream);
/* mask_by_random_topk + write-back (modeling_showo.py:166-179): per sample b,
 * mask_len = max(1, min(#unknown-1, mask_len_f)); conf = log(p)+temp*gumbel(u); cut = sort(conf)[mask_len];
 * masking = conf < cut; ids_cond/ids_uncond[b, img_start + i] = masking ? mask_id : sampled+offset;
 * cur[b,i] = masking ? mask_id : sampled.  uniform: optional injected fp32 [B,N] draws. */
int showo_mask_by_topk(const float* sel_prob, const int64_t* sampled, int64_t* cur, int64_t* ids_cond,
                       int64_t* ids_uncond, int ld_ids, int img_start, int64_t mask_id, int64_t id_offset,
                       float mask_len_f, float temperature, const float* uniform, uint64_t seed, uint32_t step,
                       uint8_t* masking_out, int B, int N, void* stream);

/* ---------------------------------------------------------------------------------------------
 * VQGAN building blocks (reference models/common_modules.py, channel-last activations)
 * ------------------------------------------------------------------------------------------- */
/* GroupNorm(32, eps) statistics over x fp32 NHWC [B, HW, C]: stats double [B, 32, 2] = (sum, sumsq).
 * Deterministic two-pass reduction (no atomics): `stats` must provide showo_gn_stats_doubles(B, HW) doubles
 * ([B,32,2] result followed by per-block partials). */
int showo_gn_stats_doubles(int B, int HW);
int showo_gn_stats(const float* x, double* stats, int B, int HW, int C, void* stream);
/* second pass alone: stats[b][g][2] = sum_k part[b][k][g][2] over nblk per-block partials, in k order */
int showo_gn_finalize(const double* part, double* stats, int B, int nblk, void* stream);
/* y bf16 NHWC = act((x - mean) * rstd * gamma + beta); act = swish if do_swish (common_modules.py:16-24).
 * ylo (optional): low half for split precision, ylo = bf16(value - float(y)). */
int showo_gn_apply(const float* x, const double* stats, const float* gamma, const float* beta, uint16_t* y, uint16_t* ylo,
                   int B, int HW, int C, float eps, int do_swish, void* stream);
/* 3x3 convolution as implicit GEMM on MFMA.  x bf16 NHWC [B,Hin,Win,Cin]; w bf16 [Cout][3][3][Cin];
 * out fp32 NHWC [B,Hout,Wout,Cout] = conv + bias (+ resid).  mode: 0 = stride 1 pad 1;
 * 1 = nearest-2x upsample then stride 1 pad 1 (common_modules.py:36-40); 2 = pad (0,1,0,1) then stride 2
 * (common_modules.py:83-88).  Cin % 64 == 0 (thin inputs are zero-padded to 64 channels); any Cout. */
int showo_conv3x3_bf16(const uint16_t* x, const uint16_t* w, const float* bias, const float* resid, float* out,
                       int B, int Hin, int Win, int Cin, int Cout, int mode, void* stream);
/* direct convolution for the thin layers (Cin or Cout in {3,13}): x fp32 NHWC, w fp32 [Cout][k][k][Cin],
 * out fp32 NHWC; ksize 1 or 3, stride 1, pad (k-1)/2. */
int showo_conv_small_f32(const float* x, const float* w, const float* bias, float* out, int B, int H, int W,
                         int Cin, int Cout, int ksize, void* stream);
/* row softmax with scale: x fp32 [rows, n] -> y bf16 [rows, ldy] = softmax(x * scale) (AttnBlock, common_modules.py:199-201) */
int showo_softmax_rows_bf16(const float* x, uint16_t* y, uint16_t* ylo, int rows, int n, int ldy, float scale, void* stream);
/* fp32 [P, C] -> bf16 [P, Cpad] with zero-padded channels (feeds the MFMA conv for the 3-/13-channel inputs) */
int showo_pad_cast_bf16(const float* x, uint16_t* y, uint16_t* ylo, int64_t P, int C, int Cpad, void* stream);
/* ids int64 [B, hw] -> z_q fp32 [B, hw, C] (channel-last form of get_codebook_entry) */
int showo_lfq_unpack_nhwc(const int64_t* ids, float* zq, int B, int C, int hw, void* stream);
/* layout changes at the image / latent boundary */
int showo_nchw_to_nhwc_f32(const float* x, float* y, int B, int C, int HW, void* stream);
int showo_nhwc_to_nchw_f32(const float* x, float* y, int B, int C, int HW, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Engines: whole-module entry points.  Handles own packed bf16 weights + workspaces in HBM.
 * ------------------------------------------------------------------------------------------- */
typedef struct showo_engine showo_engine;
typedef struct {
    int hidden, layers, heads, ffn, vocab;
    int rotary_dim, max_pos;
    float ln_eps, rope_theta;
    int max_batch, max_seq; /* workspace sizing: tokens = max_batch*max_seq */
} showo_engine_config;

int showo_engine_create(const showo_engine_config* cfg, showo_engine** out);
void showo_engine_destroy(showo_engine* e);
/* Load one tensor by its REFERENCE state-dict key, e.g.
 * "showo.model.layers.3.self_attn.q_proj.weight"; src = device fp32, n = element count.  GEMM weights are
 * converted to bf16 and q/k/v are packed into one [3H,H] matrix.  cos/sin tables: keys "rope.cos"/"rope.sin". */
int showo_engine_load(showo_engine* e, const char* key, const float* src, int64_t n, void* stream);
/* Where showo_engine_load would put a tensor, without copying: exactly one of *dst_bf16 (weight image) / *dst_f32 (vector, embedding
 * table) is set.  The trainer's fused optimizer launch writes the refreshed images through these pointers and then calls
 * showo_engine_weights_touched (the bookkeeping showo_engine_load does: fused weight images / cached graphs are rebuilt). */
int showo_engine_slot(showo_engine* e, const char* key, int64_t n, uint16_t** dst_bf16, float** dst_f32);
int showo_engine_weights_touched(showo_engine* e);
/* number of tensors still missing (0 = ready) */
int showo_engine_missing(const showo_engine* e);
/* number of times showo_engine_t2i_generate captured a denoise step into a hipGraph on this engine (the instantiated graph is
 * cached: identical calls replay it without capturing again) */
int showo_engine_t2i_captures(const showo_engine* e);
/* parity hook: while buf != NULL every forward copies the fp32 residual stream into buf fp32 [layers + 1, B*L, hidden] (slot 0 = the
 * embedded input, slot i = output of transformer block i - 1), so that a test can check each block against the oracle evaluated on
 * the block's own input as the GPU computed it (no error amplification across blocks). */
int showo_engine_set_collect(showo_engine* e, float* buf);
/* Operand precision.  precision 0 (default): bf16 GEMM / attention operands, fp32 accumulation.  precision 1: the reference's fp32
 * inference (inference_t2i.py:67 keeps the model in fp32; models/phi.py:1182-1183 returns fp32 logits) to ~1e-5 end to end with
 * split-bf16 operands (x = hi + lo to 2^-17, products hi*hi + lo*hi + hi*lo accumulated in fp32).
 * Production form (showo_engine_precise_fast() == 1: rotary_dim 32, 3 * hidden a multiple of 256 -- Phi-1.5's shape): the SAME
 * kernels and launch structure as precision 0 on K-concatenated split images -- activations [hi | lo | hi], weight rows
 * [hi | hi | lo], one bf16 GEMM over 3K per product (showo_gemm_qkv_fc1_split, showo_gemm_kcat_bf16), split attention
 * (showo_attn_fwd_split) -- so every entry point works in this mode, including prefix reuse and hipGraph replay in _t2i_generate and
 * the KV-cached prefill / decode (decode steps run eagerly).  Otherwise (tiny test shapes, SHOWO_PRECISE_FAST=0) the fp32 reference
 * kernels of csrc/precise.hip serve forward / forward_rows / t2i_generate and the KV-cached entry points refuse.
 * The weights must be uploaded (showo_engine_load) AFTER switching to precision 1 -- the loader then also keeps their low halves;
 * showo_engine_precise_ready tells whether they are current. */
/* precision 2 (round 6): IEEE-half ("fp16") operands.  The SAME launches, tiles, prefix reuse and hipGraph replay as precision 0 --
 * v_mfma_f32_*_f16 runs at the bf16 rate -- with operand rounding 2^-12 instead of 2^-9: weights of the 24 blocks, LayerNorm output,
 * Q / K / V^T, the soft-max numerator, the attention output and gelu(fc1) are fp16 (saturating converts, subnormals kept); the residual
 * stream, LayerNorm / RoPE / soft-max arithmetic and every accumulator stay fp32, and the final LayerNorm + lm_head run as the split-bf16
 * product of precision 1 (so the final hidden state and the lm_head weight are not rounding points).  End to end against the fp32
 * reference: rel_rms ~8e-4 / rel_max <= 1e-3 at model scale where bf16 operands give 7e-3 (oracle/predict_rounding.py,
 * tests/test_modules_gpu.py).  KV-cached decode steps run on the general (seven-launch) layer; the batched decode refuses.
 * Switching between precision 2 and 0 / 1 un-loads the GEMM weights (their images change element type): showo_engine_missing() is
 * then > 0 until the host has uploaded them again.  Training keeps bf16 images (showo_train_* refuse a precision-2 engine). */
int showo_engine_set_precision(showo_engine* e, int precision);
int showo_engine_get_precision(const showo_engine* e);
/* precision 2 range check: while count != NULL (device int64, zero it first) every forward adds the number of fp16 activation
 * elements (LayerNorm output, q|k|v or Q, attention output, gelu(fc1)) that left a convert saturated (|x| = 65504) or non-finite.
 * Extra launches: a diagnostic, not for timed runs.  Random-init weights never saturate; REAL Phi-1.5 checkpoints decide the range
 * question for a deployment -- run one batch with the counter on. */
int showo_engine_set_range_check(showo_engine* e, int64_t* count);
int showo_engine_precise_ready(const showo_engine* e);
int showo_engine_precise_fast(const showo_engine* e);
/* process-wide A/B switch of the two accuracy-mode implementations (default: SHOWO_PRECISE_FAST, on): 0 = always the fp32 reference
 * kernels of csrc/precise.hip.  Call showo_engine_set_precision(e, 1) again after switching it on (workspaces are made there). */
int showo_precise_set_fast(int on);
/* Showo.forward without labels (modeling_showo.py:76-79 -> phi.py:953-1183):
 * ids int64 [B,L] or embeds fp32 [B,L,H] (exactly one non-NULL); mask fp32 [B,1,L,L] or NULL (causal);
 * logits fp32 [B,L,vocab]. */
/* attend with caller-built visibility intervals (showo_mask_predict_next etc.) in the following calls that pass
 * mask == NULL; iv == NULL restores the default (no mask = causal).  flag may be NULL (= known representable). */
int showo_engine_use_intervals(showo_engine* e, const int32_t* iv, const int32_t* flag);
int showo_engine_forward(showo_engine* e, const int64_t* ids, const float* embeds, const float* mask, int B, int L,
                         float* logits, void* stream);
/* Final-LN'ed hidden rows + restricted lm_head: logits fp32 [nrows, ncols] for token rows `rows`
 * (int32 [nrows], index into B*L) and vocabulary columns [col0, col0+ncols).  Same math as the reference
 * restricted to the entries t2i_generate consumes (modeling_showo.py:144). */
int showo_engine_forward_rows(showo_engine* e, const int64_t* ids, const float* embeds, const float* mask, int B, int L,
                              const int32_t* rows, int nrows, int col0, int ncols, float* logits, void* stream);
/* Showo.t2i_generate (modeling_showo.py:104-181).  ids_cond int64 [B,L] is updated in place like the
 * reference; ids_uncond may be NULL (no CFG).  mask fp32 [(2)B,1,L,L].  mask_len_host/temps_host: per-step
 * host constants (floor(N*schedule((k+1)/T)) and the compounding temperature).  use_graph: bit 0 = capture one
 * denoise step into a hipGraph and replay it; bit 1 = do NOT reuse the step-invariant text rows (by default step 0 runs
 * the whole sequence and leaves every layer's K / V^T in a cache, later steps run only the rows from <soi> on).  Optional injected noise (tests): exp_noise [steps,B*N,V],
 * uniform [steps,B,N].  out sampled int64 [B,N]. */
int showo_engine_t2i_generate(showo_engine* e, int64_t* ids_cond, int64_t* ids_uncond, const float* mask, int B, int L,
                              int num_vq_tokens, int text_len, int64_t mask_id, int id_offset, int codebook,
                              float guidance, int steps, const float* mask_len_host, const float* temps_host,
                              uint64_t seed, const float* exp_noise, const float* uniform, int use_graph,
                              int64_t* sampled_out, void* stream);
/* Incremental (KV-cached) decode used by mmu_generate (modeling_showo.py:183-240).  prefill: run the prompt
 * (ids or embeds, mask [1,1,L,L]) and keep K/V; step: append one token (id or embedding row) whose mask row
 * is `last prompt mask row + causal` exactly as the reference grows it (modeling_showo.py:203-217).
 * logits_last fp32 [vocab] = logits of the last position. */
int showo_engine_prefill(showo_engine* e, const int64_t* ids, const float* embeds, const float* mask, int L,
                         float* logits_last, void* stream);
int showo_engine_decode_step(showo_engine* e, const int64_t* id, const float* embed, float* logits_last, void* stream);
/* n_steps greedy (top_k = 1: the reference caller's setting, inference_mmu.py:81) continuation steps without host round trips:
 * { embed(tok) -> layers against the KV cache -> lm_head -> arg-max -> tok }.  tok int64[1] (device) holds the token to feed
 * first and the last produced one afterwards; out_tokens int64 [n_steps] (device); logits_ws fp32 [vocab] scratch.
 * use_graph: capture one step (position and mask row live in device memory) into a hipGraph and replay it. */
int showo_engine_decode_greedy(showo_engine* e, int64_t* tok, int n_steps, int64_t* out_tokens, float* logits_ws, int use_graph,
                               void* stream);

/* Batched KV-cached greedy decode: nb <= 8 independent sequences (BASELINE cfg4's "batch=4 images"), nb caches, ONE weight stream per
 * token step -- the reference's batch-1 mmu_generate (models/modeling_showo.py:183-240) serves them one after the other
 * (inference_mmu.py:87-177) and streams the 2.66 GB of weights once per token per sequence.  Every sequence gets the bits of its own
 * batch-1 run (same lane split / accumulation order per sequence): tokens and logits equal nb showo_engine_decode_greedy runs.
 *   showo_engine_batch_begin(e, nb, cap_tokens): (re)sizes nb caches for cap_tokens >= max(prompt) + new tokens + 1 and clears them;
 *   showo_engine_batch_prefill(e, b, ...): showo_engine_prefill of sequence b into its slot (own length, own mask);
 *   showo_engine_batch_decode_greedy: tok int64 [nb] (device; in = first token to feed per sequence, out = last produced),
 *     out_tokens int64 [nb, n_steps], logits_ws fp32 [nb, vocab]; every sequence advances n_steps tokens (the caller cuts at <eot>);
 *     one hipGraph replay per step with the nb positions in device memory.  bf16 operands (precision 0) only. */
int showo_engine_batch_begin(showo_engine* e, int nb, int cap_tokens);
int showo_engine_batch_prefill(showo_engine* e, int b, const int64_t* ids, const float* embeds, const float* mask, int L,
                               float* logits_last, void* stream);
int showo_engine_batch_decode_greedy(showo_engine* e, int64_t* tok, int n_steps, int64_t* out_tokens, float* logits_ws, int use_graph,
                                     void* stream);
/* Next-token draw of the AR decode (modeling_showo.py:220-228): x = logits / temperature; values below the top_k-th largest
 * are dropped (top_k <= 0 or >= V: none); token = multinomial(softmax(x), 1) computed as argmax_i p_i / E_i, E ~ Exp(1):
 * E = exp_noise[step * V + i] when exp_noise != NULL (parity tests inject the reference's draws), else Philox(seed; step, i). */
int showo_sample_topk(const float* logits, int V, int top_k, float temperature, const float* exp_noise, uint64_t seed, int step,
                      int64_t* tok, void* stream);
/* showo_engine_decode_greedy with that draw instead of the arg-max; draw j of the call uses noise row / stream step0 + j. */
int showo_engine_decode_sample(showo_engine* e, int64_t* tok, int n_steps, int64_t* out_tokens, float* logits_ws, int top_k,
                               float temperature, const float* exp_noise, uint64_t seed, int step0, int use_graph, void* stream);
/* greedy/top-k=1 pick on device: out int64[1] = argmax(logits) (first maximal index, like torch.topk/multinomial on a one-hot). */
int showo_argmax_f32(const float* x, int n, int64_t* out, void* stream);

typedef struct showo_vq showo_vq;
typedef struct {
    int ch, z_channels;
    int enc_ch_mult[8], enc_blocks[8], enc_levels;
    int dec_ch_mult[8], dec_blocks[8], dec_levels;
    int max_batch, max_res; /* workspace sizing */
    int precision;          /* 0 = bf16 operands; 1 = split (hi+lo) bf16 operands, fp32-class accuracy (default of MAGVITv2) */
} showo_vq_config;
int showo_vq_create(const showo_vq_config* cfg, showo_vq** out);
void showo_vq_destroy(showo_vq* v);
/* key = reference state-dict key ("decoder.up.3.block.0.conv1.weight", ...); src fp32 in the reference's
 * layout (conv: [Cout,Cin,kh,kw]); repacked to channel-last bf16 inside. */
int showo_vq_load(showo_vq* v, const char* key, const float* src, int64_t n, void* stream);
int showo_vq_missing(const showo_vq* v);
/* MAGVITv2.decode_code (modeling_magvitv2.py:429-433): ids int64 [B,h*w] -> image fp32 NCHW [B,3,16h,16w]. */
int showo_vq_decode_code(showo_vq* v, const int64_t* ids, int B, int h, int w, float* image, void* stream);
/* MAGVITv2.get_code (modeling_magvitv2.py:423-427): pixels fp32 NCHW [B,3,H,W] -> ids int64 [B,(H/16)*(W/16)];
 * z_out (optional) fp32 [B,13,H/16,W/16] = pre-quantisation latents (for the |z|<eps agreement test). */
int showo_vq_get_code(showo_vq* v, const float* pixels, int B, int H, int W, int64_t* ids, float* z_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training step (reference Showo.forward with labels + loss.backward(), models/modeling_showo.py:59-102,
 * training/train.py:590-612).  A trainer borrows an engine's weights and workspaces and owns the saved activations,
 * the transposed weight images of the dgrad GEMMs and one fp32 gradient buffer per reference parameter.
 * ------------------------------------------------------------------------------------------- */
typedef struct showo_trainer showo_trainer;
int showo_train_create(showo_engine* e, int max_batch, int max_seq, showo_trainer** out);
void showo_train_destroy(showo_trainer* t);
/* call after showo_engine_load changed weights (optimizer step): the transposed images are rebuilt lazily */
int showo_train_invalidate_weights(showo_trainer* t);
/* Training with visibility intervals built on the device (showo_mask_predict_next / showo_mask_mmu; reference
 * training/train.py:522-577 builds and concatenates dense [B,1,L,L] masks instead): the next showo_train_forward call with
 * mask == NULL uses iv int32 [B,L,4] for the forward and the backward; iv == NULL restores the causal default.
 * flag (optional int32[1], as written by showo_mask_predict_next): when non-zero the intervals cannot represent the mask
 * and the three losses of that forward are returned as NaN (device-side check, no host synchronisation). */
int showo_trainer_use_intervals(showo_trainer* t, const int32_t* iv, const int32_t* flag);

/* forward with saved activations.  ids int64 [B,L]; mask [B,1,L,L] fp32 or NULL (causal); labels int64 [B,L] or NULL.
 * logits_out (optional) fp32 [B,L,V]; losses_out (optional, needs labels) fp32 [3] = (loss_t2i, loss_lm, loss_mmu).
 * Masks must be interval-representable (at most two visibility runs per row: every mask the reference builds with
 * contiguous padding is); otherwise the three losses come back as NaN. */
int showo_train_forward(showo_trainer* t, const int64_t* ids, const float* mask, const int64_t* labels, int B, int L, int b_t2i,
                        int b_lm, int b_mmu, int max_seq_len, float* logits_out, float* losses_out, void* stream);
/* the same from caller-provided input embeddings fp32 [B,L,H] (reference modeling_showo.py:77-78 `inputs_embeds`; the w_clip_vit
 * training flow, training/train_w_clip_vit.py:530-613); after the backward, showo_train_input_grad returns
 * d(weighted loss)/d(embeds) fp32 [B*L*H] so that the caller's autograd graph (embed_tokens, mm_projector) can continue. */
int showo_train_forward_embeds(showo_trainer* t, const float* embeds, const float* mask, const int64_t* labels, int B, int L,
                               int b_t2i, int b_lm, int b_mmu, int max_seq_len, float* logits_out, float* losses_out, void* stream);
int showo_train_input_grad(showo_trainer* t, float* out, int64_t n, void* stream);
/* gradients of g_t2i*loss_t2i + g_lm*loss_lm + g_mmu*loss_mmu of the last forward w.r.t. every parameter */
int showo_train_backward(showo_trainer* t, const int64_t* labels, int b_t2i, int b_lm, int b_mmu, int max_seq_len, float g_t2i,
                         float g_lm, float g_mmu, void* stream);
/* the same backward in three phases (head -> blocks nL-1 .. 0 -> embedding): a data-parallel driver starts the gradient
 * exchange of a finished bucket while the next phase runs */
int showo_train_backward_head(showo_trainer* t, const int64_t* labels, int b_t2i, int b_lm, int b_mmu, int max_seq_len,
                              float g_t2i, float g_lm, float g_mmu, void* stream);
int showo_train_backward_layer(showo_trainer* t, int layer, void* stream);
int showo_train_backward_embed(showo_trainer* t, void* stream);
/* Announce the loss weights (training/train.py:600: loss = w_t2i*loss_t2i + w_lm*loss_lm + w_mmu*loss_mmu) BEFORE the forward:
 * its cross-entropy pass then also writes d(loss)/d(logits), and a showo_train_backward[_head] call with the same labels
 * pointer (contents unchanged), batch split and weights skips its own pass over the [B*L, V] fp32 logits.  Any mismatch falls
 * back to the two-pass behaviour; enable = 0 turns the announcement off. */
int showo_train_set_loss_weights(showo_trainer* t, float w_t2i, float w_lm, float w_mmu, int enable);
/* all gradients live in one flat fp32 buffer; bucket 0 = embedding, 1 + i = block i, nL + 1 = head */
int showo_train_num_buckets(showo_trainer* t);
int showo_train_bucket(showo_trainer* t, int bucket, float** ptr, int64_t* n);
/* gradient buffer of a reference state-dict key (device pointer, element count) / copy of it */
int showo_train_grad(showo_trainer* t, const char* key, float** ptr, int64_t* n);
int showo_train_grad_copy(showo_trainer* t, const char* key, float* dst, int64_t n, void* stream);
int showo_train_losses(showo_trainer* t, float* out3, void* stream);
/* optimizer (torch.optim.AdamW with the reference's two parameter groups, training/train.py:205-231): register the fp32
 * master tensor and the two moment buffers of a state-dict key once, then one call per step updates every parameter from
 * the gradients of the last backward and refreshes the engine's bf16 weight images. */
int showo_train_bind_param(showo_trainer* t, const char* key, float* param, float* exp_avg, float* exp_avg_sq, int64_t n);
int showo_train_adamw_step(showo_trainer* t, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                           void* stream);
/* torch.nn.utils.clip_grad_norm_ on a flat fp32 gradient buffer (reference training/train.py:614-615): g *= min(max_norm /
 * (||g||_2 + 1e-6), 1).  Fixed-order reduction (block partials in double, one block sums them in index order), the coefficient stays in
 * device memory: out2 (device, 2 floats) = {total norm, coefficient}; ws = showo_grad_clip_ws_doubles() doubles of scratch. */
int showo_grad_clip_norm(float* g, int64_t n, float max_norm, double* ws, float* out2, void* stream);
int showo_grad_clip_ws_doubles(void);
int showo_gelu_bf16(const uint16_t* f, uint16_t* a, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * CLIP ViT vision tower + mm_projector of the w_clip_vit understanding path.
 * Replaces transformers.CLIPVisionModel(images, output_hidden_states=True).hidden_states[-2][:, 1:] as called by
 * models/clip_encoder.py:29-49, and model.mm_projector (models/modeling_showo.py:48-53, inference_mmu.py:133-134).
 * run_layers = layers + 1 + select_layer (23 of 24 for select_layer = -2).
 * --------------------------------------------------------------------------------------------- */
typedef struct showo_clip showo_clip;
typedef struct {
    int image_size, patch_size, hidden, heads, ffn, layers, run_layers, max_batch;
    float ln_eps;
} showo_clip_config;
int showo_clip_create(const showo_clip_config* cfg, showo_clip** out);
void showo_clip_destroy(showo_clip* c);
/* one tensor by its transformers state-dict key ("vision_model.embeddings.patch_embedding.weight", ..., optionally prefixed
 * "vision_tower." as in the reference wrapper's state dict, with or without the "vision_model." level); src = device fp32 */
int showo_clip_load(showo_clip* c, const char* key, const float* src, int64_t n, void* stream);
int showo_clip_missing(const showo_clip* c);
/* images fp32 [B,3,S,S] (normalised) -> features fp32 [B, (S/patch)^2, hidden] */
int showo_clip_features(showo_clip* c, const float* images, int B, float* features, void* stream);
/* 0 (default): bf16 GEMM / attention operands.  1: accuracy mode -- every GEMM on the split-bf16 MFMA kernel (hi + lo operand pairs),
 * LayerNorm / attention / quick_gelu in fp32: the fp32 tower of transformers' CLIPVisionModel (models/clip_encoder.py:29-49 runs it in
 * the model's dtype) to ~1e-4.  The low halves of the weights are kept from every showo_clip_load; the fp32 workspace is allocated
 * by the first call with precision 1. */
int showo_clip_set_precision(showo_clip* c, int precision);
int showo_clip_get_precision(const showo_clip* c);
/* 1 when every GEMM weight of the tower has a current low half: they are allocated by the FIRST showo_clip_set_precision(c, 1) (a
 * precision-0 user never pays the 0.6 GB), so weights loaded before that call must be uploaded again. */
int showo_clip_precise_ready(const showo_clip* c);
/* mm_projector: Linear(in,out) -> exact GELU -> Linear(out,out).  Keys "0.weight", "0.bias", "2.weight", "2.bias" (optionally
 * prefixed "mm_projector."); x fp32 [T,in] -> out fp32 [T,out], T <= max_rows. */
typedef struct showo_projector showo_projector;
int showo_projector_create(int in_dim, int out_dim, int max_rows, showo_projector** out);
void showo_projector_destroy(showo_projector* p);
int showo_projector_load(showo_projector* p, const char* key, const float* src, int64_t n, void* stream);
int showo_projector_forward(showo_projector* p, const float* x, int T, float* out, void* stream);
/* 0: bf16 operands (default, also what showo_projector_backward differentiates); 1: split-bf16 GEMMs + fp32 exact GELU (inference) */
int showo_projector_set_precision(showo_projector* p, int precision);
int showo_projector_precise_ready(const showo_projector* p);  /* as showo_clip_precise_ready */
/* backward of the LAST showo_projector_forward (same T rows): dout fp32 [T,out] -> gw0 fp32 [out,in], gb0 [out], gw1 [out,out],
 * gb1 [out], dx fp32 [T,in] (optional).  Autograd of nn.Sequential(Linear, GELU(), Linear) (training/train_w_clip_vit.py trains it). */
int showo_projector_backward(showo_projector* p, const float* dout, int T, float* dx, float* gw0, float* gb0, float* gw1, float* gb1,
                             void* stream);

/* ---------------------------------------------------------------------------------------------
 * Image pre / post-processing on the device.
 * showo_image_resize_crop_normalize = training/utils.py:178-185 image_transform (torchvision Resize(bicubic) on a PIL image ->
 * CenterCrop -> ToTensor -> Normalize(0.5, 0.5)): PIL's antialiased bicubic with its 22-bit fixed-point coefficient tables
 * (computed by the host exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do), uint8 between the two passes;
 * byte-exact with PIL.  img u8 [H,W,C] (C = 1 or 3) -> out fp32 [C,crop_h,crop_w]; bounds int32 [n,2] = (first tap, tap count),
 * kk int32 [n,ksize]; tmp: >= H*Wout*C bytes; out_u8 (optional) u8 [crop_h,crop_w,C] = the resized + cropped bytes.
 * showo_images_to_uint8 = inference_t2i.py:157-159 (clamp((x+1)/2,0,1)*255, truncated, NCHW -> NHWC).
 * showo_mask_downsample_threshold = inference_t2i.py:100-108 (F.interpolate(mode='bicubic') to s_out x s_out, >= 0.5).
 * --------------------------------------------------------------------------------------------- */
int showo_image_resize_crop_normalize(const uint8_t* img, int H, int W, int C, int Hout, int Wout, const int32_t* bounds_h,
                                      const int32_t* kk_h, int ksize_h, const int32_t* bounds_v, const int32_t* kk_v, int ksize_v,
                                      int ct, int cl, int crop_h, int crop_w, int normalize, uint8_t* tmp, float* out, uint8_t* out_u8,
                                      void* stream);
int showo_images_to_uint8(const float* x, uint8_t* out, int B, int C, int H, int W, void* stream);
int showo_mask_downsample_threshold(const float* mask, int S, int s_out, uint8_t* out, float* values, void* stream);

#ifdef __cplusplus
}
#endif
#endif
