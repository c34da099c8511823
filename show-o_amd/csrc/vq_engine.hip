// MAGVIT-v2 tokenizer engine (VQGAN encoder/decoder + lookup-free quantizer) for gfx950.
// Replaces (reference): MAGVITv2.get_code / decode_code (models/modeling_magvitv2.py:423-433),
// VQGANEncoder.forward (:143-169), VQGANDecoder.forward (:365-399), ResnetBlock / AttnBlock / Upsample /
// Downsample (models/common_modules.py:27-40, 73-90, 168-211, 298-357).
//
// Activations are channel-last.  The residual carrier is fp32 [B, H*W, C]; every 3x3 conv consumes the
// bf16 GroupNorm+swish image of it through the MFMA implicit-GEMM conv (gemm.hip) and writes fp32 with the
// residual added in the epilogue.  1x1 convs (nin_shortcut, attention q/k/v/proj_out) are plain GEMMs.
#include "common.h"
#include "../../include/showo_hip.h"
#include <cmath>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

using namespace showo;

// Precision (showo_vq_config.precision): 0 = bf16 MFMA operands; 1 (the default of the Python class) = every MFMA
// operand is a (hi, lo) bf16 pair and each product is hi*hi + hi*lo + lo*hi in the fp32 accumulators, which tracks
// the reference's fp32 convolutions to ~1e-5 so that the sign-test token ids agree with it.

namespace {

enum Kind { K_CONV3 = 0, K_CONV1 = 1, K_SMALL = 2, K_VEC = 3 };

struct Tensor {
    Kind kind;
    int cout = 0, cin = 0, cin_pad = 0, ks = 1;
    void* data = nullptr;  // bf16 for CONV3/CONV1, fp32 otherwise
    bf16_t* lo = nullptr;  // split precision: low halves of a CONV3/CONV1 weight
    bool loaded = false;
};

// src fp32 [Cout, Cin, ks, ks] -> dst bf16 [Cout][ks][ks][Cin_pad]
__global__ void repack_conv_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, bf16_t* __restrict__ dlo, int Cout,
                                   int Cin, int Cpad, int ks, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int ci = (int)(i % Cpad);
    int64_t t = i / Cpad;
    int kx = (int)(t % ks);
    t /= ks;
    int ky = (int)(t % ks);
    int co = (int)(t / ks);
    float v = (ci < Cin) ? src[(((int64_t)co * Cin + ci) * ks + ky) * ks + kx] : 0.f;
    bf16_t h = f2bf(v);
    dst[i] = h;
    if (dlo) dlo[i] = f2bf(v - bf2f(h));
}
// src fp32 [Cout, Cin, ks, ks] -> dst fp32 [Cout][ks][ks][Cin]
__global__ void repack_small_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int ks, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int ci = (int)(i % Cin);
    int64_t t = i / Cin;
    int kx = (int)(t % ks);
    t /= ks;
    int ky = (int)(t % ks);
    int co = (int)(t / ks);
    dst[i] = src[(((int64_t)co * Cin + ci) * ks + ky) * ks + kx];
}

}  // namespace

#define TRY(expr)            \
    do {                     \
        int _rc = (expr);    \
        if (_rc) return _rc; \
    } while (0)

struct showo_vq {
    showo_vq_config cfg;
    std::map<std::string, Tensor> t;
    std::vector<void*> allocs;
    int64_t act_elems = 0;  // capacity of each activation buffer (elements)
    float *f0 = nullptr, *f1 = nullptr, *f2 = nullptr;
    bf16_t *b0 = nullptr, *b1 = nullptr, *b0l = nullptr, *b1l = nullptr;  // *l: low halves (split precision only)
    double* stats = nullptr;
    bool split = false;
    // attention scratch
    bf16_t *aq = nullptr, *ak = nullptr, *avt = nullptr, *ap = nullptr, *ao = nullptr;
    bf16_t *aql = nullptr, *akl = nullptr, *avtl = nullptr, *apl = nullptr, *aol = nullptr;
    float* as = nullptr;
    int attn_hw = 0;
    float *zbuf = nullptr, *zbuf2 = nullptr;

    template <class T>
    int alloc(T** p, int64_t n) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, (size_t)(n > 0 ? n : 1) * sizeof(T));
        if (e != hipSuccess) return set_error_hip(e, "hipMalloc(vq)", __FILE__, __LINE__);
        allocs.push_back(q);
        *p = (T*)q;
        return 0;
    }
    void add_conv(const std::string& name, int cout, int cin, int ks) {
        Tensor w;
        if (ks == 1 && cin <= 16) { w.kind = K_SMALL; w.cin_pad = cin; }
        else if (ks == 1) { w.kind = K_CONV1; w.cin_pad = cin; }
        else { w.kind = K_CONV3; w.cin_pad = (cin % 64) ? ((cin + 63) / 64) * 64 : cin; }
        w.cout = cout; w.cin = cin; w.ks = ks;
        t[name + ".weight"] = w;
        Tensor b;
        b.kind = K_VEC; b.cout = cout;
        t[name + ".bias"] = b;
    }
    void add_norm(const std::string& name, int c) {
        Tensor w;
        w.kind = K_VEC; w.cout = c;
        t[name + ".weight"] = w;
        t[name + ".bias"] = w;
    }
    void add_res(const std::string& p, int cin, int cout) {
        add_norm(p + ".norm1", cin);
        add_conv(p + ".conv1", cout, cin, 3);
        add_norm(p + ".norm2", cout);
        add_conv(p + ".conv2", cout, cout, 3);
        if (cin != cout) add_conv(p + ".nin_shortcut", cout, cin, 1);
    }
    void add_attn(const std::string& p, int c) {
        add_norm(p + ".norm", c);
        for (const char* n : {"q", "k", "v", "proj_out"}) add_conv(p + "." + n, c, c, 1);
    }
    const bf16_t* W(const std::string& k) { return (const bf16_t*)t[k + ".weight"].data; }
    const bf16_t* Wl(const std::string& k) { return t[k + ".weight"].lo; }
    const float* Wf(const std::string& k) { return (const float*)t[k + ".weight"].data; }
    const float* Bv(const std::string& k) { return (const float*)t[k + ".bias"].data; }
    bool has(const std::string& k) { return t.count(k) != 0; }
};

extern "C" int showo_vq_create(const showo_vq_config* c, showo_vq** out) {
    if (!c || !out) return set_error_msg(1, "vq_create: null argument");
    if (c->ch % 128) return set_error_msg(1, "vq: ch must be a multiple of 128 (GroupNorm/MFMA tiling)");
    if (c->precision != 0 && c->precision != 1) return set_error_msg(1, "vq: precision must be 0 (bf16) or 1 (split bf16)");
    showo_vq* v = new showo_vq();
    v->cfg = *c;
    v->split = c->precision == 1;
    const int ch = c->ch, zc = c->z_channels;
    // ---- encoder tensors (reference modeling_magvitv2.py:62-139)
    v->add_conv("encoder.conv_in", ch, 3, 3);
    int block_in = ch;
    for (int l = 0; l < c->enc_levels; ++l) {
        block_in = ch * (l == 0 ? 1 : c->enc_ch_mult[l - 1]);
        int block_out = ch * c->enc_ch_mult[l];
        for (int j = 0; j < c->enc_blocks[l]; ++j) {
            v->add_res("encoder.down." + std::to_string(l) + ".block." + std::to_string(j), block_in, block_out);
            block_in = block_out;
        }
        if (l != c->enc_levels - 1) v->add_conv("encoder.down." + std::to_string(l) + ".downsample.conv", block_in, block_in, 3);
    }
    v->add_res("encoder.mid.block_1", block_in, block_in);
    v->add_attn("encoder.mid.attn_1", block_in);
    v->add_res("encoder.mid.block_2", block_in, block_in);
    v->add_norm("encoder.norm_out", block_in);
    v->add_conv("encoder.conv_out", zc, block_in, 3);
    v->add_conv("encoder.quant_conv", zc, zc, 1);
    int max_c = block_in;
    // ---- decoder tensors (reference modeling_magvitv2.py:278-362)
    block_in = ch * c->dec_ch_mult[c->dec_levels - 1];
    if (block_in > max_c) max_c = block_in;
    v->add_conv("decoder.conv_in", block_in, zc, 3);
    v->add_res("decoder.mid.block_1", block_in, block_in);
    v->add_attn("decoder.mid.attn_1", block_in);
    v->add_res("decoder.mid.block_2", block_in, block_in);
    for (int l = c->dec_levels - 1; l >= 0; --l) {
        int block_out = ch * c->dec_ch_mult[l];
        for (int j = 0; j < c->dec_blocks[l]; ++j) {
            v->add_res("decoder.up." + std::to_string(l) + ".block." + std::to_string(j), block_in, block_out);
            block_in = block_out;
        }
        if (l != 0) v->add_conv("decoder.up." + std::to_string(l) + ".upsample.conv", block_in, block_in, 3);
    }
    v->add_norm("decoder.norm_out", block_in);
    v->add_conv("decoder.conv_out", 3, block_in, 3);
    v->add_conv("decoder.post_quant_conv", zc, zc, 1);
    int rc = 0;
    for (auto& kv : v->t) {
        Tensor& w = kv.second;
        if (w.kind == K_CONV3 || w.kind == K_CONV1) {
            bf16_t* p = nullptr;
            rc |= v->alloc(&p, (int64_t)w.cout * w.ks * w.ks * w.cin_pad);
            w.data = p;
            if (v->split) rc |= v->alloc(&w.lo, (int64_t)w.cout * w.ks * w.ks * w.cin_pad);
        } else if (w.kind == K_SMALL) {
            float* p = nullptr;
            rc |= v->alloc(&p, (int64_t)w.cout * w.ks * w.ks * w.cin);
            w.data = p;
        } else {
            float* p = nullptr;
            rc |= v->alloc(&p, w.cout);
            w.data = p;
        }
    }
    // ---- workspaces.  Largest activation: [B, R*R, ch * mult0] (top level of either net) or the 64-ch padded input
    const int64_t R = c->max_res;
    int top = ch * (c->enc_ch_mult[0] > c->dec_ch_mult[0] ? c->enc_ch_mult[0] : c->dec_ch_mult[0]);
    v->act_elems = (int64_t)c->max_batch * R * R * top;
    rc |= v->alloc(&v->f0, v->act_elems); rc |= v->alloc(&v->f1, v->act_elems); rc |= v->alloc(&v->f2, v->act_elems);
    rc |= v->alloc(&v->b0, v->act_elems); rc |= v->alloc(&v->b1, v->act_elems);
    if (v->split) { rc |= v->alloc(&v->b0l, v->act_elems); rc |= v->alloc(&v->b1l, v->act_elems); }
    rc |= v->alloc(&v->stats, (int64_t)showo_gn_stats_doubles(c->max_batch, (int)(R * R)));
    int levels = c->enc_levels > c->dec_levels ? c->enc_levels : c->dec_levels;
    int lat = (int)(R >> (levels - 1));
    int hw = lat * lat, hwp = ((hw + 63) / 64) * 64;
    v->attn_hw = hw;
    int64_t BP = (int64_t)c->max_batch * hw;
    rc |= v->alloc(&v->aq, BP * max_c); rc |= v->alloc(&v->ak, BP * max_c); rc |= v->alloc(&v->ao, BP * max_c);
    rc |= v->alloc(&v->avt, (int64_t)max_c * hwp); rc |= v->alloc(&v->ap, (int64_t)hw * hwp);
    rc |= v->alloc(&v->as, (int64_t)hw * hw);
    rc |= v->alloc(&v->zbuf, BP * 64); rc |= v->alloc(&v->zbuf2, BP * 64);
    if (v->split) {
        rc |= v->alloc(&v->aql, BP * max_c); rc |= v->alloc(&v->akl, BP * max_c); rc |= v->alloc(&v->aol, BP * max_c);
        rc |= v->alloc(&v->avtl, (int64_t)max_c * hwp); rc |= v->alloc(&v->apl, (int64_t)hw * hwp);
    }
    if (rc) { showo_vq_destroy(v); return rc; }
    if ((int64_t)max_c * hwp > v->act_elems) { showo_vq_destroy(v); return set_error_msg(5, "vq: workspace too small for the attention block"); }
    *out = v;
    return 0;
}

extern "C" void showo_vq_destroy(showo_vq* v) {
    if (!v) return;
    for (void* p : v->allocs) hipFree(p);
    delete v;
}

extern "C" int showo_vq_missing(const showo_vq* v) {
    if (!v) return -1;
    int m = 0;
    for (auto& kv : v->t) m += kv.second.loaded ? 0 : 1;
    return m;
}

extern "C" int showo_vq_load(showo_vq* v, const char* key, const float* src, int64_t n, void* stream) {
    if (!v || !key || !src) return set_error_msg(1, "vq_load: null argument");
    hipStream_t s = (hipStream_t)stream;
    std::string k(key);
    if (k == "quantize.embedding" || k == "quantize.power_vals") return 0;  // implicit in the bit pack (modeling_magvitv2.py:186-197)
    auto it = v->t.find(k);
    if (it == v->t.end()) return set_error_msg(3, "vq_load: unknown state-dict key");
    Tensor& w = it->second;
    if (w.kind == K_VEC) {
        if (n != w.cout) return set_error_msg(2, "vq_load: element count mismatch");
        SHOWO_CHECK_HIP(hipMemcpyAsync(w.data, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else {
        if (n != (int64_t)w.cout * w.cin * w.ks * w.ks) return set_error_msg(2, "vq_load: element count mismatch");
        if (w.kind == K_SMALL) {
            int64_t total = n;
            repack_small_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(src, (float*)w.data, w.cout, w.cin, w.ks, total);
        } else {
            int64_t total = (int64_t)w.cout * w.ks * w.ks * w.cin_pad;
            repack_conv_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(src, (bf16_t*)w.data, w.lo, w.cout, w.cin,
                                                                                           w.cin_pad, w.ks, total);
        }
        SHOWO_CHECK_HIP(hipGetLastError());
    }
    w.loaded = true;
    return 0;
}

// ---- building blocks ------------------------------------------------------------------------------------
namespace {

struct Ctx {
    showo_vq* v;
    hipStream_t s;
    int B;
    float *cur, *t1, *t2;  // fp32 activation buffers (cur holds the running activation)
    // v->stats holds the GroupNorm statistics of this fp32 tensor (written by the conv that produced it): the next gn() of it
    // skips the statistics pass.  Cleared when the statistics are consumed or the tensor is rewritten by anything else.
    const float* stats_of = nullptr;
    int stats_hw = 0, stats_c = 0;
};

// fp32 -> MFMA operand image: bf16, or the (hi, lo) pair in split precision
int to_operand(Ctx& c, const float* src, bf16_t* hi, bf16_t* lo, int64_t n) {
    if (c.v->split) return showo_split_f32_bf16(src, hi, lo, n, c.s);
    return showo_cast_f32_bf16(src, hi, n, c.s);
}

int gn(Ctx& c, const float* x, const std::string& name, bf16_t* y, bf16_t* ylo, int HW, int C, int swish) {
    const bool have = c.stats_of == x && c.stats_hw == HW && c.stats_c == C;
    c.stats_of = nullptr;
    if (!have) TRY(showo_gn_stats(x, c.v->stats, c.B, HW, C, c.s));
    return showo_gn_apply(x, c.v->stats, c.v->Wf(name), c.v->Bv(name), y, c.v->split ? ylo : nullptr, c.B, HW, C, 1e-6f, swish,
                          c.s);
}

// 3x3 conv `name` on the operand image (x, xlo): out fp32 = conv + bias (+ resid)
int conv3(Ctx& c, const bf16_t* x, const bf16_t* xlo, const std::string& name, const float* resid, float* out, int H, int W, int cin,
          int cout, int mode) {
    showo_vq* v = c.v;
    c.stats_of = nullptr;
    if (v->split && (cout % 128) == 0) {  // every such conv feeds a GroupNorm: its statistics come with it
        TRY(showo_conv3x3_bf16x3_gn(x, xlo, v->W(name), v->Wl(name), v->Bv(name), resid, out, v->stats, c.B, H, W, cin, cout, mode, c.s));
        c.stats_of = out;
        c.stats_hw = mode == 1 ? 4 * H * W : mode == 2 ? (H / 2) * (W / 2) : H * W;
        c.stats_c = cout;
        return 0;
    }
    if (v->split)
        return showo_conv3x3_bf16x3(x, xlo, v->W(name), v->Wl(name), v->Bv(name), resid, out, c.B, H, W, cin, cout, mode, c.s);
    return showo_conv3x3_bf16(x, v->W(name), v->Bv(name), resid, out, c.B, H, W, cin, cout, mode, c.s);
}

// out fp32 [M,N] (ldo) = A[M,K] * W[N,K]^T (+ bias) (+ resid); operands as images (hi, lo)
int gemm32(Ctx& c, const bf16_t* A, const bf16_t* Alo, int lda, const bf16_t* W, const bf16_t* Wlo, int ldw, const float* bias,
           int bias_per_row, float* out, int ldo, const float* resid, int M, int N, int K) {
    if (out == c.stats_of) c.stats_of = nullptr;
    if (c.v->split) return showo_gemm_bf16x3(A, Alo, lda, W, Wlo, ldw, bias, bias_per_row, out, ldo, resid, ldo, M, N, K, c.s);
    return showo_gemm_bf16(A, lda, W, ldw, bias, bias_per_row, out, ldo, resid, ldo, M, N, K,
                           resid ? SHOWO_EPI_RESID_F32 : SHOWO_EPI_F32, c.s);
}

// ResnetBlock.forward, temb=None (common_modules.py:337-357)
int resblock(Ctx& c, const std::string& p, int cin, int cout, int H, int W) {
    showo_vq* v = c.v;
    const int HW = H * W;
    TRY(gn(c, c.cur, p + ".norm1", v->b0, v->b0l, HW, cin, 1));
    TRY(conv3(c, v->b0, v->b0l, p + ".conv1", nullptr, c.t1, H, W, cin, cout, 0));
    TRY(gn(c, c.t1, p + ".norm2", v->b0, v->b0l, HW, cout, 1));
    if (cin != cout) {
        TRY(to_operand(c, c.cur, v->b1, v->b1l, (int64_t)c.B * HW * cin));
        TRY(gemm32(c, v->b1, v->b1l, cin, v->W(p + ".nin_shortcut"), v->Wl(p + ".nin_shortcut"), cin, v->Bv(p + ".nin_shortcut"), 0,
                   c.t2, cout, nullptr, c.B * HW, cout, cin));
        TRY(conv3(c, v->b0, v->b0l, p + ".conv2", c.t2, c.t2, H, W, cout, cout, 0));
        std::swap(c.cur, c.t2);
    } else {
        TRY(conv3(c, v->b0, v->b0l, p + ".conv2", c.cur, c.cur, H, W, cout, cout, 0));
    }
    return 0;
}

// AttnBlock.forward (common_modules.py:187-211): single head over H*W positions, scale C^-0.5.
// Every product goes through the fp32-output GEMM and is re-imaged as an MFMA operand (the block runs at the
// 16x16 / 32x32 latent resolution, so the extra passes are noise).
int attnblock(Ctx& c, const std::string& p, int C, int H, int W) {
    showo_vq* v = c.v;
    const int hw = H * W, hwp = ((hw + 63) / 64) * 64;
    if (hw > v->attn_hw) return set_error_msg(5, "vq: attention resolution exceeds the configured workspace");
    const int BP = c.B * hw;
    TRY(gn(c, c.cur, p + ".norm", v->b0, v->b0l, hw, C, 0));
    TRY(gemm32(c, v->b0, v->b0l, C, v->W(p + ".q"), v->Wl(p + ".q"), C, v->Bv(p + ".q"), 0, c.t1, C, nullptr, BP, C, C));
    TRY(to_operand(c, c.t1, v->aq, v->aql, (int64_t)BP * C));
    TRY(gemm32(c, v->b0, v->b0l, C, v->W(p + ".k"), v->Wl(p + ".k"), C, v->Bv(p + ".k"), 0, c.t1, C, nullptr, BP, C, C));
    TRY(to_operand(c, c.t1, v->ak, v->akl, (int64_t)BP * C));
    const float scale = 1.0f / sqrtf((float)C);
    for (int b = 0; b < c.B; ++b) {
        const int64_t ob = (int64_t)b * hw * C;
        // v^T[c][p] = sum_ci Wv[c][ci] h[p][ci] + bv[c]  (the GEMM "activation" is the weight matrix here)
        SHOWO_CHECK_HIP(hipMemsetAsync(c.t1, 0, (size_t)C * hwp * sizeof(float), c.s));  // key padding columns stay 0
        TRY(gemm32(c, v->W(p + ".v"), v->Wl(p + ".v"), C, v->b0 + ob, v->split ? v->b0l + ob : nullptr, C, v->Bv(p + ".v"), 1, c.t1,
                   hwp, nullptr, C, hw, C));
        TRY(to_operand(c, c.t1, v->avt, v->avtl, (int64_t)C * hwp));
        TRY(gemm32(c, v->aq + ob, v->split ? v->aql + ob : nullptr, C, v->ak + ob, v->split ? v->akl + ob : nullptr, C, nullptr, 0,
                   v->as, hw, nullptr, hw, hw, C));
        TRY(showo_softmax_rows_bf16(v->as, v->ap, v->split ? v->apl : nullptr, hw, hw, hwp, scale, c.s));
        TRY(gemm32(c, v->ap, v->apl, hwp, v->avt, v->avtl, hwp, nullptr, 0, c.t2, C, nullptr, hw, C, hwp));
        TRY(to_operand(c, c.t2, v->ao + ob, v->split ? v->aol + ob : nullptr, (int64_t)hw * C));
    }
    return gemm32(c, v->ao, v->aol, C, v->W(p + ".proj_out"), v->Wl(p + ".proj_out"), C, v->Bv(p + ".proj_out"), 0, c.cur, C, c.cur,
                  BP, C, C);
}

int check_vq(showo_vq* v, int B, int H, int W) {
    if (!v) return set_error_msg(1, "vq: null handle");
    if (showo_vq_missing(v) != 0) return set_error_msg(4, "vq: weights missing (showo_vq_missing() != 0)");
    if (B > v->cfg.max_batch || (int64_t)B * H * W > (int64_t)v->cfg.max_batch * v->cfg.max_res * v->cfg.max_res)
        return set_error_msg(5, "vq: batch/resolution exceeds the configured workspace");
    return 0;
}

}  // namespace

extern "C" int showo_vq_decode_code(showo_vq* v, const int64_t* ids, int B, int h, int w, float* image, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!v) return set_error_msg(1, "vq: null handle");
    const showo_vq_config& cf = v->cfg;
    const int up = 1 << (cf.dec_levels - 1);
    TRY(check_vq(v, B, h * up, w * up));
    const int zc = cf.z_channels, ch = cf.ch;
    Ctx c{v, s, B, v->f0, v->f1, v->f2};
    int H = h, W = w;
    // get_codebook_entry -> post_quant_conv (1x1, fp32) -> conv_in (modeling_magvitv2.py:208-221, 371-374)
    TRY(showo_lfq_unpack_nhwc(ids, v->zbuf, B, zc, h * w, s));
    TRY(showo_conv_small_f32(v->zbuf, v->Wf("decoder.post_quant_conv"), v->Bv("decoder.post_quant_conv"), v->zbuf2, B, H, W, zc, zc, 1, s));
    TRY(showo_pad_cast_bf16(v->zbuf2, v->b0, v->split ? v->b0l : nullptr, (int64_t)B * H * W, zc, 64, s));
    int block_in = ch * cf.dec_ch_mult[cf.dec_levels - 1];
    TRY(conv3(c, v->b0, v->b0l, "decoder.conv_in", nullptr, c.cur, H, W, 64, block_in, 0));
    TRY(resblock(c, "decoder.mid.block_1", block_in, block_in, H, W));
    TRY(attnblock(c, "decoder.mid.attn_1", block_in, H, W));
    TRY(resblock(c, "decoder.mid.block_2", block_in, block_in, H, W));
    for (int l = cf.dec_levels - 1; l >= 0; --l) {
        int block_out = ch * cf.dec_ch_mult[l];
        for (int j = 0; j < cf.dec_blocks[l]; ++j) {
            TRY(resblock(c, "decoder.up." + std::to_string(l) + ".block." + std::to_string(j), block_in, block_out, H, W));
            block_in = block_out;
        }
        if (l != 0) {  // nearest 2x + conv (common_modules.py:36-40), fused into the conv's gather
            std::string n = "decoder.up." + std::to_string(l) + ".upsample.conv";
            TRY(to_operand(c, c.cur, v->b1, v->b1l, (int64_t)B * H * W * block_in));
            TRY(conv3(c, v->b1, v->b1l, n, nullptr, c.t1, H, W, block_in, block_in, 1));
            std::swap(c.cur, c.t1);
            H *= 2; W *= 2;
        }
    }
    TRY(gn(c, c.cur, "decoder.norm_out", v->b0, v->b0l, H * W, block_in, 1));
    TRY(conv3(c, v->b0, v->b0l, "decoder.conv_out", nullptr, c.t1, H, W, block_in, 3, 0));
    return showo_nhwc_to_nchw_f32(c.t1, image, B, 3, H * W, s);
}

extern "C" int showo_vq_get_code(showo_vq* v, const float* pixels, int B, int Hi, int Wi, int64_t* ids, float* z_out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!v) return set_error_msg(1, "vq: null handle");
    const showo_vq_config& cf = v->cfg;
    TRY(check_vq(v, B, Hi, Wi));
    const int down = 1 << (cf.enc_levels - 1);
    if (Hi % down || Wi % down) return set_error_msg(1, "vq: image size must be a multiple of 2^(levels-1)");
    const int zc = cf.z_channels, ch = cf.ch;
    Ctx c{v, s, B, v->f0, v->f1, v->f2};
    int H = Hi, W = Wi;
    TRY(showo_nchw_to_nhwc_f32(pixels, c.t1, B, 3, H * W, s));
    TRY(showo_pad_cast_bf16(c.t1, v->b0, v->split ? v->b0l : nullptr, (int64_t)B * H * W, 3, 64, s));
    TRY(conv3(c, v->b0, v->b0l, "encoder.conv_in", nullptr, c.cur, H, W, 64, ch, 0));
    int block_in = ch;
    for (int l = 0; l < cf.enc_levels; ++l) {
        int block_out = ch * cf.enc_ch_mult[l];
        for (int j = 0; j < cf.enc_blocks[l]; ++j) {
            TRY(resblock(c, "encoder.down." + std::to_string(l) + ".block." + std::to_string(j), block_in, block_out, H, W));
            block_in = block_out;
        }
        if (l != cf.enc_levels - 1) {  // pad (0,1,0,1) + stride-2 conv (common_modules.py:83-88), fused into the gather
            std::string n = "encoder.down." + std::to_string(l) + ".downsample.conv";
            TRY(to_operand(c, c.cur, v->b1, v->b1l, (int64_t)B * H * W * block_in));
            TRY(conv3(c, v->b1, v->b1l, n, nullptr, c.t1, H, W, block_in, block_in, 2));
            std::swap(c.cur, c.t1);
            H /= 2; W /= 2;
        }
    }
    TRY(resblock(c, "encoder.mid.block_1", block_in, block_in, H, W));
    TRY(attnblock(c, "encoder.mid.attn_1", block_in, H, W));
    TRY(resblock(c, "encoder.mid.block_2", block_in, block_in, H, W));
    TRY(gn(c, c.cur, "encoder.norm_out", v->b0, v->b0l, H * W, block_in, 1));
    TRY(conv3(c, v->b0, v->b0l, "encoder.conv_out", nullptr, v->zbuf, H, W, block_in, zc, 0));
    TRY(showo_conv_small_f32(v->zbuf, v->Wf("encoder.quant_conv"), v->Bv("encoder.quant_conv"), v->zbuf2, B, H, W, zc, zc, 1, s));
    // LFQuantizer sign-pack (modeling_magvitv2.py:201-206, 239-241)
    TRY(showo_lfq_pack_nhwc(v->zbuf2, ids, B, zc, H * W, zc, s));
    if (z_out) TRY(showo_nhwc_to_nchw_f32(v->zbuf2, z_out, B, zc, H * W, s));
    return 0;
}
