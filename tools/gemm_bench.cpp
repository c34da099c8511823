// Standalone GEMM A/B harness (no torch): drives libshowo_hip.so through its C ABI.
//   build: hipcc --offload-arch=gfx950 -O2 tools/gemm_bench.cpp -o tools/gemm_bench -Lshow-o_amd -lshowo_hip -Wl,-rpath,'$ORIGIN/../show-o_amd'
//   run:   tools/gemm_bench [impl list, e.g. 2,3]
// For every shape: each implementation is checked against implementation 1 (the 128^2 kernel, itself checked against
// the oracle by tests/test_kernels_gpu.py) on the full output, 3 repeats (race screen), then timed on rotating weight
// buffers (weights stream from HBM like in the 24-layer stack), random normal operands.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cmath>
#include <vector>
#include <string>
#include "../include/showo_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
#define RC(x) do { int r_ = (x); if (r_) { printf("showo error %d: %s (line %d)\n", r_, showo_last_error(), __LINE__); exit(3); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t v) { uint32_t u = ((uint32_t)v) << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void fill_kernel(uint16_t* p, size_t n, uint32_t seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
        uint32_t y = x * 1664525u + 1013904223u; y ^= y >> 15;
        float u1 = ((x >> 8) + 0.5f) / 16777216.f, u2 = ((y >> 8) + 0.5f) / 16777216.f;
        float v = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2) * scale;  // N(0, scale)
        uint32_t b = __float_as_uint(v); b += 0x7fffu + ((b >> 16) & 1u);
        p[i] = (uint16_t)(b >> 16);
    }
}

// LDS read-rate probe (mode 6): cycles per wave-instruction of ds_read_b128 vs ds_read_b64_tr_b16 at the conflict-free layouts the
// GEMMs use, with NW waves of one block issuing them back to back (16 reads in flight, then s_waitcnt): what one wave / one CU sustains.
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 probe_bf16x4;
typedef float probe_f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void lds_probe_kernel(unsigned long long* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(sm)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    unsigned acc = 0;
    // MODE 0: b128, row fr of a 16-row fragment (128-B rows), chunk fg ^ (fr & 7): gemm2p's read
    const unsigned a0 = wave * 4096 + fr * 128 + ((fg ^ (fr & 7)) << 4);
    // MODE 1: transposing read, gemm_tn's addressing (piece = 1 KiB: 8 rows x 128 B)
    const int swz = (((fr >> 3) & 1) << 1) | ((fg & 1) << 2);
    const unsigned a1 = wave * 4096 + fg * 1024 + (fr >> 2) * 128 + (fr & 1) * 8 + (((0 + ((fr >> 1) & 1)) ^ swz) << 4);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        unsigned b0 = a0, b1 = a1;
        asm volatile("" : "+v"(b0), "+v"(b1));  // the addresses are opaque per iteration: no hoisting of the reads
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                probe_f4 v = *reinterpret_cast<const probe_f4*>(sm + ((b0 + k * 2048) & 65535));
                acc += __float_as_uint(v[0]) ^ __float_as_uint(v[3]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                probe_bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) probe_bf16x4*)(sm + ((b1 + (k & 1) * 512 + (k >> 1) * 4096) & 65535)));
                acc += (unsigned)__builtin_bit_cast(unsigned long long, v);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[wave * 2] = t1 - t0; out[wave * 2 + 1] = acc; }
}

struct Shape { int M, N, K, epi; const char* name; };

int main(int argc, char** argv) {
    // variants: "impl[:gn[:flags]]" separated by ','  e.g.  2,3:1,3:4,3:8,3:4:1
    struct Var { int impl, gn, flags, bm; };
    std::vector<Var> impls = {{4, 8, 0, 0}, {5, 8, 0, 0}};
    if (argc > 1) {
        impls.clear();
        for (char* t = strtok(argv[1], ","); t; t = strtok(nullptr, ",")) {
            Var v{0, 8, 0, 0};
            sscanf(t, "%d:%d:%d:%d", &v.impl, &v.gn, &v.flags, &v.bm);
            impls.push_back(v);
        }
    }
    int quick = argc > 2 ? atoi(argv[2]) : 0;
    int dbg = argc > 3 ? atoi(argv[3]) : 0;
    unsigned long long* dbgbuf = nullptr;
    if (dbg) CK(hipMalloc(&dbgbuf, 512 * 8));
    std::vector<Shape> shapes = {
        {300, 256, 64, 2, "edge-small"}, {1100, 520, 192, 2, "edge-ragged"}, {1548, 2048, 256, 3, "ragged-resid"},
        {6192, 6144, 2048, 0, "qkv"}, {6192, 2048, 2048, 3, "dense"}, {6192, 8192, 2048, 1, "fc1+gelu"},
        {6192, 2048, 8192, 3, "fc2"}, {4096, 8192, 2048, 2, "lm_head rows"}, {6192, 14336, 2048, 0, "qkv|fc1 fused"},
        {6192, 2048, 10240, 3, "dense|fc2 fused"}, {4128, 6144, 2048, 0, "qkv act258"}, {4128, 2048, 2048, 3, "dense act258"}, {4128, 8192, 2048, 1, "fc1 act258"}, {4128, 2048, 8192, 3, "fc2 act258"}, {4128, 2048, 10240, 3, "dense|fc2 act258"}, {4128, 14336, 2048, 0, "qkv|fc1 act258"}, {9240, 6144, 2048, 0, "qkv L1155"}, {9240, 2048, 8192, 3, "fc2 L1155"}, {4096, 4096, 4096, 0, "4096^3"}, {8192, 8192, 8192, 0, "8192^3"},
    };
    if (quick == 2) {  // the shapes of the t2i pipeline (steps 1..17: 4128 rows; step 0: 6192 rows), separate and fused
        shapes = {
            {300, 256, 64, 2, "edge-small"}, {1100, 520, 192, 2, "edge-ragged"}, {1548, 2048, 256, 3, "ragged-resid"},
            {4128, 6144, 2048, 0, "qkv act258"}, {4128, 8192, 2048, 1, "fc1 act258"}, {4128, 14336, 2048, 0, "qkv|fc1 act258"},
            {4128, 2048, 2048, 3, "dense act258"}, {4128, 2048, 8192, 3, "fc2 act258"}, {4128, 2048, 10240, 3, "dense|fc2 act258"},
            {6192, 14336, 2048, 0, "qkv|fc1 full"}, {6192, 2048, 10240, 3, "dense|fc2 full"},
            {4096, 8192, 2048, 2, "lm_head rows"}, {4096, 4096, 4096, 0, "4096^3"}, {8192, 8192, 8192, 0, "8192^3"},
        };
    }
    if (quick == 4) {  // small-M shapes: the CLIP ViT-L/14-336 tower (577 rows) and the cfg4 prefill (631 rows)
        shapes = {
            {577, 3072, 1024, 0, "clip qkv"}, {577, 1024, 1024, 3, "clip out"}, {577, 4096, 1024, 1, "clip fc1"}, {577, 1024, 4096, 3, "clip fc2"},
            {631, 6144, 2048, 0, "prefill qkv"}, {631, 2048, 2048, 3, "prefill dense"}, {631, 8192, 2048, 1, "prefill fc1"}, {631, 2048, 8192, 3, "prefill fc2"},
            {631, 2048, 10240, 3, "prefill dense|fc2"}, {631, 14336, 2048, 0, "prefill qkv|fc1"},
        };
    }
    if (quick == 7) {  // row-range split of the half-empty dense|fc2 launch (DESIGN "what comes next" 2a): 4128 rows as ONE launch (208 tiles, no
                       // split) against rows [0, 4096) (128 tiles x 2 k-halves = 256 blocks on the cooperative reduction) + the 32-row tail
        shapes = {
            {4128, 2048, 10240, 3, "dense|fc2 4128"}, {4096, 2048, 10240, 3, "rows 0..4095"}, {32, 2048, 10240, 3, "rows 4096..4127"},
            {6192, 2048, 10240, 3, "dense|fc2 6192"}, {6144, 2048, 10240, 3, "rows 0..6143"}, {48, 2048, 10240, 3, "rows 6144..6191"},
        };
    }
    if (quick == 8) {  // half chip vs full chip at 256-row tiles, K = 8192: with GEMM_BENCH_ZERO=1 (all-zero operands: far less switching power,
                       // same instruction stream) this separates a clock / power bound from a shared-bandwidth bound on the k-tile time
        shapes = {
            {4096, 2048, 8192, 0, "128 tiles (half chip)"}, {4096, 4096, 8192, 0, "256 tiles (1 round)"}, {8192, 4096, 8192, 0, "512 tiles (2 rounds)"},
        };
    }
    if (quick == 3) {  // decomposition of the tile time at 192-row tiles (variant x192): exactly 256 tiles per round at N = 2048
        shapes = {
            {6144, 2048, 2048, 0, "1 round K2048"}, {6144, 2048, 4096, 0, "1 round K4096"}, {6144, 2048, 8192, 0, "1 round K8192"},
            {12288, 2048, 2048, 0, "2 rounds K2048"}, {24576, 2048, 2048, 0, "4 rounds K2048"}, {6144, 8192, 2048, 0, "4 rounds(N) K2048"},
            {6144, 14336, 2048, 0, "7 rounds(N) K2048"},
        };
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (quick == 6) {  // LDS read-rate probe
        unsigned long long* d; CK(hipMalloc(&d, 64 * 8));
        const int iters = 4096;
        for (int mode = 0; mode < 2; ++mode)
            for (int nw : {1, 4, 8}) {
                CK(hipMemset(d, 0, 64 * 8));
                if (mode == 0) lds_probe_kernel<0><<<1, nw * 64, 65536, st>>>(d, iters);
                else lds_probe_kernel<1><<<1, nw * 64, 65536, st>>>(d, iters);
                CK(hipStreamSynchronize(st));
                unsigned long long h[16]; CK(hipMemcpy(h, d, 16 * 8, hipMemcpyDeviceToHost));
                unsigned long long mx = 0; for (int w = 0; w < nw; ++w) mx = std::max(mx, h[2 * w]);
                const double per = (double)mx / (iters * 16.0);
                printf("%-22s %d wave(s): %.1f cycles per wave-instruction (counter units), %.1f B per counter cycle per CU\n",
                       mode == 0 ? "ds_read_b128" : "ds_read_b64_tr_b16", nw, per, nw * (mode == 0 ? 1024.0 : 512.0) / per);
            }
        return 0;
    }
    if (quick == 5) {  // weight-gradient shapes: showo_gemm_tn_bf16 on token-major operands vs two transposes + the k-contiguous GEMM
        struct TS { int M, N, T; const char* name; };
        const TS ts[] = {{256, 256, 200, "edge 1 tile"}, {264, 520, 1000, "edge ragged"}, {512, 768, 2100, "split-K"}, {2048, 8192, 11223, "dW2"},
                         {8192, 2048, 11223, "dW1"}, {6144, 2048, 11223, "dWqkv"}, {2048, 2048, 11223, "dWd"}, {58498, 2048, 11223, "dWlm"}};
        for (const TS& s : ts) {
            const int Tp = ((s.T + 63) / 64) * 64, lda = ((s.M + 63) / 64) * 64, ldb = s.N;
            uint16_t *A, *B, *At, *Bt; float *o_tn, *o_nt;
            CK(hipMalloc(&A, (size_t)Tp * lda * 2)); CK(hipMalloc(&B, (size_t)Tp * ldb * 2));  // rows_padded: readable up to Tp
            CK(hipMemsetAsync(A, 0xff, (size_t)Tp * lda * 2, st)); CK(hipMemsetAsync(B, 0xff, (size_t)Tp * ldb * 2, st));  // NaN bit patterns behind row T
            CK(hipMalloc(&At, (size_t)lda * Tp * 2)); CK(hipMalloc(&Bt, (size_t)s.N * Tp * 2));
            CK(hipMalloc(&o_tn, (size_t)s.M * s.N * 4)); CK(hipMalloc(&o_nt, (size_t)s.M * s.N * 4));
            fill_kernel<<<1024, 256, 0, st>>>(A, (size_t)s.T * lda, 5u, 1.0f);
            fill_kernel<<<1024, 256, 0, st>>>(B, (size_t)s.T * ldb, 6u, 0.5f);
            CK(hipMemsetAsync(o_tn, 0xff, (size_t)s.M * s.N * 4, st));
            RC(showo_gemm_set_impl(0));
            auto nt = [&]() {
                RC(showo_transpose_bf16(A, lda, At, s.T, lda, Tp, 0, nullptr, nullptr, 0, st));
                RC(showo_transpose_bf16(B, ldb, Bt, s.T, s.N, Tp, 0, nullptr, nullptr, 0, st));
                RC(showo_gemm_bf16(At, Tp, Bt, Tp, nullptr, 0, o_nt, s.N, nullptr, 0, s.M, s.N, Tp, 2, st));
            };
            auto tn = [&]() { RC(showo_gemm_tn_bf16(A, lda, B, ldb, o_tn, s.N, s.M, s.N, s.T, 0, 1, st)); };
            nt(); tn();
            CK(hipStreamSynchronize(st));
            std::vector<float> h0((size_t)s.M * s.N), h1((size_t)s.M * s.N);
            CK(hipMemcpy(h0.data(), o_nt, h0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), o_tn, h1.size() * 4, hipMemcpyDeviceToHost));
            double maxd = 0, maxr = 0; size_t bad = 0, same = 0;
            for (size_t i = 0; i < h0.size(); ++i) {
                const double d = fabs((double)h0[i] - h1[i]);
                if (!(d <= 1e-3 * (1 + fabs(h0[i])))) bad++;
                if (h0[i] == h1[i]) same++;
                maxd = std::max(maxd, d); maxr = std::max(maxr, (double)fabs(h0[i]));
            }
            float ms_nt, ms_tn, ms_g;
            const int iters = 8;
            for (int i = 0; i < 2; ++i) { nt(); tn(); }
            CK(hipEventRecord(e0, st)); for (int i = 0; i < iters; ++i) nt(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_nt, e0, e1));
            CK(hipEventRecord(e0, st)); for (int i = 0; i < iters; ++i) tn(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_tn, e0, e1));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) RC(showo_gemm_bf16(At, Tp, Bt, Tp, nullptr, 0, o_nt, s.N, nullptr, 0, s.M, s.N, Tp, 2, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_g, e0, e1));
            const double fl = 2.0 * s.M * s.N * s.T * iters / 1e9;
            printf("%-12s M=%5d N=%5d T=%5d | transposes + NT %7.1f us (GEMM alone %7.1f us, %6.1f TF) | TN %7.1f us %6.1f TF | max|d| %.3g of %.3g, equal bits %.4f %s\n",
                   s.name, s.M, s.N, s.T, ms_nt / iters * 1e3, ms_g / iters * 1e3, fl / ms_g, ms_tn / iters * 1e3, fl / ms_tn, maxd, maxr,
                   (double)same / h0.size(), bad ? "**MISMATCH**" : "");
            fflush(stdout);
            CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(At)); CK(hipFree(Bt)); CK(hipFree(o_tn)); CK(hipFree(o_nt));
        }
        return 0;
    }
    for (const Shape& s : shapes) {
        if (quick == 1 && (size_t)s.M * s.N * s.K > (size_t)6192 * 14336 * 2048) continue;
        const size_t nA = (size_t)s.M * s.K, nW = (size_t)s.N * s.K, nO = (size_t)s.M * s.N;
        int R = (int)std::max<size_t>(1, std::min<size_t>(8, ((size_t)1 << 30) / (nW * 2)));
        uint16_t *A, *W; float *bias, *resid; void *out_ref, *out;
        const bool f32 = s.epi >= 2;
        CK(hipMalloc(&A, nA * 2)); CK(hipMalloc(&W, nW * 2 * R)); CK(hipMalloc(&bias, s.N * 4)); CK(hipMalloc(&resid, nO * 4));
        CK(hipMalloc(&out_ref, nO * 4)); CK(hipMalloc(&out, nO * 4));
        fill_kernel<<<1024, 256, 0, st>>>(A, nA, 1u, 1.0f);
        fill_kernel<<<1024, 256, 0, st>>>(W, nW * R, 2u, 0.02f);
        if (getenv("GEMM_BENCH_ZERO") && atoi(getenv("GEMM_BENCH_ZERO"))) { CK(hipMemsetAsync(A, 0, nA * 2, st)); CK(hipMemsetAsync(W, 0, nW * 2 * R, st)); }
        {   // bias / resid: reuse the generator through a bf16 temp is overkill; small host fill
            std::vector<float> hb(s.N); for (int i = 0; i < s.N; ++i) hb[i] = 0.01f * ((i * 37) % 101 - 50);
            CK(hipMemcpy(bias, hb.data(), s.N * 4, hipMemcpyHostToDevice));
            CK(hipMemsetAsync(resid, 0, nO * 4, st));
        }
        CK(hipStreamSynchronize(st));
        auto run = [&](Var v, void* o, int r) {
            RC(showo_gemm_set_impl(v.impl));
            RC(showo_gemm_tune(v.gn, v.flags | (v.bm << 8), nullptr));
            // in-place residual like the engine when epi == 3: resid := o would accumulate across repeats, so use `resid`
            RC(showo_gemm_bf16(A, s.K, W + (size_t)r * nW, s.K, bias, 0, o, s.N, s.epi == 3 ? resid : nullptr, s.N, s.M, s.N, s.K, s.epi, st));
        };
        run(Var{1, 1, 0, 0}, out_ref, 0);
        CK(hipStreamSynchronize(st));
        std::vector<float> href(f32 ? nO : 0); std::vector<uint16_t> hrefb(f32 ? 0 : nO);
        if (f32) CK(hipMemcpy(href.data(), out_ref, nO * 4, hipMemcpyDeviceToHost)); else CK(hipMemcpy(hrefb.data(), out_ref, nO * 2, hipMemcpyDeviceToHost));
        printf("%-18s M=%5d N=%5d K=%5d epi=%d  tiles256=%4d |", s.name, s.M, s.N, s.K, s.epi, ((s.M + 255) / 256) * ((s.N + 255) / 256));
        for (Var impl : impls) {
            double maxd = 0, maxr = 0; size_t bad = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(out, 0xff, nO * (f32 ? 4 : 2), st));
                run(impl, out, 0);
                CK(hipStreamSynchronize(st));
                if (f32) {
                    std::vector<float> h(nO); CK(hipMemcpy(h.data(), out, nO * 4, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < nO; ++i) { double d = fabs((double)h[i] - href[i]); if (!(d <= 1e-3 * (1 + fabs(href[i])))) bad++; if (d > maxd) maxd = d; if (fabs(href[i]) > maxr) maxr = fabs(href[i]); }
                } else {
                    std::vector<uint16_t> h(nO); CK(hipMemcpy(h.data(), out, nO * 2, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < nO; ++i) { double a = bf2f(h[i]), b = bf2f(hrefb[i]); double d = fabs(a - b); if (!(d <= 1.6e-2 * (fabs(b) + 1e-2))) bad++; if (d > maxd) maxd = d; if (fabs(b) > maxr) maxr = fabs(b); }
                }
            }
            // timing
            const int iters = (size_t)s.M * s.N * s.K > ((size_t)1 << 36) ? 6 : 16;
            for (int i = 0; i < 3; ++i) run(impl, out, i % R);
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) run(impl, out, i % R);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double tf = 2.0 * s.M * s.N * s.K * iters / (ms * 1e-3) / 1e12;
            printf(" v%d:%d:%d:%d %7.1f TF %6.3f ms %s|", impl.impl, impl.gn, impl.flags, impl.bm, tf, ms / iters, bad ? "**MISMATCH** " : "");
            (void)maxd; (void)maxr;
            fflush(stdout);
        }
        printf("\n");
        if (dbg && s.epi == 0 && s.M == 4096 && s.K == 4096) {
            for (Var v : impls) {
                if (v.impl < 3 || (v.flags >> 4)) continue;
                RC(showo_gemm_set_impl(v.impl));
                RC(showo_gemm_tune(v.gn, v.flags | (v.bm << 8), dbgbuf));
                CK(hipMemsetAsync(dbgbuf, 0, 512 * 8, st));
                RC(showo_gemm_bf16(A, s.K, W, s.K, bias, 0, out, s.N, nullptr, s.N, s.M, s.N, s.K, 0, st));
                CK(hipStreamSynchronize(st));
                std::vector<unsigned long long> h(512);
                CK(hipMemcpy(h.data(), dbgbuf, 512 * 8, hipMemcpyDeviceToHost));
                printf("  timestamps (cycles since first, per barrier exit; k-tiles 8,9) gn=%d flags=%d\n", v.gn, v.flags);
                unsigned long long t0 = h[0];
                for (int w = 0; w < 8; w += 4) {
                    printf("   wave %d:", w);
                    for (int i = 0; i < 16; ++i) printf(" %5lld", (long long)(h[w * 64 + i] - t0));
                    printf("\n   delta :     ");
                    for (int i = 1; i < 16; ++i) printf(" %5lld", (long long)(h[w * 64 + i] - h[w * 64 + i - 1]));
                    printf("\n");
                }
                RC(showo_gemm_tune(v.gn, v.flags | (v.bm << 8), nullptr));
            }
        }
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(resid)); CK(hipFree(out_ref)); CK(hipFree(out));
    }

    // ---- fused entry points: K-concatenated residual GEMM and the [Wqkv ; W1] projection, checked against the separate launches
    if (quick != 3 && quick != 4 && quick < 7) for (int M : {4128, 6192, 700}) {
        const int H = 2048, F = 8192, nH = 32, B = M == 700 ? 2 : 16, L = M / B, Lp = ((L + 63) / 64) * 64;
        const int Mx = B * L;
        uint16_t *attn, *ffn, *Wd, *W2, *Wcat, *h, *Wq1, *Q0, *K0, *V0, *Q1, *K1, *V1, *f0, *f1;
        float *x0, *x1, *bd, *b2, *bsum, *bq1, *lnp, *cosT, *sinT;
        CK(hipMalloc(&attn, (size_t)Mx * H * 2)); CK(hipMalloc(&ffn, (size_t)Mx * F * 2)); CK(hipMalloc(&Wd, (size_t)H * H * 2)); CK(hipMalloc(&W2, (size_t)H * F * 2));
        const int RW = 8; CK(hipMalloc(&Wcat, (size_t)RW * H * (H + F) * 2)); CK(hipMalloc(&x0, (size_t)Mx * H * 4)); CK(hipMalloc(&x1, (size_t)Mx * H * 4));
        CK(hipMalloc(&bd, H * 4)); CK(hipMalloc(&b2, H * 4)); CK(hipMalloc(&bsum, H * 4));
        fill_kernel<<<1024, 256, 0, st>>>(attn, (size_t)Mx * H, 11u, 1.0f);
        fill_kernel<<<1024, 256, 0, st>>>(ffn, (size_t)Mx * F, 12u, 1.0f);
        fill_kernel<<<1024, 256, 0, st>>>(Wd, (size_t)H * H, 13u, 0.02f);
        fill_kernel<<<1024, 256, 0, st>>>(W2, (size_t)H * F, 14u, 0.02f);
        CK(hipMemcpy2DAsync(Wcat, (size_t)(H + F) * 2, Wd, (size_t)H * 2, (size_t)H * 2, H, hipMemcpyDeviceToDevice, st));
        CK(hipMemcpy2DAsync(Wcat + H, (size_t)(H + F) * 2, W2, (size_t)F * 2, (size_t)F * 2, H, hipMemcpyDeviceToDevice, st));
        for (int r = 1; r < RW; ++r) CK(hipMemcpyAsync(Wcat + (size_t)r * H * (H + F), Wcat, (size_t)H * (H + F) * 2, hipMemcpyDeviceToDevice, st));
        std::vector<float> hb(H), hb2(H), hs(H);
        for (int i = 0; i < H; ++i) { hb[i] = 0.01f * ((i * 37) % 101 - 50); hb2[i] = 0.02f * ((i * 53) % 89 - 44); hs[i] = hb[i] + hb2[i]; }
        CK(hipMemcpy(bd, hb.data(), H * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b2, hb2.data(), H * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(bsum, hs.data(), H * 4, hipMemcpyHostToDevice));
        std::vector<float> hx((size_t)Mx * H);
        for (size_t i = 0; i < hx.size(); ++i) hx[i] = 0.001f * (float)((i * 2654435761u) % 2001) - 1.0f;
        RC(showo_gemm_set_impl(0)); RC(showo_gemm_tune(8, 0, nullptr));
        // reference: x += attn Wd^T + bd ; x += ffn W2^T + b2 with the 128^2 kernel
        CK(hipMemcpy(x0, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        RC(showo_gemm_set_impl(1));
        RC(showo_gemm_bf16(attn, H, Wd, H, bd, 0, x0, H, x0, H, Mx, H, H, 3, st));
        RC(showo_gemm_bf16(ffn, F, W2, F, b2, 0, x0, H, x0, H, Mx, H, F, 3, st));
        RC(showo_gemm_set_impl(0));
        CK(hipStreamSynchronize(st));
        std::vector<float> r0((size_t)Mx * H), r1((size_t)Mx * H);
        CK(hipMemcpy(r0.data(), x0, r0.size() * 4, hipMemcpyDeviceToHost));
        // GEMM_BENCH_VARS / GEMM_BENCH_GNS: comma lists overriding the tile variants and the tile-group widths of the fused sections
        std::vector<int> vars = {3192, 4192, 3176, 4176, 3160, 4160, 3144, 4144}, gns = {8};
        auto parse = [](const char* e, std::vector<int>& v) { if (!e) return; v.clear(); std::string t(e); size_t p0 = 0; while (p0 < t.size()) { size_t q = t.find(',', p0); if (q == std::string::npos) q = t.size(); v.push_back(atoi(t.substr(p0, q - p0).c_str())); p0 = q + 1; } };
        parse(getenv("GEMM_BENCH_VARS"), vars); parse(getenv("GEMM_BENCH_GNS"), gns);
        uint16_t* WcatT; CK(hipMalloc(&WcatT, (size_t)RW * showo_gemm_tiled_elems(H, H + F) * 2));
        for (int r = 0; r < RW; ++r) RC(showo_gemm_tile_weight(Wcat, H + F, H, H + F, WcatT + (size_t)r * showo_gemm_tiled_elems(H, H + F), st));
        for (int tl : {0, 1}) {
        const uint16_t* Wk = tl ? WcatT : Wcat; const size_t wstride = tl ? (size_t)showo_gemm_tiled_elems(H, H + F) : (size_t)H * (H + F);
        printf("kcat M=%d tiled=%d:", Mx, tl);
        for (int v : vars) for (int gn : gns) {
            RC(showo_gemm_tune(gn, (v << 8) | 4, nullptr));
            size_t bad = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemcpy(x1, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
                RC(showo_gemm_kcat_bf16(attn, H, H, ffn, F, F, Wk, H + F, bsum, x1, H, x1, H, Mx, H, 3, tl, st));
                CK(hipStreamSynchronize(st));
                CK(hipMemcpy(r1.data(), x1, r1.size() * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < r0.size(); ++i) if (!(fabs((double)r0[i] - r1[i]) <= 2e-3 * (1 + fabs(r0[i])))) bad++;
            }
            const int iters = 10;
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) RC(showo_gemm_kcat_bf16(attn, H, H, ffn, F, F, Wk + (size_t)(i % RW) * wstride, H + F, bsum, x1, H, x1, H, Mx, H, 3, tl, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %d/gn%d:%.0fTF%s", v, gn, 2.0 * Mx * H * (H + F) * iters / (ms * 1e-3) / 1e12, bad ? "**MISMATCH**" : "");
            fflush(stdout);
        }
        printf("\n");
        }
        // [Wqkv ; W1]: bitwise against showo_gemm_qkv_bf16 + the GELU GEMM at the same tile variant
        const int Nq = 3 * H;
        CK(hipMalloc(&h, (size_t)Mx * H * 2)); CK(hipMalloc(&Wq1, (size_t)RW * (Nq + F) * H * 2)); CK(hipMalloc(&bq1, (Nq + F) * 4));
        CK(hipMalloc(&lnp, 4 * 64 * 4)); CK(hipMalloc(&cosT, 2048 * 32 * 4)); CK(hipMalloc(&sinT, 2048 * 32 * 4));
        const size_t nqk = (size_t)B * nH * L * 64, nvt = (size_t)B * nH * 64 * Lp;
        CK(hipMalloc(&Q0, nqk * 2)); CK(hipMalloc(&K0, nqk * 2)); CK(hipMalloc(&V0, nvt * 2)); CK(hipMalloc(&Q1, nqk * 2)); CK(hipMalloc(&K1, nqk * 2)); CK(hipMalloc(&V1, nvt * 2));
        CK(hipMalloc(&f0, (size_t)Mx * F * 2)); CK(hipMalloc(&f1, (size_t)Mx * F * 2));
        fill_kernel<<<1024, 256, 0, st>>>(h, (size_t)Mx * H, 21u, 1.0f);
        fill_kernel<<<1024, 256, 0, st>>>(Wq1, (size_t)(Nq + F) * H, 22u, 0.02f);
        for (int r = 1; r < RW; ++r) CK(hipMemcpyAsync(Wq1 + (size_t)r * (Nq + F) * H, Wq1, (size_t)(Nq + F) * H * 2, hipMemcpyDeviceToDevice, st));
        std::vector<float> hq(Nq + F), hl(256), hc(2048 * 32), hsn(2048 * 32);
        for (int i = 0; i < Nq + F; ++i) hq[i] = 0.01f * ((i * 37) % 101 - 50);
        for (int i = 0; i < 256; ++i) hl[i] = (i & 64) ? 0.01f * (i % 7) : 1.0f + 0.01f * (i % 5);  // qw | qb | kw | kb
        for (int p = 0; p < 2048; ++p) for (int j = 0; j < 32; ++j) { double a = p * pow(10000.0, -2.0 * (j % 16) / 32.0); hc[p * 32 + j] = (float)cos(a); hsn[p * 32 + j] = (float)sin(a); }
        CK(hipMemcpy(bq1, hq.data(), hq.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(lnp, hl.data(), 1024, hipMemcpyHostToDevice));
        CK(hipMemcpy(cosT, hc.data(), hc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sinT, hsn.data(), hsn.size() * 4, hipMemcpyHostToDevice));
        uint16_t* Wq1T; const size_t q1e = (size_t)showo_gemm_tiled_elems(Nq + F, H); CK(hipMalloc(&Wq1T, (size_t)RW * q1e * 2));
        for (int r = 0; r < RW; ++r) RC(showo_gemm_tile_weight(Wq1, H, Nq + F, H, Wq1T + (size_t)r * q1e, st));
        for (int tl : {0, 1}) {
        const uint16_t* Wq = tl ? Wq1T : Wq1; const size_t qstride = tl ? q1e : (size_t)(Nq + F) * H;
        printf("qkv|fc1 M=%d tiled=%d:", Mx, tl);
        for (int v : vars) for (int gn : gns) {
            RC(showo_gemm_tune(gn, (v << 8) | 4, nullptr));
            CK(hipMemsetAsync(V0, 0, nvt * 2, st)); CK(hipMemsetAsync(V1, 0, nvt * 2, st));
            RC(showo_gemm_qkv_bf16(h, H, Wq1, H, bq1, lnp, lnp + 64, lnp + 128, lnp + 192, cosT, sinT, Q0, K0, V0, B, L, nH, 32, 1e-5f, 0, L, Lp, st));
            RC(showo_gemm_bf16(h, H, Wq1 + (size_t)Nq * H, H, bq1 + Nq, 0, f0, F, nullptr, 0, Mx, F, H, 1, st));
            RC(showo_gemm_qkv_fc1_bf16(h, H, Wq, H, bq1, lnp, lnp + 64, lnp + 128, lnp + 192, cosT, sinT, Q1, K1, V1, f1, F, F, B, L, nH, 32, 1e-5f, 0, L, Lp, tl, st));
            CK(hipStreamSynchronize(st));
            auto same = [&](const uint16_t* a, const uint16_t* b, size_t n) { std::vector<uint16_t> x(n), y(n); CK(hipMemcpy(x.data(), a, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), b, n * 2, hipMemcpyDeviceToHost)); return memcmp(x.data(), y.data(), n * 2) == 0; };
            const bool ok = same(Q0, Q1, nqk) && same(K0, K1, nqk) && same(V0, V1, nvt) && same(f0, f1, (size_t)Mx * F);
            const int iters = 10;
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) RC(showo_gemm_qkv_fc1_bf16(h, H, Wq + (size_t)(i % RW) * qstride, H, bq1, lnp, lnp + 64, lnp + 128, lnp + 192, cosT, sinT, Q1, K1, V1, f1, F, F, B, L, nH, 32, 1e-5f, 0, L, Lp, tl, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %d/gn%d:%.0fTF%s", v, gn, 2.0 * Mx * (Nq + F) * H * iters / (ms * 1e-3) / 1e12, ok ? "" : "**MISMATCH**");
            fflush(stdout);
            if (getenv("GEMM_BENCH_EPI_SPLIT") && tl == 0) {
                // round 3: what do the two halves of the fused epilogue cost?  Same variant, same rotating weights, microseconds per launch:
                //   q = QKV projection with the LayerNorm / RoPE / relayout epilogue, qp = the same GEMM with the plain bf16 epilogue,
                //   f = fc1 with bias + GELU, fp = fc1 plain bf16
                // best of 3 x 40 launches (10 were too few: +-10 us between repeats of the same build)
                auto t_us = [&](auto&& fn) { float best = 1e30f; for (int i = 0; i < 4; ++i) fn(i);
                                             for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(e0, st)); for (int i = 0; i < 40; ++i) fn(i);
                                                 CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float m_; CK(hipEventElapsedTime(&m_, e0, e1)); best = std::min(best, m_ * 1000.f / 40); }
                                             return best; };
                const float tq = t_us([&](int i) { RC(showo_gemm_qkv_bf16(h, H, Wq1 + (size_t)(i % RW) * qstride, H, bq1, lnp, lnp + 64, lnp + 128, lnp + 192, cosT, sinT, Q1, K1, V1, B, L, nH, 32, 1e-5f, 0, L, Lp, st)); });
                const float tqp = t_us([&](int i) { RC(showo_gemm_bf16(h, H, Wq1 + (size_t)(i % RW) * qstride, H, bq1, 0, f1, Nq, nullptr, 0, Mx, Nq, H, 0, st)); });
                const float tf = t_us([&](int i) { RC(showo_gemm_bf16(h, H, Wq1 + (size_t)(i % RW) * qstride + (size_t)Nq * H, H, bq1 + Nq, 0, f1, F, nullptr, 0, Mx, F, H, 1, st)); });
                const float tfp = t_us([&](int i) { RC(showo_gemm_bf16(h, H, Wq1 + (size_t)(i % RW) * qstride + (size_t)Nq * H, H, bq1 + Nq, 0, f1, F, nullptr, 0, Mx, F, H, 0, st)); });
                printf(" [us: qkv-epi %.1f plain %.1f | fc1-gelu %.1f plain %.1f]", tq, tqp, tf, tfp);
            }
        }
        printf("\n");
        }
        for (void* p : {(void*)attn, (void*)ffn, (void*)Wd, (void*)W2, (void*)Wcat, (void*)x0, (void*)x1, (void*)bd, (void*)b2, (void*)bsum, (void*)h, (void*)Wq1, (void*)bq1,
                        (void*)lnp, (void*)cosT, (void*)sinT, (void*)WcatT, (void*)Wq1T, (void*)Q0, (void*)K0, (void*)V0, (void*)Q1, (void*)K1, (void*)V1, (void*)f0, (void*)f1}) CK(hipFree(p));
    }
    RC(showo_gemm_tune(8, 2, nullptr));
    RC(showo_gemm_set_impl(0));
    return 0;
}
