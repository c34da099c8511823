// bf16 MFMA GEMM for gfx950: C[M,N] = A[M,K] * W[N,K]^T with fused epilogues, and the same core used
// as an implicit-GEMM 3x3 convolution (NHWC, K = 9*Cin).
//
// Replaces (reference; paths relative to the reference root): q/k/v/dense projections models/phi.py:657-659,727;
// PhiMLP fc1+gelu_new+fc2 models/phi.py:208-212; lm_head models/phi.py:1182-1183; VQGAN Conv2d 3x3 / 1x1
// models/common_modules.py:27-90,168-211,298-357.
//
// Structure (v1, "2-barrier LDS-staged" tier of the CDNA4 playbook):
//   block = 256 threads = 4 waves (2x2), block tile 128(n) x 128(m) x 64(k), wave tile 64x64 as 4x4
//   v_mfma_f32_16x16x32_bf16 fragments.  The MFMA A-operand is the WEIGHT tile and the B-operand the
//   ACTIVATION tile, so each lane ends up with 4 consecutive output columns (n) of one output row (m):
//   epilogue loads/stores are 8-16 B vectors and bias is a float4.
//   Both operands are K-contiguous; they are staged global -> VGPR -> LDS (double-buffered, register
//   prefetch of the next k-tile under the MFMAs) into a [128][64] bf16 image whose 16-B chunks are
//   XOR-swizzled with (row & 7) so the ds_read_b128 fragment reads are at most 2-way conflicted.
//   Block ids are remapped so that the blocks resident on one XCD (id % 8) walk the same weight panel.
#include "gemm_common.h"
#include "prof.h"
#include <atomic>
#include <cstdlib>

using namespace showo;

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_TILE = 128 * 64;  // bf16 elements per operand tile

struct ConvArgs {
    const bf16_t* X;   // NHWC bf16 input
    const bf16_t* Xlo; // split-precision mode: low halves (same layout)
    int B, Hin, Win, Cin, Hout, Wout, mode;
    // conv2p_split only: GroupNorm(32) statistics of the OUTPUT fused into the epilogue.  gn_part = per-tile (sum, sumsq) partials,
    // double [M / 256][32][2] (requires Hout * Wout % 256 == 0: a 256-pixel tile never straddles two images), gn_cpg = Cout / 32
    double* gn_part = nullptr;
    int gn_cpg = 0;
    // conv2p_split only: 3 = hi*hi + hi*lo + lo*hi (default, ~2^-17 operands); 2 drops the lo(weight) * hi(activation) product
    // (SHOWO_CONV_PRODUCTS=2: the measurement VERDICT r3 #3c asked for -- it does not stay inside the 2e-4 gates, see DESIGN)
    int products = 3;
};

// ---- activation-operand loaders: setup(i, m) once per staged row, load(i, k) per k-tile -> 16-B chunk
struct LinearLoader {
    const bf16_t* A; int lda; int M;
    const bf16_t* Alo;
    const bf16_t* rowp[4];
    int64_t lo_delta;  // element offset from the hi to the lo operand (same layout)
    __device__ inline void setup(int i, int m) {
        int mm = m < M ? m : M - 1;
        rowp[i] = A + (int64_t)mm * lda;
        lo_delta = Alo ? (Alo - A) : 0;
    }
    __device__ inline void tile(int) {}
    __device__ inline uint4 load(int i, int k) const { return *reinterpret_cast<const uint4*>(rowp[i] + k); }
    __device__ inline uint4 load_lo(int i, int k) const { return *reinterpret_cast<const uint4*>(rowp[i] + lo_delta + k); }
};

struct ConvLoader {
    ConvArgs c; int M;
    int oy[4], ox[4];
    const bf16_t* img[4];
    int ky, kx, cbase;  // per k-tile (block-uniform)
    __device__ inline void setup(int i, int m) {
        int mm = m < M ? m : M - 1;
        int hw = c.Hout * c.Wout;
        int b = mm / hw;
        int p = mm - b * hw;
        oy[i] = p / c.Wout;
        ox[i] = p - oy[i] * c.Wout;
        img[i] = c.X + (int64_t)b * c.Hin * c.Win * c.Cin;
    }
    // k0 = first k of the 64-wide tile; Cin % 64 == 0 guarantees the tile never straddles a tap
    __device__ inline void tile(int k0) {
        int tap = k0 / c.Cin;
        cbase = k0 - tap * c.Cin;
        ky = tap / 3;
        kx = tap - ky * 3;
    }
    template <bool LO>
    __device__ inline uint4 load_t(int i, int k) const {
        int iy, ix;
        bool ok;
        if (c.mode == 2) {  // pad (0,1,0,1) then stride 2 (common_modules.py:83-88)
            iy = 2 * oy[i] + ky; ix = 2 * ox[i] + kx;
            ok = iy < c.Hin && ix < c.Win;
        } else if (c.mode == 1) {  // nearest 2x upsample, then pad 1 (common_modules.py:36-40)
            int uy = oy[i] + ky - 1, ux = ox[i] + kx - 1;
            ok = uy >= 0 && ux >= 0 && uy < c.Hout && ux < c.Wout;
            iy = uy >> 1; ix = ux >> 1;
        } else {
            iy = oy[i] + ky - 1; ix = ox[i] + kx - 1;
            ok = iy >= 0 && ix >= 0 && iy < c.Hin && ix < c.Win;
        }
        if (!ok) return make_uint4(0, 0, 0, 0);
        // k & 63 = this thread's chunk offset inside the tile
        return *reinterpret_cast<const uint4*>(img[i] + (LO ? (c.Xlo - c.X) : 0) + ((int64_t)iy * c.Win + ix) * c.Cin + cbase + (k & 63));
    }
    __device__ inline uint4 load(int i, int k) const { return load_t<false>(i, k); }
    __device__ inline uint4 load_lo(int i, int k) const { return load_t<true>(i, k); }
};



// SPLIT: both operands come as (hi, lo) bf16 pairs with x = hi + lo to ~2^-17; the product is accumulated as
// hi*hi + hi*lo + lo*hi in the fp32 MFMA accumulators (3 MFMAs per tile pair, ~fp32-class accuracy at 3/16 of the
// fp32-MFMA cost).  Used by the VQGAN path, whose token ids must track the fp32 reference (SURVEY.md §7 hard parts).
template <int EPI, class Loader, bool SPLIT, bool F16 = false>  // F16: IEEE-half operands (common.h Op16; plain form only)
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g, Loader ld_in) {
    Loader ld = ld_in;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // plain: [2][128*64] per operand (double-buffered, 64 KiB).  SPLIT: four operand images, SINGLE-buffered (64 KiB, two
    // barriers per k-tile) so that two blocks share a CU and overlap each other's staging and MFMA phases.
    constexpr int NBUF = SPLIT ? 1 : 2;
    bf16_t* sW = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* sA = sW + NBUF * LDS_TILE;
    bf16_t* sWl = sA + NBUF * LDS_TILE;                          // SPLIT only
    bf16_t* sAl = sWl + NBUF * LDS_TILE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    // XCD-aware bijective remap: blocks with equal (id % 8) share an L2; give each XCD a contiguous id range
    int nwg = tilesM * tilesN, bid = blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tn = bid / tilesM, tm = bid - tn * tilesM;
    const int m0 = tm * BM, n0 = tn * BN;

    // staging assignment: thread -> chunk c (16 B) of rows r0 + 32*i
    const int sc = tid & 7, sr0 = tid >> 3;
    uint4 ra[4], rw[4], ral[SPLIT ? 4 : 1], rwl[SPLIT ? 4 : 1];
    const int nk = g.K / BK;
    const int64_t wlo_delta = SPLIT ? (g.Wlo - g.W) : 0;

    const bf16_t* wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = sr0 + 32 * i;
        int n = n0 + row;
        n = n < g.N ? n : g.N - 1;
        wrow[i] = g.W + (int64_t)n * g.ldw;
        ld.setup(i, m0 + row);
    }
    auto gload = [&](int kt) {
        int k = kt * BK + sc * 8;
        ld.tile(kt * BK);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rw[i] = *reinterpret_cast<const uint4*>(wrow[i] + k);
            ra[i] = ld.load(i, k);
            if (SPLIT) {
                rwl[i] = *reinterpret_cast<const uint4*>(wrow[i] + wlo_delta + k);
                ral[i] = ld.load_lo(i, k);
            }
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = sr0 + 32 * i;
            int off = row * 64 + ((sc ^ (row & 7)) << 3);
            *reinterpret_cast<uint4*>(sW + buf * LDS_TILE + off) = rw[i];
            *reinterpret_cast<uint4*>(sA + buf * LDS_TILE + off) = ra[i];
            if (SPLIT) {
                *reinterpret_cast<uint4*>(sWl + buf * LDS_TILE + off) = rwl[i];
                *reinterpret_cast<uint4*>(sAl + buf * LDS_TILE + off) = ral[i];
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wn = wave >> 1, wm = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;

    gload(0);
    swrite(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = SPLIT ? 0 : (kt & 1);
        if (kt + 1 < nk) gload(kt + 1);
        const bf16_t* bW = sW + cur * LDS_TILE;
        const bf16_t* bA = sA + cur * LDS_TILE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 wf[4], af[4];
            const int chunk = kk * 4 + fg;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = wn * 64 + i * 16 + fr;
                wf[i] = *reinterpret_cast<const bf16x8*>(bW + row * 64 + ((chunk ^ (row & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int row = wm * 64 + j * 16 + fr;
                af[j] = *reinterpret_cast<const bf16x8*>(bA + row * 64 + ((chunk ^ (row & 7)) << 3));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = Op16<F16>::mfma16(wf[i], af[j], acc[i][j]);
            if (SPLIT) {
                const bf16_t* bWl = sWl + cur * LDS_TILE;
                const bf16_t* bAl = sAl + cur * LDS_TILE;
                bf16x8 lf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // hi(W) * lo(A)
                    int row = wm * 64 + j * 16 + fr;
                    lf[j] = *reinterpret_cast<const bf16x8*>(bAl + row * 64 + ((chunk ^ (row & 7)) << 3));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], lf[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // lo(W) * hi(A)
                    int row = wn * 64 + i * 16 + fr;
                    lf[i] = *reinterpret_cast<const bf16x8*>(bWl + row * 64 + ((chunk ^ (row & 7)) << 3));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lf[i], af[j], acc[i][j], 0, 0, 0);
            }
        }
        if (SPLIT) __syncthreads();  // single buffer: every wave is done reading before the next tile is written
        if (kt + 1 < nk) swrite(SPLIT ? 0 : (cur ^ 1));
        __syncthreads();
    }

    // ---- epilogue: lane holds out[m][n .. n+3] for each (i,j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + i * 16 + fg * 4;
        float bn[4];
        load_bias4(g, n, bn);
#pragma unroll
        for (int j = 0; j < 4; ++j) store_frag<EPI, F16>(g, acc[i][j], m0 + wm * 64 + j * 16 + fr, n, bn);
    }
}

constexpr int SMEM_BYTES = 4 * LDS_TILE * 2;  // 64 KiB

template <int EPI, class Loader, bool SPLIT = false, bool F16 = false>
int launch(const GemmArgs& g, const Loader& ld, hipStream_t s) {
    static bool attr_set = false;
    auto kfn = gemm_kernel<EPI, Loader, SPLIT, F16>;
    const int smem = SMEM_BYTES;  // 64 KiB in both modes (SPLIT: 4 single-buffered images)
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return set_error_hip(e, "hipFuncSetAttribute(gemm)", __FILE__, __LINE__);
        attr_set = true;
    }
    int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    kfn<<<dim3(tilesM * tilesN), dim3(256), smem, s>>>(g, ld);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error_hip(e, "gemm launch", __FILE__, __LINE__);
    return 0;
}

// IEEE-half operands (precision 2) on the 128^2 kernel: linear loader only (the convolutions have their own split-precision path)
int dispatch_f16(const GemmArgs& g, const LinearLoader& ld, int epilogue, hipStream_t s) {
    switch (epilogue) {
        case SHOWO_EPI_BF16: return launch<SHOWO_EPI_BF16, LinearLoader, false, true>(g, ld, s);
        case SHOWO_EPI_GELU_BF16: return launch<SHOWO_EPI_GELU_BF16, LinearLoader, false, true>(g, ld, s);
        case SHOWO_EPI_F32: return launch<SHOWO_EPI_F32, LinearLoader, false, true>(g, ld, s);
        case SHOWO_EPI_RESID_F32: return launch<SHOWO_EPI_RESID_F32, LinearLoader, false, true>(g, ld, s);
    }
    return set_error_msg(1, "gemm: unknown epilogue");
}

template <class Loader>
int dispatch(const GemmArgs& g, const Loader& ld, int epilogue, hipStream_t s) {
    switch (epilogue) {
        case SHOWO_EPI_BF16: return launch<SHOWO_EPI_BF16>(g, ld, s);
        case SHOWO_EPI_GELU_BF16: return launch<SHOWO_EPI_GELU_BF16>(g, ld, s);
        case SHOWO_EPI_F32: return launch<SHOWO_EPI_F32>(g, ld, s);
        case SHOWO_EPI_RESID_F32: return launch<SHOWO_EPI_RESID_F32>(g, ld, s);
    }
    return set_error_msg(1, "gemm: unknown epilogue");
}

template <class Loader>
int dispatch_split(const GemmArgs& g, const Loader& ld, int epilogue, hipStream_t s) {
    switch (epilogue) {
        case SHOWO_EPI_F32: return launch<SHOWO_EPI_F32, Loader, true>(g, ld, s);
        case SHOWO_EPI_RESID_F32: return launch<SHOWO_EPI_RESID_F32, Loader, true>(g, ld, s);
    }
    return set_error_msg(1, "gemm x3: only fp32 epilogues (2, 3) are supported");
}


// =====================================================================================================
// v2: 256 x 256 x 64 tile, 8 waves (4 along n x 2 along m, wave tile 64 x 128), operands streamed
// HBM -> LDS with global_load_lds (16 B/lane, no VGPR round trip) into two 64 KiB buffers; the LDS image is
// lane-linear per wave-instruction (8 rows x 128 B), so the (row & 7) XOR swizzle is applied to the per-lane
// SOURCE address and again on the ds_read (CDNA4 playbook: "swizzle both sides or neither").
// One barrier per k-tile; the next tile's DMA is in flight under the 64 MFMAs of the current one.
// =====================================================================================================
constexpr int LDS_TILE2 = 256 * 64;
constexpr int SMEM2_BYTES = 4 * LDS_TILE2 * 2;  // 128 KiB

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[64];  // source of out-of-image conv taps

struct LinearPtr {
    const bf16_t* A; int lda; int M;
    const bf16_t* rowp[4];
    __device__ inline void setup(int i, int m) {
        int mm = m < M ? m : M - 1;
        rowp[i] = A + (int64_t)mm * lda;
    }
    __device__ inline void tile(int) {}
    __device__ inline const bf16_t* ptr(int i, int k0, int coff) const { return rowp[i] + k0 + coff; }
};

struct ConvPtr {
    ConvArgs c; int M;
    int oy[4], ox[4];
    const bf16_t* img[4];
    int ky, kx, cbase;
    __device__ inline void setup(int i, int m) {
        int mm = m < M ? m : M - 1;
        int hw = c.Hout * c.Wout;
        int b = mm / hw;
        int p = mm - b * hw;
        oy[i] = p / c.Wout;
        ox[i] = p - oy[i] * c.Wout;
        img[i] = c.X + (int64_t)b * c.Hin * c.Win * c.Cin;
    }
    __device__ inline void tile(int k0) {
        int tap = k0 / c.Cin;
        cbase = k0 - tap * c.Cin;
        ky = tap / 3;
        kx = tap - ky * 3;
    }
    __device__ inline const bf16_t* ptr(int i, int, int coff) const {
        int iy, ix;
        bool ok;
        if (c.mode == 2) {
            iy = 2 * oy[i] + ky; ix = 2 * ox[i] + kx;
            ok = iy < c.Hin && ix < c.Win;
        } else if (c.mode == 1) {
            int uy = oy[i] + ky - 1, ux = ox[i] + kx - 1;
            ok = uy >= 0 && ux >= 0 && uy < c.Hout && ux < c.Wout;
            iy = uy >> 1; ix = ux >> 1;
        } else {
            iy = oy[i] + ky - 1; ix = ox[i] + kx - 1;
            ok = iy >= 0 && ix >= 0 && iy < c.Hin && ix < c.Win;
        }
        if (!ok) return reinterpret_cast<const bf16_t*>(g_zero_page);
        return img[i] + ((int64_t)iy * c.Win + ix) * c.Cin + cbase + coff;
    }
};


template <int EPI, class Loader>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs g, Loader ld_in) {
    Loader ld = ld_in;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* sW = reinterpret_cast<bf16_t*>(smem_raw);  // [2][256*64]
    bf16_t* sA = sW + 2 * LDS_TILE2;                     // [2][256*64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesM = (g.M + B2 - 1) / B2, tilesN = (g.N + B2 - 1) / B2;
    int nwg = tilesM * tilesN, bid = blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tn = bid / tilesM, tm = bid - tn * tilesM;
    const int m0 = tm * B2, n0 = tn * B2;
    const int nk = g.K / BK;

    // staging: wave w issues row-groups w, w+8, w+16, w+24 (8 rows x 128 B each) of both operands
    const int srow = lane >> 3;                       // row inside the group
    const int coff = ((lane & 7) ^ srow) << 3;        // swizzled source chunk (elements)
    const bf16_t* wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = (wave + 8 * i) * 8 + srow;
        int n = n0 + row;
        n = n < g.N ? n : g.N - 1;
        wrow[i] = g.W + (int64_t)n * g.ldw;
        ld.setup(i, m0 + row);
    }
    auto stage = [&](int buf, int kt) {
        const int k0 = kt * BK;
        ld.tile(k0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int goff = (wave + 8 * i) * 512;  // 8 rows * 64 elements
            glds16(wrow[i] + k0 + coff, sW + buf * LDS_TILE2 + goff);
            glds16(ld.ptr(i, k0, coff), sA + buf * LDS_TILE2 + goff);
        }
    };

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wn = wave & 3, wm = wave >> 2;
    const int fr = lane & 15, fg = lane >> 4;

    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const bf16_t* bW = sW + cur * LDS_TILE2;
        const bf16_t* bA = sA + cur * LDS_TILE2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 wf[4], af[8];
            const int chunk = kk * 4 + fg;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = wn * 64 + i * 16 + fr;
                wf[i] = *reinterpret_cast<const bf16x8*>(bW + row * 64 + ((chunk ^ (row & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int row = wm * 128 + j * 16 + fr;
                af[j] = *reinterpret_cast<const bf16x8*>(bA + row * 64 + ((chunk ^ (row & 7)) << 3));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();  // drains the in-flight DMA (vmcnt(0)) and fences the buffer swap
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + i * 16 + fg * 4;
        float bn[4];
        load_bias4(g, n, bn);
#pragma unroll
        for (int j = 0; j < 8; ++j) store_frag<EPI>(g, acc[i][j], m0 + wm * 128 + j * 16 + fr, n, bn);
    }
}

template <int EPI, class Loader>
int launch2(const GemmArgs& g, const Loader& ld, hipStream_t s) {
    static bool attr_set = false;
    auto kfn = gemm256_kernel<EPI, Loader>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES);
        if (e != hipSuccess) return set_error_hip(e, "hipFuncSetAttribute(gemm256)", __FILE__, __LINE__);
        attr_set = true;
    }
    int tilesM = (g.M + B2 - 1) / B2, tilesN = (g.N + B2 - 1) / B2;
    kfn<<<dim3(tilesM * tilesN), dim3(512), SMEM2_BYTES, s>>>(g, ld);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error_hip(e, "gemm256 launch", __FILE__, __LINE__);
    return 0;
}

template <class Loader>
int dispatch2(const GemmArgs& g, const Loader& ld, int epilogue, hipStream_t s) {
    switch (epilogue) {
        case SHOWO_EPI_BF16: return launch2<SHOWO_EPI_BF16>(g, ld, s);
        case SHOWO_EPI_GELU_BF16: return launch2<SHOWO_EPI_GELU_BF16>(g, ld, s);
        case SHOWO_EPI_F32: return launch2<SHOWO_EPI_F32>(g, ld, s);
        case SHOWO_EPI_RESID_F32: return launch2<SHOWO_EPI_RESID_F32>(g, ld, s);
    }
    return set_error_msg(1, "gemm: unknown epilogue");
}

int g_gemm_flags = 0;  // experiment flags of showo_gemm_tune (bits 0..7)
unsigned long long* g_gemm_dbg = nullptr;

// =====================================================================================================
// GEMV form for decode steps (M <= 8 token rows): the weight matrix is streamed ONCE straight into VGPRs (no LDS round
// trip: nothing is shared between waves), a wave owns GV_COLS output columns and splits K over its lanes (16-B loads,
// 4 loads in flight per column), the activation rows come from L1/L2; fp32 accumulate, wave-level reduction, the same
// epilogues as the tile kernels.  HBM-bound: N*K*2 bytes per launch (2.9 GB per decoded token over the whole model).
// =====================================================================================================
constexpr int GV_COLS = 1;  // output columns per wave (1: most waves in flight -- the kernel is latency x concurrency bound)
template <int EPI, int MR, bool F16 = false>
__global__ __launch_bounds__(256) void gemv_kernel(GemmArgs g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * GV_COLS;
    if (n0 >= g.N) return;
    float acc[MR][GV_COLS];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int c = 0; c < GV_COLS; ++c) acc[m][c] = 0.f;
    const bf16_t* wr[GV_COLS];
#pragma unroll
    for (int c = 0; c < GV_COLS; ++c) wr[c] = g.W + (int64_t)min(n0 + c, g.N - 1) * g.ldw;
    for (int k0 = lane * 8; k0 < g.K; k0 += 4 * 512) {
        uint4 wv[4][GV_COLS];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < GV_COLS; ++c) {
                const int k = k0 + u * 512;
                wv[u][c] = k < g.K ? ldg_nt16(wr[c] + k) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * 512;
            if (k >= g.K) break;
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                if (m >= g.M) break;
                const uint4 av = *reinterpret_cast<const uint4*>(g.A + (int64_t)m * g.lda + k);
#pragma unroll
                for (int c = 0; c < GV_COLS; ++c) acc[m][c] = dot8_op<F16>(wv[u][c], av, acc[m][c]);  // common.h: the decode GEMVs' order
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int c = 0; c < GV_COLS; ++c) acc[m][c] = wave_sum(acc[m][c]);
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m >= g.M) break;
#pragma unroll
            for (int c = 0; c < GV_COLS; ++c) {
                const int n = n0 + c;
                if (n >= g.N) break;
                float v = acc[m][c];
                if (g.bias) v += g.bias_per_row ? g.bias[m] : g.bias[n];
                if (EPI == SHOWO_EPI_GELU_BF16) v = gelu_new_fast(v);
                if (EPI == SHOWO_EPI_BF16 || EPI == SHOWO_EPI_GELU_BF16) {
                    reinterpret_cast<bf16_t*>(g.out)[(int64_t)m * g.ldo + n] = Op16<F16>::cvt(v);
                } else {
                    if (EPI == SHOWO_EPI_RESID_F32) v += g.resid[(int64_t)m * g.ldr + n];
                    reinterpret_cast<float*>(g.out)[(int64_t)m * g.ldo + n] = v;
                }
            }
        }
    }
}

template <int EPI, bool F16 = false>
int launch_gemv(const GemmArgs& g, hipStream_t s) {
    const int blocks = (g.N + 4 * GV_COLS - 1) / (4 * GV_COLS);
    if (g.M <= 1) gemv_kernel<EPI, 1, F16><<<dim3(blocks), dim3(256), 0, s>>>(g);
    else if (g.M <= 4) gemv_kernel<EPI, 4, F16><<<dim3(blocks), dim3(256), 0, s>>>(g);
    else gemv_kernel<EPI, 8, F16><<<dim3(blocks), dim3(256), 0, s>>>(g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error_hip(e, "gemv launch", __FILE__, __LINE__);
    return 0;
}
int dispatch_gemv(const GemmArgs& g, int epilogue, hipStream_t s) {
    if (g.op) {
        switch (epilogue) {
            case SHOWO_EPI_BF16: return launch_gemv<SHOWO_EPI_BF16, true>(g, s);
            case SHOWO_EPI_GELU_BF16: return launch_gemv<SHOWO_EPI_GELU_BF16, true>(g, s);
            case SHOWO_EPI_F32: return launch_gemv<SHOWO_EPI_F32, true>(g, s);
            case SHOWO_EPI_RESID_F32: return launch_gemv<SHOWO_EPI_RESID_F32, true>(g, s);
        }
        return set_error_msg(1, "gemm: unknown epilogue");
    }
    switch (epilogue) {
        case SHOWO_EPI_BF16: return launch_gemv<SHOWO_EPI_BF16>(g, s);
        case SHOWO_EPI_GELU_BF16: return launch_gemv<SHOWO_EPI_GELU_BF16>(g, s);
        case SHOWO_EPI_F32: return launch_gemv<SHOWO_EPI_F32>(g, s);
        case SHOWO_EPI_RESID_F32: return launch_gemv<SHOWO_EPI_RESID_F32>(g, s);
    }
    return set_error_msg(1, "gemm: unknown epilogue");
}

// 1 = 128^2 register-staged kernel, 2 = 256^2 global_load_lds kernel, 0 = pick by shape
int g_gemm_forced = -1;
int gemm_impl_choice(int M, int N) {
    if (g_gemm_forced < 0) {
        const char* e = getenv("SHOWO_GEMM_IMPL");
        g_gemm_forced = e ? atoi(e) : 0;
    }
    if (g_gemm_forced == 1 || g_gemm_forced == 2 || g_gemm_forced == 5 || g_gemm_forced == 6) return g_gemm_forced;
    if (M <= 8) return 6;  // decode step: weight-streaming GEMV
    // phase-split 256-wide kernel from 256 rows up: measured on the prefill (tools/prefill_sweep.py) 631 rows 11.3 -> 9.3 ms,
    // 387 rows 10.3 -> 9.6 ms against the 128x128 kernel
    return (M >= 256 && N >= 256) ? 5 : 1;
}

}  // namespace

// 0 = choose by shape (default), 1 = 128^2 register-staged kernel, 2 = 256^2 global_load_lds kernel
extern "C" int showo_gemm_set_impl(int impl) {
    g_gemm_forced = (impl == 1 || impl == 2 || impl == 5 || impl == 6) ? impl : 0;  // (3 / 4 = the 4-phase rung of round 1, removed from the library in round 5)
    return 0;
}

// experiment knobs of the v3 kernel (tools/gemm_bench.cpp): gn = n-panels per tile group, flags bit0 = no stagger,
// dbg = device buffer of 512 uint64 receiving block 0's per-barrier timestamps of k-tiles 8 and 9 (bf16 epilogue only)
extern "C" int showo_gemm_tune(int gn, int flags, unsigned long long* dbg) {
    g_gemm_gn = gn; g_gemm_flags = flags & 0xff; g_gemm_dbg = dbg;
    g_gemm_bm = (flags >> 8) ? (flags >> 8) : -1;  // impl 5: tile variant code (0 = automatic)
    if (flags & 2) g_gemm_pf = 1;       // impl 5: L2 prefetch of the weight panel on ...
    else if (flags & 4) g_gemm_pf = 0;  // ... off; neither bit: unchanged (SHOWO_GEMM_PF, default off)
    if (flags & 8) g_gemm_stage = 0;    // impl 5: bf16 epilogue stores direct (8) or staged through LDS (16); neither: unchanged
    else if (flags & 16) g_gemm_stage = 1;
    if (flags & 64) g_gemm_splitk = 0;  // impl 5: split-K of launches with few tiles off (64) / on (128); neither: unchanged
    else if (flags & 128) g_gemm_splitk = 1;
    return 0;
}

static int gemm_op16_impl(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, int bias_per_row,
                          void* out, int ldo, const float* resid, int ldr, int M, int N, int K, int epilogue, int op,
                          void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (op != SHOWO_OP_BF16 && op != SHOWO_OP_F16) return set_error_msg(1, "gemm: op must be SHOWO_OP_BF16 or SHOWO_OP_F16");
    if (K <= 0 || (K % BK) != 0) return set_error_msg(1, "gemm: K must be a positive multiple of 64");
    if ((lda % 8) || (ldw % 8) || (((uintptr_t)A) & 15) || (((uintptr_t)W) & 15))
        return set_error_msg(1, "gemm: A/W must be 16B aligned with lda,ldw multiples of 8");
    if (epilogue == SHOWO_EPI_RESID_F32 && !resid) return set_error_msg(1, "gemm: resid required");
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.Wlo = nullptr; g.bias = bias; g.bias_per_row = bias_per_row;
    g.out = out; g.ldo = ldo; g.resid = resid; g.ldr = ldr; g.M = M; g.N = N; g.K = K;
    g.gn = 1; g.flags = 0; g.dbg = nullptr;
    g.op = op;
    bool f32 = (epilogue == SHOWO_EPI_F32 || epilogue == SHOWO_EPI_RESID_F32);
    uintptr_t align = f32 ? 15 : 7;
    g.vec_out = ((ldo % 4) == 0) && ((((uintptr_t)out) & align) == 0);
    if (epilogue == SHOWO_EPI_RESID_F32) g.vec_out = g.vec_out && ((ldr % 4) == 0) && ((((uintptr_t)resid) & 15) == 0);
    LinearLoader ld;
    ld.A = A; ld.lda = lda; ld.M = M; ld.Alo = nullptr;
    ProfScope prof(PROF_GEMM, 2.0 * M * N * K, (hipStream_t)stream);
    int impl = gemm_impl_choice(M, N);
    // the phase-split kernels address both operands as base + 32-bit byte offset
    if (impl >= 3 && ((int64_t)M * lda * 2 >= ((int64_t)1 << 32) || (int64_t)N * ldw * 2 >= ((int64_t)1 << 32))) impl = 2;
    if (impl == 6 && M <= 8 && (K % 8) == 0) return dispatch_gemv(g, epilogue, (hipStream_t)stream);
    if (impl == 6) impl = 1;
    if (impl == 5) return gemm2p_dispatch(g, epilogue, (hipStream_t)stream);
    if (impl == 2 && !op) {  // (the 256^2 A/B rung has bf16 instances only)
        LinearPtr lp;
        lp.A = A; lp.lda = lda; lp.M = M;
        return dispatch2(g, lp, epilogue, (hipStream_t)stream);
    }
    if (op) return dispatch_f16(g, ld, epilogue, (hipStream_t)stream);
    return dispatch(g, ld, epilogue, (hipStream_t)stream);
}

extern "C" int showo_gemm_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, int bias_per_row,
                               void* out, int ldo, const float* resid, int ldr, int M, int N, int K, int epilogue,
                               void* stream) {
    return gemm_op16_impl(A, lda, W, ldw, bias, bias_per_row, out, ldo, resid, ldr, M, N, K, epilogue, SHOWO_OP_BF16, stream);
}
// the same GEMM on either 16-bit operand type (op = SHOWO_OP_BF16 | SHOWO_OP_F16: A, W and a 16-bit output are of that type)
extern "C" int showo_gemm_op16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, int bias_per_row,
                               void* out, int ldo, const float* resid, int ldr, int M, int N, int K, int epilogue, int op,
                               void* stream) {
    return gemm_op16_impl(A, lda, W, ldw, bias, bias_per_row, out, ldo, resid, ldr, M, N, K, epilogue, op, stream);
}

// ---- tiled weight layout of the production kernel: [ceil(N/256)][K/64][256][64] bf16; inside a 32 KiB block row r holds its
// eight 16-B chunks at positions p = chunk ^ (r & 7) -- the block IS the LDS image of one k-tile of one weight panel, so the DMA
// source is lane-linear (1 KiB contiguous per wave-instruction) and a k-loop walks one contiguous stream per panel.
namespace {
__global__ __launch_bounds__(256) void tile_weight_kernel(const bf16_t* __restrict__ W, int ldw, int N, int K, bf16_t* __restrict__ out) {
    const int64_t chunks = (int64_t)((N + 255) / 256) * (K / 64) * 256 * 8;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(c & 7), r = (int)((c >> 3) & 255);
        const int64_t blk = c >> 11;
        const int kt = (int)(blk % (K / 64)), np = (int)(blk / (K / 64));
        const int n = np * 256 + r;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (n < N) v = *reinterpret_cast<const uint4*>(W + (int64_t)n * ldw + kt * 64 + ((p ^ (r & 7)) << 3));
        *reinterpret_cast<uint4*>(out + c * 8) = v;
    }
}
}  // namespace

extern "C" int64_t showo_gemm_tiled_elems(int N, int K) { return (int64_t)((N + 255) / 256) * 256 * K; }

extern "C" int showo_gemm_tile_weight(const uint16_t* W, int ldw, int N, int K, uint16_t* out, void* stream) {
    if (N <= 0 || K <= 0) return 0;
    if ((K % 64) || (ldw % 8) || (((uintptr_t)W) & 15) || (((uintptr_t)out) & 15)) return set_error_msg(1, "gemm_tile_weight: K % 64, ldw % 8, 16B alignment required");
    const int64_t chunks = showo_gemm_tiled_elems(N, K) / 8;
    int64_t blocks = (chunks + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    tile_weight_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(W, ldw, N, K, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error_hip(e, "tile_weight launch", __FILE__, __LINE__);
    return 0;
}

// fused QKV projection: Q/K/V^T = relayout(rope(layernorm(A Wqkv^T + b)))  (reference models/phi.py:657-694)
// With ffn_out != NULL the weight is [Wqkv ; W1] ([3 nH 64 + F, K] rows, bias likewise) and the launch also produces
// ffn_out[m][0..F) = gelu_new(A W1^T + b1) (models/phi.py:208-212): q/k/v and fc1 read the same LayerNorm output.
static int gemm_qkv_impl(const uint16_t* A, int lda, const uint16_t* Wqkv, int ldw, const float* bias, const float* qln_w,
                         const float* qln_b, const float* kln_w, const float* kln_b, const float* cos_tab, const float* sin_tab,
                         uint16_t* Q, uint16_t* K, uint16_t* Vt, int B, int L, int nH, int rot, float eps, int pos0, int Lcap, int Lp,
                         uint16_t* ffn_out, int ldf, int F, int w_tiled, void* stream, uint16_t* raw_qkv = nullptr, int ldraw = 0,
                         uint16_t* ffn_pre = nullptr, int Kcat = 0, uint16_t* Qlo = nullptr, uint16_t* Klo = nullptr, uint16_t* Vtlo = nullptr,
                         uint16_t* ffn_lo = nullptr, int op = 0) {
    const int M = B * L, Nq = 3 * nH * 64, Kd = Kcat > 0 ? Kcat : nH * 64;
    const bool split = Qlo != nullptr;
    const int N = Nq + (ffn_out ? F : 0);
    if (M <= 0) return 0;
    if (rot != 32) return set_error_msg(1, "gemm_qkv: the fused epilogue implements rotary_dim 32 (use showo_gemm_bf16 + showo_qk_prep)");
    if (Q ? (!K || !Vt) : !raw_qkv) return set_error_msg(1, "gemm_qkv: Q, K, Vt required (only the save form may omit all three)");
    if ((Lp % 64) || Lp < pos0 + L || Lcap < pos0 + L) return set_error_msg(1, "gemm_qkv: bad Lp/Lcap");
    if ((lda % 8) || (ldw % 8) || (((uintptr_t)A) & 15) || (((uintptr_t)Wqkv) & 15))
        return set_error_msg(1, "gemm_qkv: A/W must be 16B aligned with lda,ldw multiples of 8");
    if ((int64_t)M * lda * 2 >= ((int64_t)1 << 32) || (int64_t)N * ldw * 2 >= ((int64_t)1 << 32))
        return set_error_msg(1, "gemm_qkv: operand larger than 4 GiB");
    if (ffn_out && (F <= 0 || (F % 4) || (ldf % 4) || (((uintptr_t)ffn_out) & 7) || (Nq % B2)))
        return set_error_msg(1, "gemm_qkv_fc1: F, ldf must be multiples of 4, ffn_out 8B aligned, 3*nH*64 a multiple of 256");
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = Wqkv; g.ldw = ldw; g.Wlo = nullptr; g.bias = bias; g.bias_per_row = 0;
    g.out = nullptr; g.ldo = 0; g.resid = nullptr; g.ldr = 0; g.M = M; g.N = N; g.K = Kd; g.vec_out = 1;
    g.gn = 1; g.flags = 0; g.dbg = nullptr;
    g.qw = qln_w; g.qb = qln_b; g.kw = kln_w; g.kb = kln_b; g.cosT = cos_tab; g.sinT = sin_tab;
    g.Q = Q; g.Kd = K; g.Vt = Vt; g.L = L; g.nH = nH; g.pos0 = pos0; g.Lcap = Lcap; g.Lp = Lp; g.eps = eps;
    if (ffn_out) { g.Nq = Nq; g.out2 = ffn_out; g.ldo2 = ldf; }
    if (raw_qkv) {
        if ((ldraw % 4) || (((uintptr_t)raw_qkv) & 7)) return set_error_msg(1, "gemm_qkv_fc1_save: raw_qkv 8B aligned, ld a multiple of 4");
        g.raw = raw_qkv; g.ldraw = ldraw;
    }
    if (ffn_pre) {
        if (!ffn_out || (((uintptr_t)ffn_pre) & 7)) return set_error_msg(1, "gemm_qkv_fc1_save: ffn_pre needs ffn_out, 8B aligned");
        g.pre = ffn_pre;
    }
    g.wtiled = w_tiled ? 1 : 0;
    if (op != SHOWO_OP_BF16 && op != SHOWO_OP_F16) return set_error_msg(1, "gemm_qkv: op must be SHOWO_OP_BF16 or SHOWO_OP_F16");
    if (op && (split || raw_qkv || ffn_pre)) return set_error_msg(1, "gemm_qkv: the (hi, lo) and save-for-backward forms have bf16 operands only");
    g.op = op;
    if (split) {
        if (!Klo || !Vtlo || !ffn_lo || !ffn_out || raw_qkv || ffn_pre || (((uintptr_t)ffn_lo) & 7))
            return set_error_msg(1, "gemm_qkv_fc1_split: Qlo, Klo, Vtlo, ffn_lo (8B aligned) and ffn_out required; no save-for-backward outputs");
        g.Qlo = Qlo; g.Klo = Klo; g.Vtlo = Vtlo; g.out2lo = ffn_lo;
    }
    ProfScope prof(PROF_GEMM, 2.0 * M * N * Kd, (hipStream_t)stream);
    return gemm2p_dispatch(g, split ? EPI_QKV_SPLIT : EPI_QKV, (hipStream_t)stream);
}

extern "C" int showo_gemm_qkv_bf16(const uint16_t* A, int lda, const uint16_t* Wqkv, int ldw, const float* bias,
                                   const float* qln_w, const float* qln_b, const float* kln_w, const float* kln_b,
                                   const float* cos_tab, const float* sin_tab, uint16_t* Q, uint16_t* K, uint16_t* Vt, int B,
                                   int L, int nH, int rot, float eps, int pos0, int Lcap, int Lp, void* stream) {
    return gemm_qkv_impl(A, lda, Wqkv, ldw, bias, qln_w, qln_b, kln_w, kln_b, cos_tab, sin_tab, Q, K, Vt, B, L, nH, rot, eps, pos0,
                         Lcap, Lp, nullptr, 0, 0, 0, stream);
}

extern "C" int showo_gemm_qkv_fc1_bf16(const uint16_t* A, int lda, const uint16_t* Wqkv_fc1, int ldw, const float* bias,
                                       const float* qln_w, const float* qln_b, const float* kln_w, const float* kln_b,
                                       const float* cos_tab, const float* sin_tab, uint16_t* Q, uint16_t* K, uint16_t* Vt,
                                       uint16_t* ffn_out, int ldf, int F, int B, int L, int nH, int rot, float eps, int pos0,
                                       int Lcap, int Lp, int w_tiled, void* stream) {
    if (!ffn_out) return set_error_msg(1, "gemm_qkv_fc1: ffn_out required");
    return gemm_qkv_impl(A, lda, Wqkv_fc1, ldw, bias, qln_w, qln_b, kln_w, kln_b, cos_tab, sin_tab, Q, K, Vt, B, L, nH, rot, eps,
                         pos0, Lcap, Lp, ffn_out, ldf, F, w_tiled, stream);
}

// showo_gemm_qkv_bf16 / showo_gemm_qkv_fc1_bf16 on either 16-bit operand type (ffn_out == NULL: the projection alone)
extern "C" int showo_gemm_qkv_fc1_op16(const uint16_t* A, int lda, const uint16_t* Wqkv_fc1, int ldw, const float* bias,
                                       const float* qln_w, const float* qln_b, const float* kln_w, const float* kln_b,
                                       const float* cos_tab, const float* sin_tab, uint16_t* Q, uint16_t* K, uint16_t* Vt,
                                       uint16_t* ffn_out, int ldf, int F, int B, int L, int nH, int rot, float eps, int pos0,
                                       int Lcap, int Lp, int w_tiled, int op, void* stream) {
    return gemm_qkv_impl(A, lda, Wqkv_fc1, ldw, bias, qln_w, qln_b, kln_w, kln_b, cos_tab, sin_tab, Q, K, Vt, B, L, nH, rot, eps,
                         pos0, Lcap, Lp, ffn_out, ffn_out ? ldf : 0, ffn_out ? F : 0, ffn_out ? w_tiled : 0, stream, nullptr, 0, nullptr, 0,
                         nullptr, nullptr, nullptr, nullptr, op);
}

// Accuracy-mode form of the same launch (showo_engine_set_precision 1 on the production kernel).  The operands are K-CONCATENATED
// split-bf16 images: A = [a_hi | a_lo | a_hi] ([M, 3 K], lda), W rows = [w_hi | w_hi | w_lo] ([N, 3 K]) -- one bf16 GEMM over Kcat = 3 K
// whose fp32 accumulators receive hi*hi + lo*hi + hi*lo, the three products of the split-precision scheme (the lo*lo term is below
// 2^-34) -- and every output is written as a (hi, lo) bf16 pair: Q / Qlo, K / Klo, Vt / Vtlo, ffn_out / ffn_lo (same layouts, same
// leading dimension ldf for both halves of the fc1 output).  LayerNorm / RoPE see the fp32 accumulators, gelu_new uses IEEE exp and
// division.  Reference: models/phi.py:657-694, 208-212 evaluated in fp32 (inference_t2i.py:67).
extern "C" int showo_gemm_qkv_fc1_split(const uint16_t* A3, int lda, const uint16_t* W3, int ldw, int Kcat, const float* bias,
                                        const float* qln_w, const float* qln_b, const float* kln_w, const float* kln_b,
                                        const float* cos_tab, const float* sin_tab, uint16_t* Q, uint16_t* Qlo, uint16_t* K, uint16_t* Klo,
                                        uint16_t* Vt, uint16_t* Vtlo, uint16_t* ffn_out, uint16_t* ffn_lo, int ldf, int F, int B, int L,
                                        int nH, int rot, float eps, int pos0, int Lcap, int Lp, int w_tiled, void* stream) {
    if (!ffn_out || !Qlo) return set_error_msg(1, "gemm_qkv_fc1_split: ffn_out and the low-half outputs are required");
    if (Kcat <= 0 || (Kcat % 64)) return set_error_msg(1, "gemm_qkv_fc1_split: Kcat must be a positive multiple of 64");
    return gemm_qkv_impl(A3, lda, W3, ldw, bias, qln_w, qln_b, kln_w, kln_b, cos_tab, sin_tab, Q, K, Vt, B, L, nH, rot, eps,
                         pos0, Lcap, Lp, ffn_out, ldf, F, w_tiled, stream, nullptr, 0, nullptr, Kcat, Qlo, Klo, Vtlo, ffn_lo);
}

// Training forward of the same launch: additionally saves what backward needs -- raw_qkv[m][0..3 nH 64) = bf16(A Wqkv^T + b) (the values
// showo_qkln_rope_bwd recomputes the q/k LayerNorm statistics from, and V) and ffn_pre = bf16(A W1^T + b1) (the input of dgelu) -- and
// derives Q / K / V^T and ffn_out = gelu_new(ffn_pre) from those ROUNDED values, i.e. the bits of showo_gemm_bf16 + showo_qk_prep +
// showo_gemm_bf16 + showo_gelu_bf16 (training/train.py:510-628 forward through models/phi.py:657-694, 208-212) in one launch.
extern "C" int showo_gemm_qkv_fc1_save_bf16(const uint16_t* A, int lda, const uint16_t* Wqkv_fc1, int ldw, const float* bias,
                                            const float* qln_w, const float* qln_b, const float* kln_w, const float* kln_b,
                                            const float* cos_tab, const float* sin_tab, uint16_t* Q, uint16_t* K, uint16_t* Vt,
                                            uint16_t* raw_qkv, int ldraw, uint16_t* ffn_pre, uint16_t* ffn_out, int ldf, int F, int B, int L,
                                            int nH, int rot, float eps, int pos0, int Lcap, int Lp, int w_tiled, void* stream) {
    if (!ffn_out || !ffn_pre || !raw_qkv) return set_error_msg(1, "gemm_qkv_fc1_save: raw_qkv, ffn_pre and ffn_out required");
    if (!Q && (K || Vt)) return set_error_msg(1, "gemm_qkv_fc1_save: Q, K and Vt are given together or not at all (raw-only form)");
    return gemm_qkv_impl(A, lda, Wqkv_fc1, ldw, bias, qln_w, qln_b, kln_w, kln_b, cos_tab, sin_tab, Q, K, Vt, B, L, nH, rot, eps,
                         pos0, Lcap, Lp, ffn_out, ldf, F, w_tiled, stream, raw_qkv, ldraw, ffn_pre);
}

// K-concatenated GEMM: out[M,N] = epilogue([A0 | A1] [W0 | W1]^T + bias), A0 [M,K0] (lda0), A1 [M,K1] (lda1), weight rows
// [W0[n,:] | W1[n,:]] (ldw >= K0 + K1).  Phi's parallel block adds dense(attn) and fc2(ffn) into the same residual row
// (models/phi.py:774-790): x += [attn | ffn] [Wd | W2]^T + (bd + b2) is ONE launch with one residual read-modify-write.
static int gemm_kcat_impl(const uint16_t* A0, int lda0, int K0, const uint16_t* A1, int lda1, int K1, const uint16_t* W,
                          int ldw, const float* bias, void* out, int ldo, const float* resid, int ldr, int M, int N,
                          int epilogue, int w_tiled, int op, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (op != SHOWO_OP_BF16 && op != SHOWO_OP_F16) return set_error_msg(1, "gemm_kcat: op must be SHOWO_OP_BF16 or SHOWO_OP_F16");
    if (K0 <= 0 || K1 <= 0 || (K0 % BK) || (K1 % BK)) return set_error_msg(1, "gemm_kcat: K0, K1 must be positive multiples of 64");
    if (!A0 || !A1 || !W) return set_error_msg(1, "gemm_kcat: null operand");
    if ((lda0 % 8) || (lda1 % 8) || (ldw % 8) || (((uintptr_t)A0) & 15) || (((uintptr_t)A1) & 15) || (((uintptr_t)W) & 15))
        return set_error_msg(1, "gemm_kcat: operands must be 16B aligned with leading dimensions multiples of 8");
    if (epilogue != SHOWO_EPI_RESID_F32) return set_error_msg(1, "gemm_kcat: only SHOWO_EPI_RESID_F32 is implemented (the residual projections)");
    if (!resid) return set_error_msg(1, "gemm_kcat: resid required");
    if ((int64_t)M * lda0 * 2 >= ((int64_t)1 << 32) || (int64_t)M * lda1 * 2 >= ((int64_t)1 << 32) || (int64_t)N * ldw * 2 >= ((int64_t)1 << 32))
        return set_error_msg(1, "gemm_kcat: operand larger than 4 GiB");
    GemmArgs g;
    g.A = A0; g.lda = lda0; g.W = W; g.ldw = ldw; g.Wlo = nullptr; g.bias = bias; g.bias_per_row = 0;
    g.out = out; g.ldo = ldo; g.resid = resid; g.ldr = ldr; g.M = M; g.N = N; g.K = K0 + K1;
    g.gn = 1; g.flags = 0; g.dbg = nullptr;
    g.A2 = A1; g.lda2 = lda1; g.Ksplit = K0;
    g.wtiled = w_tiled ? 1 : 0;
    g.op = op;
    if (w_tiled && ldw != K0 + K1) return set_error_msg(1, "gemm_kcat: a tiled weight has ldw == K0 + K1");
    const bool f32 = (epilogue == SHOWO_EPI_F32 || epilogue == SHOWO_EPI_RESID_F32);
    g.vec_out = ((ldo % 4) == 0) && ((((uintptr_t)out) & (f32 ? 15 : 7)) == 0);
    if (epilogue == SHOWO_EPI_RESID_F32) g.vec_out = g.vec_out && ((ldr % 4) == 0) && ((((uintptr_t)resid) & 15) == 0);
    ProfScope prof(PROF_GEMM, 2.0 * M * N * (K0 + K1), (hipStream_t)stream);
    return gemm2p_dispatch(g, epilogue, (hipStream_t)stream);
}
extern "C" int showo_gemm_kcat_bf16(const uint16_t* A0, int lda0, int K0, const uint16_t* A1, int lda1, int K1, const uint16_t* W,
                                    int ldw, const float* bias, void* out, int ldo, const float* resid, int ldr, int M, int N,
                                    int epilogue, int w_tiled, void* stream) {
    return gemm_kcat_impl(A0, lda0, K0, A1, lda1, K1, W, ldw, bias, out, ldo, resid, ldr, M, N, epilogue, w_tiled, SHOWO_OP_BF16, stream);
}
extern "C" int showo_gemm_kcat_op16(const uint16_t* A0, int lda0, int K0, const uint16_t* A1, int lda1, int K1, const uint16_t* W,
                                    int ldw, const float* bias, void* out, int ldo, const float* resid, int ldr, int M, int N,
                                    int epilogue, int w_tiled, int op, void* stream) {
    return gemm_kcat_impl(A0, lda0, K0, A1, lda1, K1, W, ldw, bias, out, ldo, resid, ldr, M, N, epilogue, w_tiled, op, stream);
}

extern "C" int showo_conv3x3_bf16(const uint16_t* x, const uint16_t* w, const float* bias, const float* resid, float* out,
                                  int B, int Hin, int Win, int Cin, int Cout, int mode, void* stream) {
    if (B <= 0) return 0;
    if (Cin % 64) return set_error_msg(1, "conv3x3: Cin % 64 == 0 required (pad thin inputs to 64 channels)");
    if (mode < 0 || mode > 2) return set_error_msg(1, "conv3x3: bad mode");
    ConvArgs c;
    c.X = x; c.Xlo = nullptr; c.B = B; c.Hin = Hin; c.Win = Win; c.Cin = Cin; c.mode = mode;
    if (mode == 1) { c.Hout = Hin * 2; c.Wout = Win * 2; }
    else if (mode == 2) { c.Hout = Hin / 2; c.Wout = Win / 2; }
    else { c.Hout = Hin; c.Wout = Win; }
    GemmArgs g;
    g.A = nullptr; g.lda = 0; g.W = w; g.ldw = 9 * Cin; g.Wlo = nullptr; g.bias = bias; g.bias_per_row = 0;
    g.out = out; g.ldo = Cout; g.resid = resid; g.ldr = Cout;
    g.M = B * c.Hout * c.Wout; g.N = Cout; g.K = 9 * Cin;
    g.vec_out = ((Cout % 4) == 0) && ((((uintptr_t)out) & 15) == 0) && (!resid || (((uintptr_t)resid) & 15) == 0);
    ConvLoader ld;
    ld.c = c; ld.M = g.M;
    // algorithmic flops of the convolution: real (unpadded) taps x channels
    ProfScope prof(PROF_CONV, 2.0 * g.M * Cout * 9.0 * Cin, (hipStream_t)stream);
    if (gemm_impl_choice(g.M, Cout) == 2) {
        ConvPtr cp;
        cp.c = c; cp.M = g.M;
        return dispatch2(g, cp, resid ? SHOWO_EPI_RESID_F32 : SHOWO_EPI_F32, (hipStream_t)stream);
    }
    return dispatch(g, ld, resid ? SHOWO_EPI_RESID_F32 : SHOWO_EPI_F32, (hipStream_t)stream);
}

// =====================================================================================================
// Split-precision 3x3 convolution on the phase-split template (the VQGAN hot kernel: 84 % of MAGVITv2.get_code).
//   out[M, Cout] = im2col(x)[M, 9 Cin] W[Cout, 9 Cin]^T with x = xh + xl, W = Wh + Wl (bf16 pairs): per fragment pair
//   three MFMAs (Wh xh + Wh xl + Wl xh) into fp32 accumulators.
// Tile 128 (cout) x 256 (pixels) x 32 (k), 8 waves = 2 groups x (2 along n x 2 along m), wave tile 64 x 64 (4 x 4 fragments).
// LDS per k-tile: Wh, Wl [128][32] and Ah, Al [256][32] bf16 = 48 KiB, double-buffered (96 KiB).  Rows are 64 B; a DMA piece
// (1 KiB wave-instruction) is 16 rows; chunk c of row r sits at c ^ ((-(r >> 2)) & 3): a 16x32 fragment read (ds_read_b128,
// lane -> row lane & 15, chunk lane >> 4) touches 16 distinct 16-B slots per lane group.
// The im2col gather (3x3 taps, nearest-2x upsample, pad-(0,1,0,1)+stride-2) is folded into the per-lane DMA source address;
// out-of-image taps read a zero page.  A wave stages the SAME 16 rows of the hi and the lo image, so one address computation
// feeds two DMAs.  Schedule = gemm2p (2 phases per k-tile, groups one barrier apart, counted vmcnt):
//   ph0: read W (8 b128) + A frags m0,m1 (4); DMA W + A-lo rows of tile t+1 (4 pieces); vmcnt(4); 24 MFMAs
//   ph1: read A frags m2,m3 (4);              DMA A-hi rows of tile t+1 (2 pieces);     vmcnt(2); 24 MFMAs
// =====================================================================================================
// store_frag for the fp32 epilogues of a full-width, 16-byte aligned output (vec_out, N % 4 == 0) that also returns what it stored
template <int EPI>
static __device__ inline bool store_frag_vals(const GemmArgs& g, const f32x4& acc, int m, int n, const float (&bn)[4], float (&v)[4]) {
    static_assert(EPI == SHOWO_EPI_F32 || EPI == SHOWO_EPI_RESID_F32, "fp32 epilogues only");
    if (m >= g.M || n >= g.N) return false;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[r] + bn[r] + 0.f;  // store_frag's expression with bias_per_row = 0
    if (EPI == SHOWO_EPI_RESID_F32) {
        const float4 rv = *reinterpret_cast<const float4*>(g.resid + (int64_t)m * g.ldr + n);
        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
    }
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.out) + (int64_t)m * g.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
    return true;
}

constexpr int CS_WROWS = 128, CS_AROWS = 256, CS_BK = 32;
constexpr int CS_BUF = (2 * CS_WROWS + 2 * CS_AROWS) * CS_BK;  // elements per buffer (48 KiB)
constexpr int CS_SMEM = 2 * CS_BUF * 2;                        // 96 KiB

struct ConvRow {  // per-lane im2col state of one staged output pixel
    const bf16_t* img;  // image base of the hi operand
    int oy, ox;
};

// split-K exchange of the conv kernel: gemm_common.h::splitk_exchange for its 4 x 4 fragments per thread (16 float4 = 128 KiB per block).
// Returns true in the block that runs the epilogue, acc = sum over the splits IN SPLIT ORDER (independent of the arrival order).
static __device__ __forceinline__ bool conv_splitk_exchange(const GemmArgs& g, f32x4 (&acc)[4][4], int tile, int split, int* s_last) {
    const int tid = threadIdx.x;
    float4* base = g.ws + (size_t)tile * g.splits * 16 * 512;
    float4* mine = base + (size_t)split * 16 * 512 + tid;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mine[(i * 4 + j) * 512] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __threadfence();  // release: one L2 write-back for the whole block's partial
        const unsigned old = atomicAdd(g.tick + tile, 1u);
        const int last = old == (unsigned)(g.splits - 1);
        if (last) atomicExch(g.tick + tile, 0u);  // ready for the next launch on this stream
        *s_last = last;
        if (last) __threadfence();  // acquire
    }
    __syncthreads();
    if (!*s_last) return false;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < g.splits; ++sp) {
        const float4* p = base + (size_t)sp * 16 * 512 + tid;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = p[(i * 4 + j) * 512];
                acc[i][j][0] += v.x; acc[i][j][1] += v.y; acc[i][j][2] += v.z; acc[i][j][3] += v.w;
            }
    }
    return true;
}

// Epilogue shared by the split-precision conv kernels: fp32 stores of the wave's 64 (cout) x 64 (pixel) tile (acc[i][j]: couts
// n0 + wn * 64 + i * 16 + fg * 4 .. + 3, pixel m0 + arow0 + j * 16 + fr), optionally with the GroupNorm(32) statistics of what was stored.
template <int EPI>
static __device__ __forceinline__ void conv_split_epilogue(const GemmArgs& g, const ConvArgs& c, f32x4 (&acc)[4][4], unsigned char* smem_raw,
                                                           int m0, int n0, int tm, int arow0, int wn, int wave, int fr, int fg, int tid) {
    if (c.gn_part == nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + fg * 4;
            float bn[4];
            load_bias4(g, n, bn);
#pragma unroll
            for (int j = 0; j < 4; ++j) store_frag<EPI>(g, acc[i][j], m0 + arow0 + j * 16 + fr, n, bn);
        }
    } else {
        // Same stores, plus the GroupNorm statistics of what was stored (the next op of the VQGAN block is GroupNorm(32) of this
        // tensor: showo_gn_stats would read it back from HBM).  A lane's 4 columns belong to ONE group (Cout / 32 is a multiple of 4);
        // sums are carried in double and combined in a fixed order (lane butterfly -> wave slots in LDS -> quads -> groups): the
        // partial of a tile has the same bits on every run, like gn_partial_kernel's.
        double gs[4], gq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + fg * 4;
            float bn[4];
            load_bias4(g, n, bn);
            gs[i] = 0.0; gq[i] = 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4];
                if (store_frag_vals<EPI>(g, acc[i][j], m0 + arow0 + j * 16 + fr, n, bn, v)) {
                    gs[i] += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
                    gq[i] += ((double)v[0] * v[0] + (double)v[1] * v[1]) + ((double)v[2] * v[2] + (double)v[3] * v[3]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {  // the 16 lanes fr = 0..15 hold the tile rows of the same 4 columns
                gs[i] += __shfl_xor(gs[i], o, 64);
                gq[i] += __shfl_xor(gq[i], o, 64);
            }
        double* sred = reinterpret_cast<double*>(smem_raw);  // [8 waves][16 quads][2]; every LDS read of the main loop has retired
        __syncthreads();
        if (fr == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sred[(wave * 16 + i * 4 + fg) * 2 + 0] = gs[i];
                sred[(wave * 16 + i * 4 + fg) * 2 + 1] = gq[i];
            }
        }
        __syncthreads();
        if (tid < 64) {
            const int q = tid >> 1, which = tid & 1;  // q: column quad of the block (columns n0 + 4 q ..)
            const int wq = q >> 4, ql = q & 15;
            double a = 0.0;
#pragma unroll
            for (int pw = 0; pw < 4; ++pw) a += sred[((((pw >> 1) * 4 + (pw & 1) * 2 + wq) * 16) + ql) * 2 + which];
            const int qpg = c.gn_cpg >> 2;  // quads per group: 1, 2, 4 or 8
            if (qpg >= 2) a += __shfl_down(a, 2, 64);
            if (qpg >= 4) a += __shfl_down(a, 4, 64);
            if (qpg >= 8) a += __shfl_down(a, 8, 64);
            const int ncol = n0 + 4 * q;
            if ((q % qpg) == 0 && ncol < g.N) c.gn_part[((int64_t)tm * 32 + ncol / c.gn_cpg) * 2 + which] = a;
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(512) void conv2p_split_kernel(GemmArgs g, ConvArgs c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesM = (g.M + CS_AROWS - 1) / CS_AROWS, tilesN = (g.N + CS_WROWS - 1) / CS_WROWS;
    int nwg = tilesM * tilesN, bid = blockIdx.x;
    // split-K (GemmArgs::splits > 1: the 16 x 16 / 32 x 32 levels of the VQGAN launch 32-128 tiles on 256 CUs, with K = 9 Cin up to 4 608):
    // the grid is tiles x splits, block (tile, s) accumulates k-tiles [s per, (s + 1) per), the last block of a tile to arrive sums the
    // partials in split order and runs the epilogue below (GroupNorm statistics included) -- conv_splitk_exchange
    int split = 0;
    if (g.splits > 1) { split = bid / nwg; bid -= split * nwg; }
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tn = bid / tilesM, tm = bid - tn * tilesM;  // pixels fastest: neighbouring blocks share the weight panel
    const int m0 = tm * CS_AROWS, n0 = tn * CS_WROWS;
    const int nk_all = g.K / CS_BK;
    int kt0 = 0, nk = nk_all;  // this block's k-tiles: [kt0, nk)
    if (g.splits > 1) {
        const int per = (nk_all + g.splits - 1) / g.splits;
        kt0 = split * per;
        nk = min(nk_all, kt0 + per);
    }
    const int grp = wave >> 2, wn = wave & 1, wmg = (wave >> 1) & 1;
    const int arow0 = grp * 128 + wmg * 64;  // first tile row (pixel) of this wave

    // ---- DMA roles.  Piece = 16 rows x 64 B; lane -> row lane >> 2, physical chunk lane & 3 = logical chunk ^ f(row >> 2)
    const int prow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((0 - (prow >> 2)) & 3);
    const int koff = lchunk * 8;  // k offset (elements) of this lane inside the 32-wide k-tile
    // W rows: wave w stages rows 16 w .. 16 w + 15 of the weight tile (hi and lo image)
    uint32_t woffb;
    {
        int n = n0 + wave * 16 + prow;
        n = n < g.N ? n : g.N - 1;
        woffb = (uint32_t)(((int64_t)n * g.ldw + koff) * 2);
    }
    const char* whb = reinterpret_cast<const char*>(g.W);
    const char* wlb = reinterpret_cast<const char*>(g.Wlo);
    // A rows: "lo" rows of a wave = its fragments m0,m1 (tile rows arow0 + 0..31), "hi" rows = m2,m3 (arow0 + 32..63).
    // 8 lo pieces and 8 hi pieces per tile; wave w stages lo piece w and hi piece w: tile rows (w >> 1) * 64 + (w & 1) * 16 (+ 32)
    ConvRow rlo, rhi;
    const int64_t xlo_delta = c.Xlo - c.X;
    const int hw = c.Hout * c.Wout;
    {
        const int base = (wave >> 1) * 64 + (wave & 1) * 16 + prow;
        int m = m0 + base;
        m = m < g.M ? m : g.M - 1;
        int b = m / hw, p = m - b * hw;
        rlo.oy = p / c.Wout; rlo.ox = p - rlo.oy * c.Wout;
        rlo.img = c.X + (int64_t)b * c.Hin * c.Win * c.Cin;
        m = m0 + base + 32;
        m = m < g.M ? m : g.M - 1;
        b = m / hw; p = m - b * hw;
        rhi.oy = p / c.Wout; rhi.ox = p - rhi.oy * c.Wout;
        rhi.img = c.X + (int64_t)b * c.Hin * c.Win * c.Cin;
    }
    auto src = [&](const ConvRow& r, int ky, int kx, int cb) -> const bf16_t* {
        int iy, ix;
        bool ok;
        if (c.mode == 2) {
            iy = 2 * r.oy + ky; ix = 2 * r.ox + kx;
            ok = iy < c.Hin && ix < c.Win;
        } else if (c.mode == 1) {
            const int uy = r.oy + ky - 1, ux = r.ox + kx - 1;
            ok = uy >= 0 && ux >= 0 && uy < c.Hout && ux < c.Wout;
            iy = uy >> 1; ix = ux >> 1;
        } else {
            iy = r.oy + ky - 1; ix = r.ox + kx - 1;
            ok = iy >= 0 && ix >= 0 && iy < c.Hin && ix < c.Win;
        }
        return ok ? r.img + ((int64_t)iy * c.Win + ix) * c.Cin + cb + koff : nullptr;
    };
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
    // LDS layout of a buffer (elements): Wh [128][32] | Wl [128][32] | Ah [256][32] | Al [256][32]
    constexpr int O_WL = CS_WROWS * CS_BK, O_AH = 2 * CS_WROWS * CS_BK, O_AL = O_AH + CS_AROWS * CS_BK;
    const int a_lo_row = (wave >> 1) * 64 + (wave & 1) * 16;
#define CS_DMA_W(BUF, K0)                                                                                         \
    do {                                                                                                          \
        bf16_t* d_ = smem + (BUF) * CS_BUF + wave * 16 * CS_BK;                                                   \
        glds16(reinterpret_cast<const bf16_t*>(whb + (size_t)(K0) * 2 + (size_t)woffb), d_);                      \
        glds16(reinterpret_cast<const bf16_t*>(wlb + (size_t)(K0) * 2 + (size_t)woffb), d_ + O_WL);               \
    } while (0)
#define CS_DMA_A(BUF, ROW, R, KY, KX, CB)                                                                         \
    do {                                                                                                          \
        const bf16_t* p_ = src(R, KY, KX, CB);                                                                    \
        bf16_t* d_ = smem + (BUF) * CS_BUF + O_AH + (ROW) * CS_BK;                                                \
        glds16(p_ ? p_ : zero, d_);                                                                               \
        glds16(p_ ? p_ + xlo_delta : zero, d_ + (O_AL - O_AH));                                                   \
    } while (0)

    // ---- fragment read offsets (elements): lane -> row lane & 15, logical chunk lane >> 4
    const int fr = lane & 15, fg = lane >> 4;
    const int lsw = fr * CS_BK + ((fg ^ ((0 - (fr >> 2)) & 3)) << 3);
    const bf16_t* ldsW = smem + (wn * 64) * CS_BK + lsw;
    const bf16_t* ldsA = smem + O_AH + arow0 * CS_BK + lsw;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 wh[4], wl[4], ah[2], al[2];

#define CS_READ_W(BUF)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                               \
        wh[i] = *reinterpret_cast<const bf16x8*>(ldsW + (BUF) * CS_BUF + i * 16 * CS_BK);                         \
        wl[i] = *reinterpret_cast<const bf16x8*>(ldsW + (BUF) * CS_BUF + O_WL + i * 16 * CS_BK);                  \
    }
#define CS_READ_A(BUF, MB)                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                               \
        ah[j] = *reinterpret_cast<const bf16x8*>(ldsA + (BUF) * CS_BUF + ((MB) + j) * 16 * CS_BK);                \
        al[j] = *reinterpret_cast<const bf16x8*>(ldsA + (BUF) * CS_BUF + (O_AL - O_AH) + ((MB) + j) * 16 * CS_BK); \
    }
#define CS_MFMA(MB)                                                                                               \
    do {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
                acc[i][(MB) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], ah[j], acc[i][(MB) + j], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
                acc[i][(MB) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], al[j], acc[i][(MB) + j], 0, 0, 0); \
        if (c.products == 3) {                                                                                    \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                     \
                    acc[i][(MB) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[i], ah[j], acc[i][(MB) + j], 0, 0, 0); \
        }                                                                                                         \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)
    // tap / channel base of k-tile T (Cin % 32 == 0: a k-tile never straddles two taps)
#define CS_TAP(T)                                                                                                 \
    const int k0_ = (T) * CS_BK;                                                                                  \
    const int tap_ = k0_ / c.Cin;                                                                                 \
    const int cb_ = k0_ - tap_ * c.Cin;                                                                           \
    const int ky_ = tap_ / 3, kx_ = tap_ - 3 * (tap_ / 3);
#define CS_TILE(BUF, T)                                                                                           \
    do {                                                                                                          \
        const bool has1 = (T) + 1 < nk;                                                                           \
        CS_TAP((T) + 1)                                                                                           \
        CS_READ_W(BUF)                                                                                            \
        CS_READ_A(BUF, 0)                                                                                         \
        if (has1) {                                                                                               \
            CS_DMA_W((BUF) ^ 1, k0_);                                                                             \
            CS_DMA_A((BUF) ^ 1, a_lo_row, rlo, ky_, kx_, cb_);                                                    \
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                      \
        } else {                                                                                                  \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
        }                                                                                                         \
        bar_raw_fn();                                                                                             \
        CS_MFMA(0);                                                                                               \
        bar_raw_fn();                                                                                             \
        CS_READ_A(BUF, 2)                                                                                         \
        if (has1) {                                                                                               \
            CS_DMA_A((BUF) ^ 1, a_lo_row + 32, rhi, ky_, kx_, cb_);                                               \
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                                      \
        } else {                                                                                                  \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
        }                                                                                                         \
        bar_raw_fn();                                                                                             \
        CS_MFMA(2);                                                                                               \
        bar_raw_fn();                                                                                             \
    } while (0)

    {   // prologue: all of this block's first k-tile
        CS_TAP(kt0)
        CS_DMA_W(0, k0_);
        CS_DMA_A(0, a_lo_row, rlo, ky_, kx_, cb_);
        CS_DMA_A(0, a_lo_row + 32, rhi, ky_, kx_, cb_);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar_raw_fn();
    if (grp == 1) bar_raw_fn();
    int t = kt0;
    for (; t + 1 < nk; t += 2) {
        CS_TILE(0, t);
        CS_TILE(1, t + 1);
    }
    if (t < nk) CS_TILE(0, t);
    if (grp == 0) bar_raw_fn();
    if (g.splits > 1) {
        __shared__ int s_last;
        if (!conv_splitk_exchange(g, acc, tn * tilesM + tm, split, &s_last)) return;
    }

    conv_split_epilogue<EPI>(g, c, acc, smem_raw, m0, n0, tm, arow0, wn, wave, fr, fg, tid);
#undef CS_TILE
#undef CS_TAP
#undef CS_MFMA
#undef CS_READ_A
#undef CS_READ_W
#undef CS_DMA_A
#undef CS_DMA_W
}

// =====================================================================================================
// Split-precision 3x3 convolution with the activation operand staged ONCE per (ky, channel chunk) for all three kx taps.
// conv2p_split_kernel folds the tap into the DMA source address, so the nine taps of a pixel are nine DMA reads of the same L2 lines:
// 48 KiB through LDS-DMA per k-tile for 6.3 MFLOP = 75 GB/s per CU to keep the MFMAs fed, against the ~30 GB/s a CU lands:
// the kernel sat at 39 % MFMA busy on the operand fetch.  Here a 256-pixel tile = R image rows x TW columns (TW = min(Wout, 256),
// R = 256 / TW); a stage holds, per tile row, the TW + 2 input entries of ONE ky (columns c0 - 1 .. c0 + TW, zero outside the image)
// x 32 channels: NE = 256 + 2 R entries of 64 B (hi and lo image).  The k-tile of tap (ky, kx) reads its pixel rows from that stage
// at entry r (TW + 2) + x + kx -- a shifted view, no second fetch: A bytes per k-tile 32 KiB -> (256 + 2 R) / 768 of that (12 KiB),
// W unchanged (16 KiB): 28 instead of 48 KiB per k-tile.  k order: (ky, channel chunk, kx); weights are addressed by k0 = (3 ky + kx)
// Cin + cb as before.  modes 0 (stride 1, pad 1) and 1 (nearest 2x upsample + pad 1: the entry's source is (uy >> 1, ux >> 1));
// the stride-2 form and split-K launches stay on conv2p_split_kernel.  Same wave tiling, fragment layout, swizzle (a function of the
// PHYSICAL LDS row, so 16 consecutive rows from any start are conflict-free) and epilogue as conv2p_split_kernel.
// Schedule: one barrier per k-tile; the DMA of the next k-tile's weights (and, at kx = 0, of the next stage) is issued before the
// fragment reads + 48 MFMAs of the current one and waited for at the barrier.
// =====================================================================================================
constexpr int C3_NE_MAX = 288;                                   // 256 + 2 R, R <= 16
constexpr int C3_A_IMG = C3_NE_MAX * CS_BK;                      // elements of one A image of a stage
constexpr int C3_W_BUF = 2 * CS_WROWS * CS_BK;                   // Wh | Wl of one k-tile
constexpr int C3_A_BUF = 2 * C3_A_IMG;                           // Ah | Al of one stage
constexpr int C3_O_A = 3 * C3_W_BUF;                             // A buffers behind the three W buffers
constexpr int C3_SMEM = (3 * C3_W_BUF + 2 * C3_A_BUF) * 2;       // 122 880 B
// s_waitcnt vmcnt(n) for a wave-uniform run-time n in {0, 2, 4, 6, 8} (the immediate must be a constant)
static __device__ __forceinline__ void c3_wait_vm(int n) {
    if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EPI>
__global__ __launch_bounds__(512) void conv3t_split_kernel(GemmArgs g, ConvArgs c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesM = g.M / CS_AROWS, tilesN = (g.N + CS_WROWS - 1) / CS_WROWS;
    int nwg = tilesM * tilesN, bid = blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tn = bid / tilesM, tm = bid - tn * tilesM;
    const int m0 = tm * CS_AROWS, n0 = tn * CS_WROWS;
    const int grp = wave >> 2, wn = wave & 1, wmg = (wave >> 1) & 1;
    const int arow0 = grp * 128 + wmg * 64;

    // ---- tile geometry
    const int hw = c.Hout * c.Wout;
    const int TW = c.Wout < 256 ? c.Wout : 256, TW2 = TW + 2;
    const int R = 256 / TW, NE = R * TW2, NP = (NE + 15) >> 4;
    const int bimg = m0 / hw, p0 = m0 - bimg * hw;
    const int oy0 = p0 / c.Wout, c0 = p0 - oy0 * c.Wout;
    const bf16_t* img = c.X + (int64_t)bimg * c.Hin * c.Win * c.Cin;
    const int64_t xlo_delta = c.Xlo - c.X;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);

    // ---- DMA roles (piece = 16 entries x 64 B; lane -> entry lane >> 2, physical chunk lane & 3 = logical chunk ^ f(entry >> 2))
    const int prow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((0 - (prow >> 2)) & 3);
    const int koff = lchunk * 8;
    uint32_t woffb;
    {
        int n = n0 + wave * 16 + prow;
        n = n < g.N ? n : g.N - 1;
        woffb = (uint32_t)(((int64_t)n * g.ldw + koff) * 2);
    }
    const char* whb = reinterpret_cast<const char*>(g.W);
    const char* wlb = reinterpret_cast<const char*>(g.Wlo);
    // this lane's entries: pieces wave, wave + 8, wave + 16 (< NP).  colofs = element offset of the entry's source column + chunk
    // (-1: outside the image), uyb = its (upsampled) source row for ky = 0
    int colofs[3], uyb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = (wave + 8 * i) * 16 + prow;
        const int r = e / TW2, xe = e - r * TW2;
        const int ux = c0 + xe - 1;
        const bool ok = e < NE && ux >= 0 && ux < c.Wout;
        colofs[i] = ok ? (c.mode == 1 ? (ux >> 1) : ux) * c.Cin + koff : -1;
        uyb[i] = oy0 + r - 1;
    }
    const uint32_t lds0 = lds_addr_of(smem);
    auto dma_w = [&](int wb, int k0) {
        const uint32_t d = lds0 + (uint32_t)(wb * C3_W_BUF + wave * 16 * CS_BK) * 2;
        glds16_untracked(reinterpret_cast<const bf16_t*>(whb + (size_t)k0 * 2 + (size_t)woffb), d);
        glds16_untracked(reinterpret_cast<const bf16_t*>(wlb + (size_t)k0 * 2 + (size_t)woffb), d + CS_WROWS * CS_BK * 2);
    };
    // piece slot i of this wave (i = 2 exists for the first NP - 16 waves only: has2) of stage (ky, cb) -> A buffer ab: two DMA instructions
    const bool has2 = wave + 16 < NP;
    auto dma_a = [&](int i, int ab, int ky, int cb) {
        const int uy = uyb[i] + ky;
        const bool ok = colofs[i] >= 0 && uy >= 0 && uy < c.Hout;
        const int iy = c.mode == 1 ? (uy >> 1) : uy;
        const bf16_t* p = img + (int64_t)iy * c.Win * c.Cin + colofs[i] + cb;
        const uint32_t d = lds0 + (uint32_t)(C3_O_A + ab * C3_A_BUF + (wave + 8 * i) * 16 * CS_BK) * 2;
        glds16_untracked(ok ? p : zero, d);
        glds16_untracked(ok ? p + xlo_delta : zero, d + C3_A_IMG * 2);
    };

    // ---- fragment read offsets (elements): lane -> row lane & 15, logical chunk lane >> 4; the swizzle follows the physical row
    const int fr = lane & 15, fg = lane >> 4;
    const int lsw = fr * CS_BK + ((fg ^ ((0 - (fr >> 2)) & 3)) << 3);
    const bf16_t* ldsW = smem + (wn * 64) * CS_BK + lsw;
    int aoff[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = arow0 + j * 16;
        const int r = p / TW, x = p - r * TW;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int row = r * TW2 + x + kx + fr;
            aoff[j][kx] = C3_O_A + row * CS_BK + ((fg ^ ((0 - (row >> 2)) & 3)) << 3);
        }
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // k-tile t = 3 s + kx of stage s = (ky, channel chunk).  DMA issued during k-tile t, in this order: W(t + 2) [2 instructions], then the
    // A pieces of stage s + 1 assigned to this kx (slots 0 and 2 at kx = 0, slot 1 at kx = 1, none at kx = 2) [2 each].  vmcnt counts in
    // order, so at the end of k-tile t the wave waits until only the instructions YOUNGER than what k-tile t + 1 needs are in flight:
    //   kx = 0: W(t + 1) is the head of the previous list  -> may stay in flight: W(t + 2) + this tile's A pieces        = 2 + 2 a0
    //   kx = 1: W(t + 1) heads the list of kx = 0          -> its A pieces (2 a0) + W(t + 2) + slot 1                  = 2 a0 + 4
    //   kx = 2: the next stage must be complete             -> W(t + 2) only                                             = 2
    // (a0 = 1 + has2).  Weights have two k-tiles to land, activations one to two.  The last stage (no prefetch left) waits for everything.
    const int cchunks = c.Cin / CS_BK, nstages = 3 * cchunks, nk = 3 * nstages;
    const int a0 = has2 ? 2 : 1;
    dma_w(0, 0);
    dma_w(1, c.Cin);  // tap (0, 1), channel chunk 0
    dma_a(0, 0, 0, 0);
    dma_a(1, 0, 0, 0);
    if (has2) dma_a(2, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar_raw_fn();
    // Two phases per k-tile -- [DMA issue + fragment reads + the counted wait] | barrier | [48 MFMAs] | barrier -- with wave group 1
    // (waves 4-7: the SIMD partners of waves 0-3) running ONE barrier behind group 0, so that on every SIMD one wave multiplies while the
    // other reads its fragments (conv2p_split_kernel's arrangement).  A wave's reads have returned (lgkmcnt(0)) and its DMA of what the
    // next k-tile needs has landed before it passes the mid barrier, i.e. one full phase before any wave touches that data / that buffer.
    if (grp == 1) bar_raw_fn();
    int ky = 0, cbi = 0;  // stage s = ky * cchunks + cbi
    int wb = 0;           // W buffer of k-tile t: t % 3
    for (int s = 0; s < nstages; ++s) {
        const int cb = cbi * CS_BK;
        int nky = ky, ncbi = cbi + 1;
        if (ncbi == cchunks) { ncbi = 0; ++nky; }
        const bool more = s + 1 < nstages;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int t = s * 3 + kx;
            const int wb2 = wb == 0 ? 2 : wb - 1;  // (t + 2) % 3
            if (t + 2 < nk) {
                // k-tile t + 2: tap (ky, 2) of this stage at kx = 0, taps (nky, 0) / (nky, 1) of the next stage at kx = 1 / 2
                const int k2 = kx == 0 ? (ky * 3 + 2) * c.Cin + cb : (nky * 3 + (kx - 1)) * c.Cin + ncbi * CS_BK;
                dma_w(wb2, k2);
            }
            if (more) {
                if (kx == 0) { dma_a(0, (s + 1) & 1, nky, ncbi * CS_BK); if (has2) dma_a(2, (s + 1) & 1, nky, ncbi * CS_BK); }
                if (kx == 1) dma_a(1, (s + 1) & 1, nky, ncbi * CS_BK);
            }
            const bf16_t* wbase = ldsW + wb * C3_W_BUF;
            const bf16_t* abase = smem + (s & 1) * C3_A_BUF;
            bf16x8 wh[4], wl[4], ah[4], al[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wh[i] = *reinterpret_cast<const bf16x8*>(wbase + i * 16 * CS_BK);
                wl[i] = *reinterpret_cast<const bf16x8*>(wbase + CS_WROWS * CS_BK + i * 16 * CS_BK);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ah[j] = *reinterpret_cast<const bf16x8*>(abase + aoff[j][kx]);
                al[j] = *reinterpret_cast<const bf16x8*>(abase + C3_A_IMG + aoff[j][kx]);
            }
            if (!more) c3_wait_vm(0);
            else if (kx == 0) c3_wait_vm(2 + 2 * a0);
            else if (kx == 1) c3_wait_vm(2 * a0 + 4);
            else c3_wait_vm(2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bar_raw_fn();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], ah[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], al[j], acc[i][j], 0, 0, 0);
            if (c.products == 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[i], ah[j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            bar_raw_fn();
            wb = wb == 2 ? 0 : wb + 1;
        }
        ky = nky; cbi = ncbi;
    }
    if (grp == 0) bar_raw_fn();
    conv_split_epilogue<EPI>(g, c, acc, smem_raw, m0, n0, tm, arow0, wn, wave, fr, fg, tid);
}

static std::atomic<long> g_conv3t_launches{0};  // tests assert that their shapes reached this kernel (showo_conv3t_launches)
// shapes the 3-tap-reuse kernel serves (everything else: conv2p_split_kernel)
static bool conv3t_ok(const GemmArgs& g, const ConvArgs& c) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SHOWO_CONV_3TAP"); on = e ? atoi(e) : 1; }
    if (!on || c.mode == 2) return false;
    const int hw = c.Hout * c.Wout;
    if ((hw % CS_AROWS) || (c.Wout % 16) || (c.Cin % CS_BK)) return false;
    if (c.Wout <= 256 ? (256 % c.Wout) != 0 : (c.Wout % 256) != 0) return false;
    return (g.M % CS_AROWS) == 0;
}

template <int EPI>
int launch_conv2p_split(const GemmArgs& g, const ConvArgs& c, hipStream_t s) {
    static bool attr_set = false;
    auto kfn = conv2p_split_kernel<EPI>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, CS_SMEM);
        if (e != hipSuccess) return set_error_hip(e, "hipFuncSetAttribute(conv2p_split)", __FILE__, __LINE__);
        attr_set = true;
    }
    const int tilesM = (g.M + CS_AROWS - 1) / CS_AROWS, tilesN = (g.N + CS_WROWS - 1) / CS_WROWS;
    const int tiles = tilesM * tilesN, nk = g.K / CS_BK;
    // split-K policy (a function of the problem alone: run-to-run identical bits): launches that fill at most half of the 256 CUs split
    // K until about one block per CU, every split at least 16 k-tiles (512 of the 9 Cin contraction) long.  SHOWO_CONV_SPLITK=0: off.
    GemmArgs gs = g;
    gs.splits = 1;
    static int splitk_on = -1;
    if (splitk_on < 0) { const char* e = getenv("SHOWO_CONV_SPLITK"); splitk_on = e ? atoi(e) : 1; }
    const int cus = showo_cu_usable((void*)s);  // the CUs this stream may use (256 unless it is a masked stream)
    if (splitk_on && tiles * 2 <= cus && nk >= 32) {
        int S = cus / tiles;
        if (S > nk / 16) S = nk / 16;
        if (S > 16) S = 16;
        if (S >= 2) {
            const int per = (nk + S - 1) / S;
            S = (nk + per - 1) / per;  // no empty split
        }
        if (S >= 2 && tiles * S <= gemm_splitk_ticks() &&
            gemm_splitk_ws(s, (size_t)tiles * S * 16 * 512 * sizeof(float4), &gs.ws, &gs.tick))
            gs.splits = S;  // (no workspace -- first use inside a stream capture: the unsplit launch is always valid)
    }
    if (gs.splits == 1 && conv3t_ok(g, c)) {
        static bool attr3_set = false;
        auto k3 = conv3t_split_kernel<EPI>;
        if (!attr3_set) {
            hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(k3), hipFuncAttributeMaxDynamicSharedMemorySize, C3_SMEM);
            if (e3 != hipSuccess) return set_error_hip(e3, "hipFuncSetAttribute(conv3t_split)", __FILE__, __LINE__);
            attr3_set = true;
        }
        ++g_conv3t_launches;
        k3<<<dim3(tiles), dim3(512), C3_SMEM, s>>>(gs, c);
        hipError_t e3 = hipGetLastError();
        if (e3 != hipSuccess) return set_error_hip(e3, "conv3t_split launch", __FILE__, __LINE__);
        return 0;
    }
    kfn<<<dim3(tiles * gs.splits), dim3(512), CS_SMEM, s>>>(gs, c);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error_hip(e, "conv2p_split launch", __FILE__, __LINE__);
    return 0;
}

int g_conv_split_impl = -1;  // SHOWO_CONV_SPLIT_IMPL: 1 = 128^2 register-staged kernel, 2 = phase-split kernel (default for M >= 2048)

// ---- split-precision entry points (operands as hi/lo bf16 pairs, fp32 output) -----------------------------
extern "C" int showo_gemm_bf16x3(const uint16_t* A, const uint16_t* Alo, int lda, const uint16_t* W, const uint16_t* Wlo, int ldw,
                                 const float* bias, int bias_per_row, float* out, int ldo, const float* resid, int ldr, int M,
                                 int N, int K, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K % BK) != 0) return set_error_msg(1, "gemm x3: K must be a positive multiple of 64");
    if (!Alo || !Wlo) return set_error_msg(1, "gemm x3: lo operands required");
    if ((lda % 8) || (ldw % 8) || (((uintptr_t)A) & 15) || (((uintptr_t)W) & 15) || (((uintptr_t)Alo) & 15) || (((uintptr_t)Wlo) & 15))
        return set_error_msg(1, "gemm x3: operands must be 16B aligned with lda,ldw multiples of 8");
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.Wlo = Wlo; g.bias = bias; g.bias_per_row = bias_per_row;
    g.out = out; g.ldo = ldo; g.resid = resid; g.ldr = ldr; g.M = M; g.N = N; g.K = K;
    g.vec_out = ((ldo % 4) == 0) && ((((uintptr_t)out) & 15) == 0);
    if (resid) g.vec_out = g.vec_out && ((ldr % 4) == 0) && ((((uintptr_t)resid) & 15) == 0);
    LinearLoader ld;
    ld.A = A; ld.lda = lda; ld.M = M; ld.Alo = Alo;
    ProfScope prof(PROF_GEMM, 6.0 * M * N * K, (hipStream_t)stream);
    return dispatch_split(g, ld, resid ? SHOWO_EPI_RESID_F32 : SHOWO_EPI_F32, (hipStream_t)stream);
}

static int conv3x3_x3_impl(const uint16_t* x, const uint16_t* xlo, const uint16_t* w, const uint16_t* wlo, const float* bias,
                           const float* resid, float* out, double* stats, int B, int Hin, int Win, int Cin, int Cout, int mode, void* stream) {
    if (B <= 0) return 0;
    if (Cin % 64) return set_error_msg(1, "conv3x3 x3: Cin % 64 == 0 required");
    if (mode < 0 || mode > 2) return set_error_msg(1, "conv3x3 x3: bad mode");
    if (!xlo || !wlo) return set_error_msg(1, "conv3x3 x3: lo operands required");
    ConvArgs c;
    c.X = x; c.Xlo = xlo; c.B = B; c.Hin = Hin; c.Win = Win; c.Cin = Cin; c.mode = mode;
    if (mode == 1) { c.Hout = Hin * 2; c.Wout = Win * 2; }
    else if (mode == 2) { c.Hout = Hin / 2; c.Wout = Win / 2; }
    else { c.Hout = Hin; c.Wout = Win; }
    GemmArgs g;
    g.A = nullptr; g.lda = 0; g.W = w; g.ldw = 9 * Cin; g.Wlo = wlo; g.bias = bias; g.bias_per_row = 0;
    g.out = out; g.ldo = Cout; g.resid = resid; g.ldr = Cout;
    g.M = B * c.Hout * c.Wout; g.N = Cout; g.K = 9 * Cin;
    g.vec_out = ((Cout % 4) == 0) && ((((uintptr_t)out) & 15) == 0) && (!resid || (((uintptr_t)resid) & 15) == 0);
    ConvLoader ld;
    ld.c = c; ld.M = g.M;
    const int HW = c.Hout * c.Wout;
    int rc;
    bool fused = false;
    {
        ProfScope prof(PROF_CONV, 2.0 * g.M * Cout * 9.0 * Cin, (hipStream_t)stream);
        if (g_conv_split_impl < 0) { const char* e = getenv("SHOWO_CONV_SPLIT_IMPL"); g_conv_split_impl = e ? atoi(e) : 0; }
        static int gn_fuse = -1;  // SHOWO_CONV_GN_FUSE=0: statistics by the separate two-pass reduction (A/B)
        if (gn_fuse < 0) { const char* e = getenv("SHOWO_CONV_GN_FUSE"); gn_fuse = e ? atoi(e) : 1; }
        // the phase-split kernel needs enough blocks: 256-pixel tiles x Cout / 128, times its split-K (launch_conv2p_split).  Round 4: with
        // split-K it also takes the small launches (a single 256 x 256 image has 256 ... 4 096 output pixels on the 16 x 16 ... 64 x 64
        // levels), which used to fall back to the 128^2 register-staged kernel; SHOWO_CONV_SPLIT_MINM restores a threshold (A/B)
        static int min_m = -1;
        if (min_m < 0) { const char* e = getenv("SHOWO_CONV_SPLIT_MINM"); min_m = e ? atoi(e) : 256; }
        static int products = 0;
        if (!products) { const char* e = getenv("SHOWO_CONV_PRODUCTS"); products = (e && atoi(e) == 2) ? 2 : 3; }
        c.products = products;
        const bool phase_split = g_conv_split_impl == 2 || (g_conv_split_impl != 1 && g.M >= min_m);
        if (phase_split && (int64_t)Cout * g.ldw * 2 < ((int64_t)1 << 32)) {
            if (stats && gn_fuse && (HW % CS_AROWS) == 0 && (Cout % 128) == 0 && Cout <= 1024 && g.vec_out) {
                c.gn_part = stats + (int64_t)B * 64;  // partials behind the [B, 32, 2] result, like showo_gn_stats
                c.gn_cpg = Cout / 32;
                fused = true;
            }
            rc = resid ? launch_conv2p_split<SHOWO_EPI_RESID_F32>(g, c, (hipStream_t)stream)
                       : launch_conv2p_split<SHOWO_EPI_F32>(g, c, (hipStream_t)stream);
        } else {
            rc = dispatch_split(g, ld, resid ? SHOWO_EPI_RESID_F32 : SHOWO_EPI_F32, (hipStream_t)stream);
        }
    }
    if (rc || !stats) return rc;
    if (fused) return showo_gn_finalize(stats + (int64_t)B * 64, stats, B, HW / CS_AROWS, stream);
    return showo_gn_stats(out, stats, B, HW, Cout, stream);
}

extern "C" int showo_conv3x3_bf16x3(const uint16_t* x, const uint16_t* xlo, const uint16_t* w, const uint16_t* wlo, const float* bias,
                                    const float* resid, float* out, int B, int Hin, int Win, int Cin, int Cout, int mode,
                                    void* stream) {
    return conv3x3_x3_impl(x, xlo, w, wlo, bias, resid, out, nullptr, B, Hin, Win, Cin, Cout, mode, stream);
}

// The same convolution plus the GroupNorm(32) statistics of its output (what showo_gn_stats(out, stats, B, Hout * Wout, Cout) would
// return, to the last bits of the double sums): produced by the conv epilogue when a 256-pixel tile never straddles two images,
// by the separate reduction otherwise.  stats: showo_gn_stats_doubles(B, Hout * Wout) doubles.
extern "C" int showo_conv3x3_bf16x3_gn(const uint16_t* x, const uint16_t* xlo, const uint16_t* w, const uint16_t* wlo, const float* bias,
                                       const float* resid, float* out, double* stats, int B, int Hin, int Win, int Cin, int Cout,
                                       int mode, void* stream) {
    if (!stats) return set_error_msg(1, "conv3x3 x3 gn: stats buffer required");
    if (Cout % 128) return set_error_msg(1, "conv3x3 x3 gn: Cout must be a multiple of 128 (GroupNorm(32) over channel quads)");
    return conv3x3_x3_impl(x, xlo, w, wlo, bias, resid, out, stats, B, Hin, Win, Cin, Cout, mode, stream);
}

// number of launches of the 3-tap-reuse conv kernel (conv3t_split_kernel) so far in this process: tests check their coverage with it
extern "C" int64_t showo_conv3t_launches(void) { return (int64_t)g_conv3t_launches.load(); }
