// IEEE-half (fp16) instances of the production GEMM (gemm2p_kernel.h): Showo.set_precision(2).  Same tiles, phase programs, split-K and
// epilogues as the bfloat16 instances of gemm2p.hip; the MFMA is v_mfma_f32_16x16x32_f16 and the 16-bit outputs are packed with the
// saturating fp16 conversion (common.h Op16<true>).  Inference epilogues only: the save-for-backward and (hi, lo) forms stay bf16.
#include "gemm2p_kernel.h"

namespace showo {

int gemm2p_variant_f16(const GemmArgs& g, int epilogue, int h, hipStream_t s) {
    switch (epilogue) {
        case SHOWO_EPI_BF16: return g2p::launch2p_h<SHOWO_EPI_BF16, true>(g, h, s);
        case SHOWO_EPI_GELU_BF16: return g2p::launch2p_h<SHOWO_EPI_GELU_BF16, true>(g, h, s);
        case SHOWO_EPI_F32: return g2p::launch2p_h<SHOWO_EPI_F32, true>(g, h, s);
        case SHOWO_EPI_RESID_F32: return g2p::launch2p_h<SHOWO_EPI_RESID_F32, true>(g, h, s);
        case EPI_QKV: return g2p::launch2p_h<EPI_QKV, true>(g, h, s);
    }
    return set_error_msg(1, "gemm: epilogue not available with fp16 operands");
}

}  // namespace showo
