// IEEE-half (fp16) instances of the weight-ring GEMM (gemm3w_kernel.h): Showo.set_precision(2); inference epilogues only.
#include "gemm3w_kernel.h"

namespace showo {

int gemm3w_variant_f16(const GemmArgs& g, int epilogue, int rows, hipStream_t s) {
    switch (epilogue) {
        case SHOWO_EPI_BF16: return g3w::launch3w_h<SHOWO_EPI_BF16, true>(g, rows, s);
        case SHOWO_EPI_GELU_BF16: return g3w::launch3w_h<SHOWO_EPI_GELU_BF16, true>(g, rows, s);
        case SHOWO_EPI_F32: return g3w::launch3w_h<SHOWO_EPI_F32, true>(g, rows, s);
        case SHOWO_EPI_RESID_F32: return g3w::launch3w_h<SHOWO_EPI_RESID_F32, true>(g, rows, s);
        case EPI_QKV: return g3w::launch3w_h<EPI_QKV, true>(g, rows, s);
    }
    return set_error_msg(1, "gemm3w: epilogue not available with fp16 operands");
}

}  // namespace showo
