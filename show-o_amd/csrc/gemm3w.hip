// gemm3w: the m-split 2-phase GEMM of gemm2p.hip with a THREE-deep LDS ring for the weight operand.
//
// Why: in gemm2p's m-split program the load segment of ph0 issues 6 of the 8 DMA pieces of a k-tile (all of W + A-lo) next to its
// 16 ds_read_b128, ph1 only 2 (A-hi) next to 8 reads.  A DMA issue costs 100-185 cycles in such a segment (MI355X_MICROARCH.md,
// "LDS-DMA piece issue cost"), so ph0's load segment (~1000 cycles) is twice as long as the partner group's 32-MFMA segment
// (~544 cycles) it is supposed to hide behind, and every k-tile pays the difference twice.  With one more weight buffer the weight
// pieces of tile T+2 can be issued in EITHER phase of tile T (their data is not needed before ph0 of T+2), which balances the two
// load segments at 4 DMA pieces each and gives the weight stream -- the operand that comes cold from HBM in the 24-layer stack -- a
// lead of 1.5 k-tiles instead of half a phase.
//
// LDS: W ring 3 x 32 KiB at 0 / 32 / 64 KiB, A double buffer 2 x 32 KiB at 96 / 128 KiB = 160 KiB (all of a CU's LDS).
// Tile T reads W[T % 3] and A[T % 2].  Phase program (group 1 runs one barrier behind group 0, as in gemm2p):
//   ph0(T): ds_read all W fragments + the 4 lo A fragments | DMA A-lo(T+1) (2 pieces), W rows 0..127 of tile T+2 (2) | vmcnt(6)
//   ph1(T): ds_read the hi A fragments                     | DMA A-hi(T+1) (2),        W rows 128..255 of tile T+2 (2) | vmcnt(6)
//   (A pieces are issued BEFORE the W pieces of the same phase: VMEM returns in order, so vmcnt(6) after the 4 new pieces retires
//    everything up to and including the A pieces of the PREVIOUS phase and leaves that phase's 2 W pieces + these 4 in flight.)
// Hazards:
//   RAW  A-lo(T+1): issued ph0(T), retired by ph1(T)'s wait, read in ph0(T+1)  (2 barriers after every wave's wait)
//        A-hi(T+1): issued ph1(T), retired by ph0(T+1)'s wait, read in ph1(T+1)
//        W(T+2):    halves issued ph0(T) / ph1(T), retired by the waits of ph0(T+1) / ph1(T+1), read in ph0(T+2)
//   WAR  W[(T+2) % 3] = W[(T-1) % 3] was last read in ph0(T-1): its reads retire at the head of that phase's MFMA segment, at least
//        three barriers before any wave's ph0(T) DMA issue.  A: as in gemm2p (re-staged two phases after its last read).
// The last two tiles issue fewer pieces; their waits fall back to vmcnt(0) (over-waiting is safe).
// n-split form (NS = true, 8..12 fragments; gemm2p's balanced program for short tiles) on the same ring:
//   ph0(T): ds_read W fragments 0,1 + ALL A fragments | DMA A(T+1) (NPW pieces)                       | no wait needed
//   ph1(T): ds_read W fragments 2,3                   | DMA all of W(T+2) (4 pieces)                  | vmcnt(4)
//   vmcnt(4) after the 4 new W pieces retires A(T+1) and every piece of W(T+1): both are read from ph0(T+1) on; W(T+2) has a
//   full k-tile of lead.  WAR: W[(T+2) % 3] = W[(T-1) % 3] was last read in ph1(T-1), three barriers before ph1(T)'s issue.
// K-concatenated A operand, tiled weights, XCD-aware tile order, epilogues: identical to gemm2p (gemm_common.h).
#include "gemm3w_kernel.h"

namespace showo {

// variant codes 2256 / 2240 / 2224 / 2208 (m-split), 3192 / 3176 / 3160 / 3144 (n-split) and 4192 / 4176 / 4160 / 4144 (n-split,
// buffer-descriptor DMAs) of gemm2p's tile table
int gemm3w_launch(const GemmArgs& g, int epilogue, int rows, hipStream_t s) {
    if (g.op) return gemm3w_variant_f16(g, epilogue, rows, s);  // IEEE-half operands (precision 2)
    switch (epilogue) {
        case SHOWO_EPI_BF16: return g3w::launch3w_h<SHOWO_EPI_BF16>(g, rows, s);
        case SHOWO_EPI_GELU_BF16: return g3w::launch3w_h<SHOWO_EPI_GELU_BF16>(g, rows, s);
        case SHOWO_EPI_F32: return g3w::launch3w_h<SHOWO_EPI_F32>(g, rows, s);
        case SHOWO_EPI_RESID_F32: return g3w::launch3w_h<SHOWO_EPI_RESID_F32>(g, rows, s);
        case EPI_QKV: return g3w::launch3w_h<EPI_QKV>(g, rows, s);
        case EPI_QKV_SPLIT: return g3w::launch3w_h<EPI_QKV_SPLIT>(g, rows, s);
    }
    return set_error_msg(1, "gemm3w: unknown epilogue");
}

}  // namespace showo
