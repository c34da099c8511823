"""assembles DESIGN.md from the preserved sections 1-7 (head.md, with the round-4 edits applied below) and the round-4 status text (status.md)
   run:  python docs/design_src/build_design.py"""
import os
D = os.path.dirname(os.path.abspath(__file__))
h = open(os.path.join(D, 'head.md')).read()

def rep(a, b, cnt=1):
    global h
    assert a in h, a[:60]
    h = h.replace(a, b, cnt)

rep('''SURVEY.md §8 (the coverage contract). Measured numbers live in "Status & measurements" at the end and in
`profiles/`.''', '''SURVEY.md §8 (the coverage contract). The CURRENT measured numbers are in "Status & measurements (round 4)" at the end and in
`profiles/`; the status sections of rounds 1-3 (with their negative results) moved to `docs/HISTORY.md`.''')
rep('''Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg import `oracle/`''',
'''**Model-scale fixtures (round 4).** `make_golden.py --full` also runs the REAL reference at 1.45 B parameters on every BASELINE config
that the tiny fixtures only covered in shape: `showo_full_cfg3.npz` ([8,1155] CFG-doubled 512x512 inpainting batch, logits rows x cols
subset), `showo_full_cfg4.npz` (the 631-embedding `w_clip_vit` prompt: `mm_projector` splice, `create_attention_mask_for_mmu_vit`,
prefill logits, the first 8 greedy tokens of the reference's no-cache `mmu_generate` with the logits each was drawn from and their
top-2 gaps) and `magvit_512.npz` (1 024 ids, the 13x32x32 latent, strided + dense-crop pixels of `decode_code`); the restatement is
asserted against the reference while they are written (logits 5-6e-6, VQ ids bit-exact).
`attention_lds_model` restates the LDS-tiled attention kernel's arithmetic (32-row waves, 32-key sub-tiles, wave-wide deferred
running maximum) so that the rounding-point oracle rounds P at the scale the kernel does.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg import `oracle/`''')
# kernel table: add rows after the gemm_tn row
rep('''| `attn_bwd_dq_kernel`, `attn_bwd_dkv_kernel`''', '''| `gemm4h_kernel` (`gemm4h.hip`, round 4; variant 5256, opt-in `SHOWO_GEMM_4H=1` / `SHOWO_GEMM_BM=5256`) — the structural experiment VERDICT r3 #2 asked for: FOUR waves with 128 x 128 wave tiles (0.25 LDS reads per MFMA instead of 0.42), all 256 accumulators in AGPRs, `asm volatile` MFMAs with "+a" operands so that the source order IS the instruction stream (3 MFMA : 1 `ds_read_b128`, one LDS-DMA per 8 MFMAs, two fragment register sets, ONE raw barrier per k-tile, W ring 3 / A ring 2, counted `vmcnt(8)`; ISA audited by `tools/check_gemm4h_isa.py`: 0 spills, 0 `v_accvgpr` moves, no compiler waits on the DMA queue), short bodies (MJ = 4 / 2 fragments per wave) for a ragged last tile row, the shared epilogues: bit-identical to the 8-wave family (178 GEMM tests green with the variant forced). **Result: not faster** — 1 285 TF/s at 4096^3 (8-wave 256^2: 1 313), 1 035-1 101 on the [Wqkv;W1] shape with a plain epilogue (3192: 1 146), 655-670 with the fused QKV epilogue (998), half-empty grids on N = 2 048 (`profiles/r4b_gemm4h_harness.txt`): with one wave per SIMD nothing hides the wave's own LDS-DMA issue time (16 pieces x ~70 cycles per k-tile = the ~1 100 cycles a k-tile takes beyond its 2 048 MFMA cycles) nor its epilogue. LDS read instructions were not the limiter | same as the row above | MFMA | same |
| `splitk_coop_finish` (`gemm_common.h`, round 4) — **cooperative split-K reduction**: when every block of a split launch is resident at once (tiles x splits <= CUs no masked stream keeps free) the `splits` blocks of a tile wait for each other (release fence + ticket, one relaxed poll + acquire fence, Guideline 16) and each sums + stores the fragments it owns, in split order (same bits as the last-arriver form): one tile of partial reads per block instead of `splits` tiles in ONE block at 62-70 GB/s. cfg1's `dense|fc2` launch (M = 516, 24 tiles x 10 splits) 94 -> 72 us in the harness, batch-1 t2i 85.9 -> 76.7 ms per image, cfg4 prefill -> first token 5.58 -> 4.52 ms (`profiles/r4d_*`); `SHOWO_GEMM_COOP=0` restores the old form | small-M launches of `dense`+`fc2`, CLIP `fc2` / `out_proj` | HBM (weight stream) + exchange | S x M x N x 4 B written and read once |
| `attn_bwd_dq_kernel`, `attn_bwd_dkv_kernel`''')
rep('''MFMA | 3 × 2·9·Cin·Cout flop executed per output pixel (2·9·Cin·Cout algorithmic) |''', '''MFMA | 3 × 2·9·Cin·Cout flop executed per output pixel (2·9·Cin·Cout algorithmic). Round 4: **split-K** for launches that fill at most half the chip (the 16 x 16 / 32 x 32 levels: 32-128 tiles, K = 9 Cin up to 4 608): grid = tiles x splits ~ 256 blocks, last arriver sums the partials in split order and runs the epilogue incl. the GroupNorm statistics (`conv_splitk_exchange`; `SHOWO_CONV_SPLITK=0` = off) |''')
# numerics section 5 item 2
i = h.index('  2. *`north_star`\'s 1e-3, where it can be decided — per block*')
j = h.index('  3. *`north_star`\'s 1e-3 END TO END — accuracy mode*')
h = h[:i] + '''  2. *`north_star`'s 1e-3, where it can be decided — per block, ALL 24 blocks (round 4)*: `showo_engine_set_collect` hands out the fp32
     residual stream after every block; each block's update `x_out − x_in` (and the final LayerNorm + lm_head) is compared with the
     rounding-point oracle evaluated on the block's OWN input as the GPU computed it. Gates: `rms(Δ)/rms(ref) ≤ 1e-3` on every block
     (measured at full size: worst of 24 blocks 1.7e-4 at [2,387], 1.4e-4 on cfg3 rows), `max|Δ|/max|ref| ≤ 1e-3` on the logits (2.6-3.2e-4)
     and `≤ 2e-3` on the block updates (measured 2.8e-4 … 1.05e-3). The factor 2 on the max statistic is measured, not chosen: the
     ORACLE AGAINST ITSELF, with every value perturbed by 1e-6 relative before it is rounded to bf16 (the size of an fp32
     accumulation-order difference), differs by rel_max 1.08e-3 / rel_rms 3.8e-4 on the block where the GPU differs by 9.0e-4 / 1.5e-4
     (`tests/test_modules_gpu.py::_blockwise_bf16_points`, `floor_block`; `profiles/r4c_*`): one flipped rounding (2⁻⁸ of ONE element) reaches the
     softmax and the K = 10 240 projection behind it, and the maximum over 1.6 M elements finds it. Modelling the kernel's P rounding
     scale (`attention_lds_model`) did not lower the maximum — there is no systematic term left, the comparison sits on its flip floor.
     The K = 128 test model is gated at 4e-3 / 8e-3 (floor ∝ 1/√K; measured 2.4e-4 / 1.4e-3). Kernel-level tests compare against the
     oracle on the *same* bf16-rounded operands (≤1e-3 of the output scale for fp32 outputs, one bf16 ulp for bf16 outputs).
''' + h[j:]
rep('''REFERENCE: full-size logits rel_max 1.3e-5, tiny 8.7e-6 (`tests/test_modules_gpu.py`, gate 1e-3). bf16 operands stay the timed default.''',
'''REFERENCE: full-size logits rel_max 1.3e-5 at [2,387], **1.27e-5 on the cfg3 [8,1155] batch, 1.29e-5 on the cfg4 631-embedding prefill, 1.2-1.4e-5
     on its decode-step logits with the 8 greedy tokens identical** (round-4 fixtures), tiny 8.7e-6 (`tests/test_modules_gpu.py`, gate 1e-3).
     The CLIP tower and `mm_projector` have the same mode (`CLIPVisionTower.set_precision(1)`, `showo_clip_set_precision`: ViT-L/14-336 subset
     rel_max 5.5e-5 where bf16 operands give 2.8e-2; projector 6.8e-6). bf16 operands stay the timed default; what the 1e-3 mode costs is in
     the default bench line (`accuracy_mode`: 2.17 images/s vs 35.7, 16x).
  4. *Training gradients* (bf16 operands, fp32 accumulation / master weights): REAL reference gradients on the 324-row production-path fixture,
     oracle autograd on the full-width 2-layer stage-1 batch (11 223 rows, worst rel_rms 1.9e-2, gate 3e-2) and — round 4 — on the FULL
     24-layer model (1 161 rows, all 245 tensors, gate rel_rms 6e-2: `test_full_size_24_layer_training_gradients_vs_oracle_autograd`).''')
rep('''Optional global-norm clipping (`max_grad_norm`, `training/train.py:614-615`) runs on the flat buffer before the optimizer.''',
'''Optional global-norm clipping (`max_grad_norm`, `training/train.py:614-615`) runs on the flat buffer before the optimizer.
**Round 4, readiness for the first N > 1 run.** (1) `Trainer(reserve_cus=r)` / `SHOWO_RESERVE_CUS`: while an exchange is active the step's
kernels run on a stream created with a CU mask (`showo_stream_create_cu_mask`) so that RCCL's channel kernels find idle CUs; default 0,
justified by the one-GPU experiment `profiles/r4_exchange_contention.txt` (exchange on, unmasked: +3.8 ms per step, 1.8 ms of it GPU time in
`finish()`; masking 8 / 16 CUs costs 20-45 % because split-K / weight-gradient / conv grids are sized to 256 CUs and fall into a second
round). (2) `GradientExchange.exposed_ms_per_bucket()`: exposure per bucket, not just the mean. (3) `Trainer.logging_means`: the
reference's four logging gathers (`training/train.py:603-610`) as ONE all-reduce of a 4-vector. (4) the cooperative split-K reduction
is disabled automatically when a masked stream exists (its residency condition counts the reserved CUs).''')
rep('''## 7. Out of scope (and why)

Control plane and I/O of the reference: the `training/train.py` driver loop (logging, evaluation, checkpoint cadence),''',
'''## 7. Out of scope (and why)

Control plane and I/O of the reference: the `training/train.py` driver loop (logging, evaluation, checkpoint cadence), learning-rate
schedules (`models/lr_schedulers.py`: the reference's own module keeps driving `Trainer.set_lr`; our re-implementation was removed in round 4),''')
rep('''| HBM | **60 B per token** (13·4 B read + 8 B written) |''', '''| HBM | **60 B per token** (13·4 B read + 8 B written); round 4: the NHWC form streams a block's 256 tokens with consecutive 16-byte loads and parks the sign tests as bytes in LDS (2.6 → 6.1 TB/s by rocprofv3 bytes / time) |''')
status = open(os.path.join(D, 'status.md')).read()
open(os.path.join(D, '..', '..', 'DESIGN.md'), 'w').write(h + status)
print('DESIGN.md', len(h) + len(status), 'bytes')
