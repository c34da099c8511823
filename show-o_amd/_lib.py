"""ctypes binding of libshowo_hip.so (the C ABI declared in include/showo_hip.h).

PyTorch is used only as the owner of device memory and streams: every call below passes raw device
pointers (`tensor.data_ptr()`) and the current HIP stream.  There is NO CPU / eager fallback: if the
library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SHOWO_LIB_PATH") or os.path.join(_HERE, "libshowo_hip.so")  # SHOWO_LIB_PATH: a second build for same-box A/B runs
_lib = None

c_p = C.c_void_p
c_i = C.c_int
c_i64 = C.c_int64
c_u64 = C.c_uint64
c_u32 = C.c_uint32
c_f = C.c_float

# name -> argtypes (all return int unless noted).  Mirrors include/showo_hip.h one to one.
_PROTOS = {
    "showo_abi_version": [],
    "showo_device_info": [C.POINTER(c_i), C.POINTER(c_i), C.c_char_p, c_i],
    "showo_lfq_pack_nchw": [c_p, c_p, c_i, c_i, c_i, c_p],
    "showo_lfq_pack_nhwc": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "showo_lfq_unpack_nchw": [c_p, c_p, c_i, c_i, c_i, c_p],
    "showo_lfq_unpack_nhwc": [c_p, c_p, c_i, c_i, c_i, c_p],
    "showo_layernorm_f32_bf16": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_p],
    "showo_gemm_bf16": [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "showo_gemm_set_impl": [c_i],
    "showo_gemm_tune": [c_i, c_i, c_p],
    "showo_gemm_counters": [c_p, c_i],
    "showo_gemm_set_coop_polls": [c_i],
    "showo_attn_set_impl": [c_i],
    "showo_decode_set_impl": [c_i],
    "showo_decode_set_prefetch": [c_i, c_i, c_i],
    "showo_decode_set_tuning": [C.c_char_p, c_i],
    "showo_mask_predict_next": [c_p, c_i, c_i, c_i64, c_i64, c_i64, c_i, c_p, c_p, c_p, c_p],
    "showo_mask_mmu": [c_p, c_i, c_i, c_i64, c_p, c_p, c_p],
    "showo_mask_mmu_vit": [c_i, c_i, c_i, c_i, c_p, c_p, c_p],
    "showo_mask_tokens": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i64, c_i64, c_i, c_p, c_p, c_p, c_p],
    "showo_transpose_bf16": [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p],
    "showo_ln_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_p],
    "showo_ln_bwd_blocks": [c_i],
    "showo_ln_bwd_colsum": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_p],
    "showo_qkln_rope_bwd": [c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p],
    "showo_qkln_rope_bwd_blocks": [c_i, c_i],
    "showo_ce_loss": [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_p, c_p, c_p, c_p, c_i, c_p, c_p],
    "showo_embed_bwd": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "showo_adamw": [c_p, c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_f, c_f, c_i, c_p],
    "showo_scale_f32": [c_p, c_i64, c_f, c_p],
    "showo_grad_wire_pack": [c_p, c_p, c_i64, c_f, c_p],
    "showo_grad_wire_unpack": [c_p, c_p, c_i64, c_p],
    "showo_dgelu_bf16": [c_p, c_p, c_p, c_i64, c_p],
    "showo_dgelu_colsum_bf16": [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p],
    "showo_gelu_bf16": [c_p, c_p, c_i64, c_p],
    "showo_train_create": [c_p, c_i, c_i, c_p],
    "showo_train_invalidate_weights": [c_p],
    "showo_trainer_use_intervals": [c_p, c_p, c_p],
    "showo_train_forward": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p],
    "showo_train_forward_embeds": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p],
    "showo_train_input_grad": [c_p, c_p, c_i64, c_p],
    "showo_train_backward": [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_p],
    "showo_train_backward_head": [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_p],
    "showo_train_backward_layer": [c_p, c_i, c_p],
    "showo_train_backward_embed": [c_p, c_p],
    "showo_train_set_loss_weights": [c_p, c_f, c_f, c_f, c_i],
    "showo_train_num_buckets": [c_p],
    "showo_train_bucket": [c_p, c_i, c_p, c_p],
    "showo_train_grad": [c_p, C.c_char_p, c_p, c_p],
    "showo_train_grad_copy": [c_p, C.c_char_p, c_p, c_i64, c_p],
    "showo_train_losses": [c_p, c_p, c_p],
    "showo_train_bind_param": [c_p, C.c_char_p, c_p, c_p, c_p, c_i64],
    "showo_train_adamw_step": [c_p, c_f, c_f, c_f, c_f, c_f, c_i, c_p],
    "showo_attn_fwd_lse": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "showo_head_transpose": [c_p, c_p, c_i, c_i, c_i, c_i, c_i64, c_i64, c_i, c_p],
    "showo_attn_bwd": [c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_i,
                       c_i, c_i, c_i, c_i, c_p],
    "showo_gemm_bf16x3": [c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p],
    "showo_gemm_tn_bf16": [c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "showo_colsum_bf16": [c_p, c_i, c_i, c_i, c_p, c_p, c_i, c_p],
    "showo_conv3x3_bf16x3": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "showo_conv3x3_bf16x3_gn": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "showo_split_f32_bf16": [c_p, c_p, c_p, c_i64, c_p],
    "showo_cast_f32_bf16": [c_p, c_p, c_i64, c_p],
    # 16-bit operand type as an argument (SHOWO_OP_BF16 = 0 | SHOWO_OP_F16 = 1, second to last): the `_bf16` entry points are op = 0
    "showo_cast_f32_op16": [c_p, c_p, c_i64, c_i, c_p],
    "showo_count_f16_saturated": [c_p, c_i64, c_p, c_p],
    "showo_layernorm_f32_op16": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_i, c_p],
    "showo_gemm_op16": [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "showo_gemm_qkv_fc1_op16": [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i,
                                c_f, c_i, c_i, c_i, c_i, c_i, c_p],
    "showo_gemm_kcat_op16": [c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "showo_qk_prep_op16": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_p],
    "showo_attn_fwd_op16": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "showo_engine_set_range_check": [c_p, c_p],
    "showo_copy_b128": [c_p, c_p, c_i64, c_p],
    "showo_embed_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "showo_gemm_qkv_bf16": [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_p],
    "showo_gemm_qkv_fc1_bf16": [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i,
                                c_f, c_i, c_i, c_i, c_i, c_p],
    "showo_gemm_qkv_fc1_save_bf16": [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i,
                                     c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_p],
    "showo_gemm_qkv_fc1_split": [c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i,
                                 c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_p],
    "showo_attn_fwd_split": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "showo_gemm_kcat_bf16": [c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "showo_gemm_tile_weight": [c_p, c_i, c_i, c_i, c_p, c_p],
    "showo_qk_prep": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_p],
    "showo_mask_compress": [c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "showo_attn_fwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "showo_cfg_softmax_sample": [c_p, c_p, c_i, c_f, c_p, c_i64, c_p, c_u64, c_u32, c_p, c_p, c_i, c_i, c_i, c_p],
    "showo_mask_by_topk": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i64, c_i64, c_f, c_f, c_p, c_u64, c_u32, c_p, c_i, c_i, c_p],
    "showo_gn_stats": [c_p, c_p, c_i, c_i, c_i, c_p],
    "showo_gn_stats_doubles": [c_i, c_i],
    "showo_gn_finalize": [c_p, c_p, c_i, c_i, c_p],
    "showo_gn_apply": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_i, c_p],
    "showo_conv3x3_bf16": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "showo_conv_small_f32": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "showo_softmax_rows_bf16": [c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p],
    "showo_pad_cast_bf16": [c_p, c_p, c_p, c_i64, c_i, c_i, c_p],
    "showo_nchw_to_nhwc_f32": [c_p, c_p, c_i, c_i, c_i, c_p],
    "showo_nhwc_to_nchw_f32": [c_p, c_p, c_i, c_i, c_i, c_p],
    "showo_argmax_f32": [c_p, c_i, c_p, c_p],
    "showo_engine_create": [c_p, C.POINTER(c_p)],
    "showo_engine_load": [c_p, C.c_char_p, c_p, c_i64, c_p],
    "showo_engine_slot": [c_p, c_p, c_i64, c_p, c_p],
    "showo_engine_weights_touched": [c_p],
    "showo_engine_missing": [c_p],
    "showo_engine_t2i_captures": [c_p],
    "showo_engine_set_collect": [c_p, c_p],
    "showo_engine_set_precision": [c_p, c_i],
    "showo_engine_get_precision": [c_p],
    "showo_engine_precise_ready": [c_p],
    "showo_engine_precise_fast": [c_p],
    "showo_precise_set_fast": [c_i],
    "showo_engine_use_intervals": [c_p, c_p, c_p],
    "showo_engine_forward": [c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p],
    "showo_engine_forward_rows": [c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_p, c_p],
    "showo_engine_t2i_generate": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i64, c_i, c_i, c_f, c_i, c_p, c_p, c_u64,
                                  c_p, c_p, c_i, c_p, c_p],
    "showo_engine_prefill": [c_p, c_p, c_p, c_p, c_i, c_p, c_p],
    "showo_engine_decode_step": [c_p, c_p, c_p, c_p, c_p],
    "showo_engine_decode_greedy": [c_p, c_p, c_i, c_p, c_p, c_i, c_p],
    "showo_engine_batch_begin": [c_p, c_i, c_i],
    "showo_engine_batch_prefill": [c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p],
    "showo_engine_batch_decode_greedy": [c_p, c_p, c_i, c_p, c_p, c_i, c_p],
    "showo_clip_create": [c_p, c_p],
    "showo_clip_load": [c_p, C.c_char_p, c_p, c_i64, c_p],
    "showo_clip_missing": [c_p],
    "showo_clip_features": [c_p, c_p, c_i, c_p, c_p],
    "showo_clip_set_precision": [c_p, c_i],
    "showo_clip_get_precision": [c_p],
    "showo_clip_precise_ready": [c_p],
    "showo_projector_create": [c_i, c_i, c_i, c_p],
    "showo_projector_load": [c_p, C.c_char_p, c_p, c_i64, c_p],
    "showo_projector_forward": [c_p, c_p, c_i, c_p, c_p],
    "showo_projector_set_precision": [c_p, c_i],
    "showo_projector_precise_ready": [c_p],
    "showo_projector_backward": [c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p],
    "showo_image_resize_crop_normalize": [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p,
                                          c_p, c_p],
    "showo_images_to_uint8": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "showo_mask_downsample_threshold": [c_p, c_i, c_i, c_p, c_p, c_p],
    "showo_sample_topk": [c_p, c_i, c_i, c_f, c_p, c_u64, c_i, c_p, c_p],
    "showo_engine_decode_sample": [c_p, c_p, c_i, c_p, c_p, c_i, c_f, c_p, c_u64, c_i, c_i, c_p],
    "showo_vq_create": [c_p, C.POINTER(c_p)],
    "showo_vq_load": [c_p, C.c_char_p, c_p, c_i64, c_p],
    "showo_vq_missing": [c_p],
    "showo_vq_decode_code": [c_p, c_p, c_i, c_i, c_i, c_p, c_p],
    "showo_vq_get_code": [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p],
    "showo_stream_create_cu_mask": [c_i, C.POINTER(c_p)],
    "showo_stream_destroy": [c_p],
    "showo_cu_reserved_max": [],
    "showo_cu_usable": [c_p],
    "showo_grad_clip_norm": [c_p, c_i64, c_f, c_p, c_p, c_p],
    "showo_grad_clip_ws_doubles": [],
    "showo_cu_census": [c_p, c_i, c_i, c_p],
    "showo_wave_reduce_probe": [c_p, c_p, c_i, c_p],
    "showo_prof_enable": [c_i],
    "showo_prof_reset": [],
    "showo_prof_read": [c_i, C.POINTER(C.c_double), C.POINTER(c_i64), C.POINTER(C.c_double)],
    "showo_prof_set_stride": [c_i],
    "showo_prof_totals": [c_i, c_p, c_p],
}
_I64 = {"showo_gemm_tiled_elems": [c_i, c_i], "showo_conv3t_launches": []}
_VOID = {"showo_engine_destroy": [c_p], "showo_vq_destroy": [c_p], "showo_train_destroy": [c_p], "showo_clip_destroy": [c_p], "showo_projector_destroy": [c_p]}
EXPORTED_SYMBOLS = sorted(list(_PROTOS) + list(_VOID) + list(_I64) + ["showo_last_error"])

EPI_BF16, EPI_GELU_BF16, EPI_F32, EPI_RESID_F32 = 0, 1, 2, 3


class EngineConfig(C.Structure):
    _fields_ = [("hidden", c_i), ("layers", c_i), ("heads", c_i), ("ffn", c_i), ("vocab", c_i),
                ("rotary_dim", c_i), ("max_pos", c_i), ("ln_eps", c_f), ("rope_theta", c_f),
                ("max_batch", c_i), ("max_seq", c_i)]


class VQConfig(C.Structure):
    _fields_ = [("ch", c_i), ("z_channels", c_i),
                ("enc_ch_mult", c_i * 8), ("enc_blocks", c_i * 8), ("enc_levels", c_i),
                ("dec_ch_mult", c_i * 8), ("dec_blocks", c_i * 8), ("dec_levels", c_i),
                ("max_batch", c_i), ("max_res", c_i), ("precision", c_i)]


class ClipConfig(C.Structure):
    _fields_ = [("image_size", c_i), ("patch_size", c_i), ("hidden", c_i), ("heads", c_i), ("ffn", c_i), ("layers", c_i),
                ("run_layers", c_i), ("max_batch", c_i), ("ln_eps", c_f)]


def build(force=False):
    """Compile libshowo_hip.so in-tree for gfx950 (hipcc cross-compiles; no GPU needed)."""
    script = os.path.join(_HERE, "csrc", "build.sh")
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["bash", script])
    return LIB_PATH


def load():
    """Load the shared library (fails loudly if it is absent — there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(show-o_amd has no CPU/eager fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, args in _PROTOS.items():
        if os.environ.get("SHOWO_LIB_PATH") and not hasattr(lib, name):
            continue  # an older build loaded for a same-box A/B run lacks the newest entry points; the shipped library must have all
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_i
    for name, args in _VOID.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = None
    for name, args in _I64.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_i64
    lib.showo_last_error.argtypes = []
    lib.showo_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().showo_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libshowo_hip {what} failed (code {rc}): {msg}")


def call(name, *args):
    check(getattr(load(), name)(*args), name)


def ptr(t):
    """device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda, "show-o_amd kernels take device tensors"
    assert t.is_contiguous(), "show-o_amd kernels take contiguous tensors"
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("show-o_amd needs an AMD GPU (gfx950); there is no CPU fallback")
