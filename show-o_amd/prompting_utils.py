"""Attention-mask construction on the device — drop-in for the mask builders of the reference's
`training/prompting_utils.py` (create_attention_mask_predict_next :466-511, create_attention_mask_for_mmu :591-604,
create_attention_mask_for_mmu_vit :606-624).

Same names, arguments and return values (dense [N,1,L,L] masks: 0 / float(iinfo(int64).min), or bool with
return_inverse_mask=False), computed by one HIP kernel per call instead of torch ops + a Python loop over the batch.
`intervals_*` return an `IntervalMask` instead: the per-row visibility intervals the fused attention consumes directly;
pass it as `attention_mask=` to `Showo.forward` / `t2i_generate` and no [N,1,L,L] tensor is ever built.
"""
import torch

from . import _lib


class IntervalMask:
    """per-row visibility intervals iv int32 [N,L,4] = (lo1, hi1, lo2, hi2) + the 'not representable' flag"""

    def __init__(self, iv, flag):
        self.iv, self.flag = iv, flag
        self.shape = (iv.shape[0], 1, iv.shape[1], iv.shape[1])

    def check(self):
        if self.flag is not None and int(self.flag[0]) != 0:
            raise ValueError("this batch needs more than two visibility runs per row (non-contiguous padding): use the dense mask")
        return self


def _ids(sequence):
    if not sequence.is_cuda:
        raise RuntimeError("show-o_amd builds masks on the GPU: move the token ids to the device (no CPU path exists)")
    return sequence.to(torch.int64).contiguous()


def _predict_next(sequence, pad_id, soi_id, eoi_id, rm_pad_in_image, want_iv, want_dense):
    ids = _ids(sequence)
    N, L = ids.shape
    iv = torch.empty((N, L, 4), dtype=torch.int32, device=ids.device) if want_iv else None
    flag = torch.zeros(4, dtype=torch.int32, device=ids.device) if want_iv else None
    dense = torch.empty((N, 1, L, L), dtype=torch.float32, device=ids.device) if want_dense else None
    _lib.call("showo_mask_predict_next", _lib.ptr(ids), N, L, int(pad_id), int(soi_id), int(eoi_id), int(bool(rm_pad_in_image)),
              _lib.ptr(iv), _lib.ptr(flag), _lib.ptr(dense), _lib.stream())
    return iv, flag, dense


def create_attention_mask_predict_next(sequence, pad_id=128256, soi_id=128257, eoi_id=128258, rm_pad_in_image=False,
                                       return_inverse_mask=True):
    _, _, dense = _predict_next(sequence, pad_id, soi_id, eoi_id, rm_pad_in_image, False, True)
    return dense if return_inverse_mask else dense == 0


def intervals_predict_next(sequence, pad_id=128256, soi_id=128257, eoi_id=128258, rm_pad_in_image=False):
    iv, flag, _ = _predict_next(sequence, pad_id, soi_id, eoi_id, rm_pad_in_image, True, False)
    return IntervalMask(iv, flag)


def _mmu(sequence, eoi_id, want_dense):
    ids = _ids(sequence)
    N, L = ids.shape
    iv = torch.empty((N, L, 4), dtype=torch.int32, device=ids.device)
    dense = torch.empty((N, 1, L, L), dtype=torch.float32, device=ids.device) if want_dense else None
    _lib.call("showo_mask_mmu", _lib.ptr(ids), N, L, int(eoi_id), _lib.ptr(iv), _lib.ptr(dense), _lib.stream())
    return iv, dense


def create_attention_mask_for_mmu(sequence, eoi_id=128258, return_inverse_mask=True):
    _, dense = _mmu(sequence, eoi_id, True)
    return dense if return_inverse_mask else dense == 0


def intervals_for_mmu(sequence, eoi_id=128258):
    return IntervalMask(_mmu(sequence, eoi_id, False)[0], None)


def _mmu_vit(sequence, system_prompt_len, want_dense):
    N, L = sequence.shape[:2]
    dev = sequence.device
    if dev.type != "cuda":
        raise RuntimeError("show-o_amd builds masks on the GPU (no CPU path exists)")
    iv = torch.empty((N, L, 4), dtype=torch.int32, device=dev)
    dense = torch.empty((N, 1, L, L), dtype=torch.float32, device=dev) if want_dense else None
    _lib.call("showo_mask_mmu_vit", N, L, int(system_prompt_len), 576, _lib.ptr(iv), _lib.ptr(dense), _lib.stream())
    return iv, dense


def create_attention_mask_for_mmu_vit(sequence, return_inverse_mask=True, system_prompt_len=0):
    _, dense = _mmu_vit(sequence, system_prompt_len, True)
    return dense if return_inverse_mask else dense == 0


def intervals_for_mmu_vit(sequence, system_prompt_len=0):
    return IntervalMask(_mmu_vit(sequence, system_prompt_len, False)[0], None)
