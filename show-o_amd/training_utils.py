"""Training-step input construction on the device -- the hot-path half of the reference's `training/utils.py`
(`mask_or_random_replace_tokens` :77-154, `get_loss_weight` :73-74) and the batch assembly of `training/train.py:466-585`
(`prepare_inputs_and_labels` + the three task flows), SURVEY.md §8a row T1.

Same names, arguments and return values as the reference.  The masking itself is ONE HIP kernel (`showo_mask_tokens`:
rank-select of the noise row out of LDS) instead of rand -> argsort -> compare -> two torch.where; torch is used for the
random draws (so a seeded run consumes the device generator exactly like the reference: `rand(B)` then `rand(B, N)`) and
for evaluating the caller's `mask_schedule` callable.  `build_training_batch` returns the per-row visibility intervals
(`IntervalMask`) instead of the [B,1,L,L] fp32 masks: the training engine and `Showo.forward` consume them directly.
"""
import math
import random

import torch

from . import _lib
from . import prompting_utils as _pu


def _cfg_get(section, key, default=None):
    """`config.training.get(key, default)` for OmegaConf nodes, dicts and plain namespaces alike"""
    if hasattr(section, "get"):
        return section.get(key, default)
    return getattr(section, key, default)


def get_loss_weight(t, mask, min_val=0.3):
    return 1 - (1 - mask) * ((1 - t) * (1 - min_val))[:, None]


def mask_or_random_replace_tokens(image_tokens, mask_id, config, mask_schedule, is_train=True, noise=None):
    """reference training/utils.py:77-154.  `noise=(timesteps [B], rand [B, N])` injects the two random draws (parity tests);
    by default they are drawn from the device generator in the reference's order."""
    if not image_tokens.is_cuda:
        raise RuntimeError("show-o_amd masks tokens on the GPU: move image_tokens to the device (no CPU path exists)")
    tr = config.training
    batch_size, seq_len = image_tokens.shape
    dev = image_tokens.device
    tokens = image_tokens.to(torch.int64).contiguous()
    ratios = _cfg_get(tr, "eval_mask_ratios", None)
    if not is_train and ratios:
        mask_prob = torch.tensor(random.choices(list(ratios), k=batch_size), device=dev)
    else:
        timesteps = noise[0].to(dev) if noise is not None else torch.rand(batch_size, device=dev)
        mask_prob = mask_schedule(timesteps)
        mask_prob = mask_prob.clip(tr.min_masking_rate)
    num_token_masked = (seq_len * mask_prob).round().clamp(min=1)

    region_prob = _cfg_get(tr, "mask_contiguous_region_prob", None)
    contiguous = False if region_prob is None else random.random() < region_prob

    input_ids = torch.empty_like(tokens)
    labels = torch.empty_like(tokens)
    mask = torch.empty((batch_size, seq_len), dtype=torch.uint8, device=dev)
    # `config.training.get("noise_type", "mask")` is truthy for every non-empty string, so the reference always takes the
    # mask-token branch for the inputs (:132-133); "random_replace" only changes the labels / loss weights (:143-147)
    noise_type = _cfg_get(tr, "noise_type", "mask")
    if not noise_type:
        raise ValueError(f"noise_type {noise_type} not supported")
    predict_all = bool(_cfg_get(tr, "predict_all_tokens", False)) or noise_type == "random_replace"
    if not contiguous:
        u = noise[1].to(dev) if noise is not None else torch.rand(batch_size, seq_len, device=dev)
        u = u.to(torch.float32).contiguous()
        n = num_token_masked.to(torch.int32).contiguous()
        _lib.call("showo_mask_tokens", _lib.ptr(tokens), _lib.ptr(u), _lib.ptr(n), None, 0, batch_size, seq_len, int(mask_id), -100,
                  int(predict_all), _lib.ptr(input_ids), _lib.ptr(labels), _lib.ptr(mask), _lib.stream())
    else:
        resolution = int(seq_len ** 0.5)
        rects = []
        for k in num_token_masked.tolist():  # same host RNG calls, in the same order, as the reference loop (:109-122)
            k = int(k)
            h = random.randint(math.ceil(k / resolution), min(resolution, k))
            h = min(h, resolution)
            w = min(math.ceil(k / h), resolution)
            y0 = random.randint(0, resolution - h)
            x0 = random.randint(0, resolution - w)
            rects.append([y0, y0 + h, x0, x0 + w])
        rect = torch.tensor(rects, dtype=torch.int32, device=dev)
        _lib.call("showo_mask_tokens", _lib.ptr(tokens), None, None, _lib.ptr(rect), resolution, batch_size, seq_len, int(mask_id),
                  -100, int(predict_all), _lib.ptr(input_ids), _lib.ptr(labels), _lib.ptr(mask), _lib.stream())
    loss_weight = get_loss_weight(mask_prob, mask.long()) if predict_all else None
    return input_ids, labels, loss_weight, mask_prob


def build_training_batch(uni_prompting, config, mask_id, mask_schedule, image_tokens_t2i, texts_t2i, texts_lm, image_tokens_mmu,
                         texts_mmu, is_train=True, noise=None):
    """The mixed batch of one optimisation step (reference training/train.py:466-585): t2i rows (MLM-corrupted image tokens
    behind the caption), lm rows padded to the same length, mmu rows (image prefix + caption).

    image_tokens_* are `vq_model.get_code(...)` outputs ALREADY offset by len(tokenizer) (train.py:475-476, 555-556).
    Returns input_ids [B,L], labels [B,L], ONE IntervalMask for the whole batch (t2i rows with rm_pad_in_image, lm rows,
    mmu rows, in that order -- what the reference concatenates as dense masks, train.py:522-577), mask_prob [b_t2i], and
    the batch sizes (b_t2i, b_lm, b_mmu) `Showo.forward` / `Trainer.step` need."""
    sp = uni_prompting.sptids_dict
    pad, soi, eoi = int(sp['<|pad|>']), int(sp['<|soi|>']), int(sp['<|eoi|>'])
    ids_img, lab_img, _, mask_prob = mask_or_random_replace_tokens(image_tokens_t2i, mask_id, config, mask_schedule, is_train, noise)
    ids_t2i, _, lab_t2i = uni_prompting((texts_t2i, ids_img, lab_img), 't2i')
    iv_t2i = _pu.intervals_predict_next(ids_t2i, pad_id=pad, soi_id=soi, eoi_id=eoi, rm_pad_in_image=True)
    dev = ids_t2i.device
    ids_lm, _, lab_lm = uni_prompting((texts_lm, ids_t2i.shape[-1]), 'lm')
    ids_lm, lab_lm = ids_lm.to(dev), lab_lm.to(dev)
    iv_lm = _pu.intervals_predict_next(ids_lm, pad_id=pad, soi_id=soi, eoi_id=eoi)
    ids_mmu, _, lab_mmu = uni_prompting((image_tokens_mmu, texts_mmu), 'mmu')
    iv_mmu = _pu.intervals_for_mmu(ids_mmu, eoi_id=eoi)
    input_ids = torch.cat([ids_t2i, ids_lm, ids_mmu], dim=0)
    labels = torch.cat([lab_t2i, lab_lm, lab_mmu], dim=0)
    flag = iv_t2i.flag | iv_lm.flag  # "needs more than two runs" of either predict-next block (mmu rows never do)
    mask = _pu.IntervalMask(torch.cat([iv_t2i.iv, iv_lm.iv, iv_mmu.iv], dim=0), flag)
    return input_ids, labels, mask, mask_prob, (ids_t2i.shape[0], ids_lm.shape[0], ids_mmu.shape[0])
