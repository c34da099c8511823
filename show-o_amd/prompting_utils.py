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


# ------------------------------------------------------------------------------------------------------------------
# Sequence layout of every task (reference training/prompting_utils.py:17-464 `UniversalPrompting`).
#
# Same constructor, attributes (`sptids_dict`, `pad_id`, `max_text_len` = given + 1, `ignore_id`, `cond_dropout_prob`),
# methods and `__call__(input, task, padding=True, config=None)` dispatch, same return tuples.  Built differently: the
# ragged host-side text lists of a batch become ONE int64 [B, width] block (a single host->device copy), and the
# image-token part is spliced on the device with whole-batch tensor ops -- no per-sample torch.cat / .to(device) chains,
# so the result tensors live where `image_ids` lives and the step's mask builders / kernels consume them directly.
#
# Reference behaviours kept on purpose (they are observable through the return values):
#   * text lists get the <bos> id prepended in place when missing (:48-51);
#   * the returned `attention_masks` are all ones -- the reference computes the pad count AFTER padding the row
#     (:58-60, :107-109, :140-143, :179-182), and their widths are max_text_len + N + 3 (t2i), max_text_len (t2i_gen),
#     max_seq_len (lm), max_text_len - 1 + N + 3 (mmu), one more than the sequence for t2i; nothing downstream reads them
#     (the visibility masks come from create_attention_mask_* above);
#   * a prompt longer than the text window keeps its first window - 1 ids and ends with <eos> (t2i, mmu), while `lm`
#     truncates without re-adding <eos> (:144-149);
#   * condition dropout draws `torch.rand(B)` on the host generator exactly once per t2i / t2v / lvg call (:44).
# ------------------------------------------------------------------------------------------------------------------
class UniversalPrompting:
    def __init__(self, text_tokenizer,
                 special_tokens=("<|soi|>", "<|eoi|>", "<|sov|>", "<|eov|>", "<|t2i|>", "<|mmu|>", "<|t2v|>", "<|v2v|>", "<|lvg|>"),
                 max_text_len=8000, max_seq_len=377, ignore_id=-100, cond_dropout_prob=0.1):
        self.text_tokenizer = text_tokenizer
        self.text_tokenizer.add_special_tokens({'pad_token': '[PAD]'})
        self.text_tokenizer.add_tokens(list(special_tokens))
        self.sptids_dict = {tok: torch.tensor(self.text_tokenizer.convert_tokens_to_ids([tok])) for tok in special_tokens}
        self.sptids_dict['<|sot|>'] = torch.tensor([self.text_tokenizer.bos_token_id])
        self.sptids_dict['<|eot|>'] = torch.tensor([self.text_tokenizer.eos_token_id])
        self.sptids_dict['<|pad|>'] = torch.tensor([self.text_tokenizer.pad_token_id])
        self.max_text_len = max_text_len + 1  # the task token is prepended to the text window
        self.pad_id = self.text_tokenizer.convert_tokens_to_ids('[PAD]')
        self.ignore_id = ignore_id
        self.cond_dropout_prob = cond_dropout_prob

    # -- host side: ragged id lists -> one [B, width] block ----------------------------------------------------------
    def _sp(self, name):
        return int(self.sptids_dict[name])

    def _text_block(self, text_ids, width, head=(), left_pad=True, dropped=None, keep_eos=True):
        bos, eos = self.text_tokenizer.bos_token_id, self.text_tokenizer.eos_token_id
        block = torch.full((len(text_ids), width), self.pad_id, dtype=torch.int64)
        for i in range(len(text_ids)):
            if len(text_ids[i]) == 0:
                text_ids[i] = [bos]
            elif text_ids[i][0] != bos:
                text_ids[i] = [bos] + text_ids[i]
            row = list(head) + ([bos] if dropped is not None and dropped[i] else text_ids[i]) + [eos]
            if len(row) > width:
                row = row[:width - 1] + [eos] if keep_eos else row[:width]
            r = torch.tensor(row, dtype=torch.int64)
            if left_pad:
                block[i, width - len(row):] = r
            else:
                block[i, :len(row)] = r
        return block

    @staticmethod
    def _col(value, like):
        return torch.full((like.shape[0], 1), int(value), dtype=torch.int64, device=like.device)

    def _ones(self, B, width, device):
        return torch.ones((B, width), dtype=torch.int64, device=device)

    def _gen_like(self, task_tok, open_tok, close_tok, text_ids, image_ids, labels=None, dropout=False):
        """[task] [bos] text [eos] (left padded to max_text_len) [open] image [close]"""
        dropped = None
        if dropout:
            dropped = (torch.rand(len(text_ids)) < self.cond_dropout_prob).tolist()
        device = image_ids.device
        text = self._text_block(text_ids, self.max_text_len, head=(self._sp(task_tok),), left_pad=True, dropped=dropped).to(device)
        image_ids = image_ids.to(torch.int64)
        seq = torch.cat([text, self._col(self._sp(open_tok), text), image_ids, self._col(self._sp(close_tok), text)], dim=1)
        if labels is None:
            return seq, self._ones(len(text_ids), self.max_text_len, device)
        lab = torch.cat([text, self._col(self._sp(open_tok), text), labels.to(torch.int64), self._col(self._sp(close_tok), text)], dim=1)
        lab = torch.where(lab == self.pad_id, torch.full_like(lab, self.ignore_id), lab)
        return seq, self._ones(len(text_ids), self.max_text_len + image_ids.shape[-1] + 3, device), lab

    # -- the reference's task methods ------------------------------------------------------------------------------
    def t2i_prompt(self, text_ids, image_ids, labels):
        return self._gen_like('<|t2i|>', '<|soi|>', '<|eoi|>', text_ids, image_ids, labels, dropout=True)

    def t2i_gen_prompt(self, text_ids, image_ids):
        return self._gen_like('<|t2i|>', '<|soi|>', '<|eoi|>', text_ids, image_ids)

    def t2v_prompt(self, text_ids, image_ids, labels):
        return self._gen_like('<|t2v|>', '<|sov|>', '<|eov|>', text_ids, image_ids, labels, dropout=True)

    def t2v_gen_prompt(self, text_ids, image_ids):
        return self._gen_like('<|t2v|>', '<|sov|>', '<|eov|>', text_ids, image_ids)

    def lvg_prompt(self, text_ids, image_ids, labels):
        out = self._gen_like('<|t2i|>', '<|soi|>', '<|eoi|>', text_ids, image_ids, labels, dropout=True)
        torch.rand(len(text_ids))  # the reference draws a second, unused vector (:316); keep the host RNG stream aligned
        return out

    def lvg_gen_prompt(self, text_ids, image_ids):
        return self._gen_like('<|t2i|>', '<|soi|>', '<|eoi|>', text_ids, image_ids)

    def lm_prompt(self, text_ids, max_seq_len):
        text = self._text_block(text_ids, max_seq_len, left_pad=False, keep_eos=False)
        lab = torch.where(text == self.pad_id, torch.full_like(text, self.ignore_id), text)
        return text, torch.ones_like(text), lab

    def mmu_prompt(self, image_ids, text_ids):
        device = image_ids.device
        width = self.max_text_len - 1  # the task token sits in front of the image here
        text = self._text_block(text_ids, width, left_pad=False).to(device)
        image_ids = image_ids.to(torch.int64)
        seq = torch.cat([self._col(self._sp('<|mmu|>'), text), self._col(self._sp('<|soi|>'), text), image_ids,
                         self._col(self._sp('<|eoi|>'), text), text], dim=1)
        lab = torch.cat([torch.full((text.shape[0], image_ids.shape[-1] + 3), self.ignore_id, dtype=torch.int64, device=device),
                         torch.where(text == self.pad_id, torch.full_like(text, self.ignore_id), text)], dim=1)
        return seq, self._ones(len(text_ids), width + image_ids.shape[-1] + 3, device), lab

    def i2v_prompt(self, image_ids, video_ids):
        pass

    def mask_prompt(self):
        pass

    def __call__(self, input, task, padding=True, config=None):
        tok = self.text_tokenizer
        if task == "t2i":
            return self.t2i_prompt(tok(input[0])['input_ids'], input[1], input[2])
        if task == "t2v":
            return self.t2v_prompt(tok(input[0])['input_ids'], input[1], input[2])
        if task == "t2i_plus_lm":
            text_ids = tok(input[0])['input_ids']
            nb = config.training.batch_size
            return self.t2i_prompt(text_ids[:nb], input[1], input[2]), self.lm_prompt(text_ids[nb:], input[3])
        if task == "t2i_gen":
            return self.t2i_gen_prompt(tok(input[0])['input_ids'], input[1])
        if task == "t2v_gen":
            return self.t2v_gen_prompt(tok(input[0])['input_ids'], input[1])
        if task == "lm":
            return self.lm_prompt(tok(input[0], truncation=True)['input_ids'], input[1])
        if task == "mmu":
            return self.mmu_prompt(input[0], tok(input[1])['input_ids'])
        if task == "lvg":
            return self.lvg_prompt(tok(input[0])['input_ids'], input[1], input[2])
        if task == "lvg_gen":
            return self.lvg_gen_prompt(tok(input[0])['input_ids'], input[1])
        raise NotImplementedError
