"""Shared test helpers.  The oracle (oracle/showo_oracle.py) is the CHECKER; the thing under test is always
the HIP path reached through the C ABI (ctypes) or through the drop-in classes in show-o_amd/."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import showo_oracle as O  # noqa: E402
import weights as Wt  # noqa: E402


def golden(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def pkg():
    import showo_amd
    return showo_amd


def lib():
    return pkg()._lib


_KEEP = []  # device tensors created by dev() stay alive until the end of the test: the C ABI takes raw pointers,
            # and a temporary freed before the launch could be recycled by the caching allocator


def dev(x, dtype=None):
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    t = t.cuda().contiguous()
    _KEEP.append(t)
    return t


def release():
    torch.cuda.synchronize() if torch.cuda.is_available() else None
    _KEEP.clear()


def to_bf16_bits(t):
    """fp32 tensor -> uint16 bit pattern tensor (int16 storage) of the bf16 rounding"""
    return t.to(torch.bfloat16).view(torch.int16)


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def from_bf16_bits(t):
    return t.view(torch.bfloat16).to(torch.float32)


def relerr(a, b):
    """max|a-b| / max|b| and rms(a-b)/rms(b)"""
    a, b = a.double().cpu(), b.double().cpu()
    d = (a - b)
    return float(d.abs().max() / b.abs().max().clamp(min=1e-30)), float(d.pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp(min=1e-30))


def tiny_dims():
    return Wt.ShowoDims(**Wt.TINY)


def tiny_state(seed=11):
    d = tiny_dims()
    return d, Wt.make_showo_state(d, seed=seed)


def build_showo(d, sd_np, max_batch=8, max_seq=128):
    """drop-in Showo on the GPU with the given (numpy) state dict"""
    S = pkg().Showo
    with torch.device("meta"):  # parameters are materialised directly on the GPU (no CPU init of 1.45 B values)
        m = S(w_clip_vit=d.w_clip_vit, vocab_size=d.vocab, llm_vocab_size=d.llm_vocab, codebook_size=d.codebook,
              num_vq_tokens=d.num_vq_tokens, hidden_size=d.hidden, intermediate_size=d.ffn, num_hidden_layers=d.layers,
              num_attention_heads=d.heads, max_position_embeddings=d.max_pos, max_batch=max_batch, max_seq=max_seq)
    m = m.to_empty(device="cuda")
    m.load_state_dict(O.to_torch(sd_np), strict=True)
    return m.eval()


def gen_config(d):
    return pkg().gen_config(llm_vocab_size=d.llm_vocab, num_new_special_tokens=d.num_new_special_tokens,
                            num_vq_tokens=d.num_vq_tokens, max_seq_length=d.max_text_len)
