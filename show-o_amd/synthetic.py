"""Synthetic workloads for the benchmarks (no network: no tokenizer files, no checkpoints, no datasets).

Everything here goes through the product's own host classes -- `UniversalPrompting` for the sequence layouts, the
device-side mask builders, `Showo` / `MAGVITv2` for the weights -- exactly the calls the reference's scripts make
(`inference_t2i.py:284-333`, `training/train.py:466-585`), with a stand-in tokenizer whose "words" are decimal token ids.
"""
import numpy as np
import torch

from .modeling_showo import Showo
from .prompting_utils import UniversalPrompting, create_attention_mask_predict_next

# configs/showo_demo.yaml:19-24 (Show-o 256/512 on phi-1.5): 50295 text ids + 10 special + 8192 codes + 1 mask id
SHOWO_DEMO = dict(w_clip_vit=False, vocab_size=58498, llm_vocab_size=50295, codebook_size=8192, num_vq_tokens=256)
SPECIAL_TOKENS = ("<|soi|>", "<|eoi|>", "<|sov|>", "<|eov|>", "<|t2i|>", "<|mmu|>", "<|t2v|>", "<|v2v|>", "<|lvg|>")  # inference_t2i.py:55-57


class SyntheticTokenizer:
    """GPT-2 / phi-1.5 numbering (50295 ids, <|endoftext|> = 50256 as bos and eos, nothing added by __call__); a text is a
    string of decimal ids.  Implements what UniversalPrompting touches (training/prompting_utils.py:18-37)."""

    def __init__(self, vocab=50295, eot=50256):
        self.vocab, self.bos_token_id, self.eos_token_id, self.pad_token_id = vocab, eot, eot, None
        self.added = {}

    def __len__(self):
        return self.vocab + len(self.added)

    def add_special_tokens(self, d):
        for key, tok in d.items():
            self.added.setdefault(tok, len(self))
            if key == "pad_token":
                self.pad_token_id = self.added[tok]

    def add_tokens(self, toks):
        for t in toks:
            self.added.setdefault(t, len(self))

    def convert_tokens_to_ids(self, toks):
        return self.added[toks] if isinstance(toks, str) else [self.added[t] for t in toks]

    def __call__(self, texts, truncation=False, **kw):
        return {"input_ids": [[int(w) for w in t.split()] for t in ([texts] if isinstance(texts, str) else texts)]}


def prompting(max_text_len=128, cond_dropout_prob=0.1):
    return UniversalPrompting(SyntheticTokenizer(), max_text_len=max_text_len, special_tokens=SPECIAL_TOKENS, ignore_id=-100,
                              cond_dropout_prob=cond_dropout_prob)


def random_text(rs, n_words, vocab=50256):
    return " ".join(str(int(w)) for w in rs.randint(0, vocab, size=n_words))


def random_init_showo(max_batch, max_seq, ln_jitter=False, **overrides):
    """random-init weights of the true architecture, generated on the GPU: N(0, 0.02) like the reference's Phi init
    (models/phi.py:833-842); LayerNorm weights 1 (or N(1, 0.1) / biases N(0, 0.02) with ln_jitter, so that nothing is a
    structural zero in a throughput run)"""
    kw = dict(SHOWO_DEMO)
    kw.update(overrides)
    with torch.device("meta"):
        model = Showo(kw.pop("w_clip_vit"), kw.pop("vocab_size"), kw.pop("llm_vocab_size"), max_batch=max_batch, max_seq=max_seq, **kw)
    model = model.to_empty(device="cuda")
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "layernorm" in n and n.endswith("weight"):
                p.normal_(1.0, 0.1) if ln_jitter else p.fill_(1.0)
            elif n.endswith("bias"):
                p.normal_(0.0, 0.02) if ln_jitter else p.zero_()
            else:
                p.normal_(0.0, 0.02)
    return model


def t2i_inputs(uni, batch, num_vq_tokens, mask_token_id, seed=0, device="cuda"):
    """BASELINE cfg2 inputs the way inference_t2i.py:284-318 builds them: all-mask image tokens behind `batch` prompts of
    2..37 words (different pad counts per row), the unconditional twins with empty text, and the dense omni mask of the
    CFG-doubled batch (create_attention_mask_predict_next, rm_pad_in_image=True)."""
    rs = np.random.RandomState(seed)
    prompts = [random_text(rs, 2 + (i * 5) % 36) for i in range(batch)]
    image_tokens = torch.full((batch, num_vq_tokens), mask_token_id, dtype=torch.int64, device=device)
    ids_cond, _ = uni((prompts, image_tokens), 't2i_gen')
    ids_uncond, _ = uni(([''] * batch, image_tokens), 't2i_gen')
    sp = uni.sptids_dict
    mask = create_attention_mask_predict_next(torch.cat([ids_cond, ids_uncond], dim=0), pad_id=int(sp['<|pad|>']),
                                              soi_id=int(sp['<|soi|>']), eoi_id=int(sp['<|eoi|>']), rm_pad_in_image=True)
    return ids_cond.contiguous(), ids_uncond.contiguous(), mask
