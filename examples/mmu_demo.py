"""The w_clip_vit understanding flow of the reference's `inference_mmu.py` (lines 96-175) on the MI355X path:

    image -> CLIP ViT-L/14-336 tower (hidden_states[-2][:,1:]) -> mm_projector -> spliced between the embedded system prompt and the
    question -> mmu_vit visibility intervals -> prefill + KV-cached decode (greedy like the reference's top_k = 1, or --top-k / --temperature)

Checkpoints / tokenizer are optional local directories; without them random-init weights of the true architecture and the
synthetic decimal-id tokenizer are used (same kernels, meaningless text).

    python examples/mmu_demo.py --max-new-tokens 32
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import showo_amd  # noqa: E402
from showo_amd import synthetic  # noqa: E402
from showo_amd.clip_encoder import CLIP_VIT_L_14_336, CLIPVisionTower, vision_state_spec  # noqa: E402
from showo_amd.prompting_utils import UniversalPrompting, intervals_for_mmu_vit  # noqa: E402

SYSTEM_PROMPT_LEN = 28  # inference_mmu.py:36


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--showo", default=None, help="local Show-o (w_clip_vit) checkpoint directory")
    ap.add_argument("--clip", default=None, help="local openai/clip-vit-large-patch14-336 directory")
    ap.add_argument("--tokenizer", default=None)
    ap.add_argument("--image", default=None, help="image file (needs --clip with its preprocessor_config.json); default: noise")
    ap.add_argument("--question", default=None)
    ap.add_argument("--max-new-tokens", type=int, default=100)
    ap.add_argument("--top-k", type=int, default=1)
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    torch.manual_seed(a.seed)
    rs = np.random.RandomState(a.seed)
    if a.tokenizer:
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(a.tokenizer, padding_side="left")
        system, question = "A chat between a curious user and an artificial intelligence assistant.", a.question or "Describe the image."
    else:
        tok = synthetic.SyntheticTokenizer()
        system, question = synthetic.random_text(rs, SYSTEM_PROMPT_LEN), a.question or synthetic.random_text(rs, 24)
    uni = UniversalPrompting(tok, max_text_len=128, special_tokens=synthetic.SPECIAL_TOKENS, ignore_id=-100, cond_dropout_prob=0.1)
    if a.showo:
        model = showo_amd.Showo.from_pretrained(a.showo, max_batch=1, max_seq=1024)
    else:
        model = synthetic.random_init_showo(max_batch=1, max_seq=1024, w_clip_vit=True).eval()
    if a.clip:
        tower = CLIPVisionTower(a.clip, max_batch=1).cuda()
    else:
        spec = vision_state_spec(CLIP_VIT_L_14_336)
        sd = {k: torch.randn(shape) * (0.02 if len(shape) > 1 else 0.0) + (1.0 if k.endswith("norm.weight") or "layer_norm" in k and k.endswith("weight") else 0.0)
              for k, shape in spec.items()}
        tower = CLIPVisionTower("synthetic", config=CLIP_VIT_L_14_336, state_dict=sd, max_batch=1).cuda()
    if a.image and tower.image_processor is not None:
        from PIL import Image
        pixels = tower.image_processor.preprocess(Image.open(a.image).convert("RGB"), return_tensors="pt")["pixel_values"][0].cuda()
    else:
        pixels = torch.randn(3, 336, 336, device="cuda")
    sys_ids = tok([system])["input_ids"][0][:SYSTEM_PROMPT_LEN]
    q_ids = tok([question])["input_ids"][0]
    sp = uni.sptids_dict
    ids = torch.tensor([[int(sp['<|mmu|>'])] + sys_ids + [int(sp['<|soi|>']), int(sp['<|eoi|>'])] + q_ids], device="cuda")
    with torch.no_grad():
        img_emb = model.mm_projector(tower(pixels[None]))                    # [1, 576, 2048]
        txt = model.showo.model.embed_tokens(ids)
        cut = 2 + len(sys_ids)                                               # after <|mmu|>, the system prompt and <|soi|>
        emb = torch.cat([txt[:, :cut], img_emb, txt[:, cut:]], dim=1)
        mask = intervals_for_mmu_vit(emb, system_prompt_len=len(sys_ids))     # per-row intervals, no [1,1,L,L] tensor
        toks = model.mmu_generate(input_embeddings=emb, attention_mask=mask, max_new_tokens=a.max_new_tokens, top_k=a.top_k,
                                  temperature=a.temperature, eot_token=tok.eos_token_id)
    out = [int(t) for t in toks]
    print(f"prompt of {emb.shape[1]} embeddings -> {len(out)} tokens:", out[:24], "..." if len(out) > 24 else "")
    if a.tokenizer:
        print(tok.decode(out, skip_special_tokens=True))


if __name__ == "__main__":
    main()
