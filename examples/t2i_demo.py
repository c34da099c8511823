"""The flow of the reference's `inference_t2i.py` (mode t2i, lines 284-340) on the MI355X path, end to end:

    prompts -> UniversalPrompting('t2i_gen') -> omni mask -> Showo.t2i_generate (18 mask-predict steps, CFG) -> MAGVITv2.decode_code
            -> uint8 NHWC -> PNG files

With --showo / --vq pointing at local checkpoint directories (config.json + weights, the reference's layout) and --tokenizer at a
local HF tokenizer directory this is a real run; without them it uses random-init weights of the true architecture and the
synthetic tokenizer (prompts are then strings of decimal token ids), which exercises exactly the same kernels.

    python examples/t2i_demo.py --prompts "a photo of a cat" "a red cube" --out /tmp/showo_out
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import showo_amd  # noqa: E402
from showo_amd import synthetic  # noqa: E402
from showo_amd.image_utils import images_to_uint8  # noqa: E402
from showo_amd.prompting_utils import UniversalPrompting, create_attention_mask_predict_next  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--showo", default=None, help="local Show-o checkpoint directory (default: random init)")
    ap.add_argument("--vq", default=None, help="local MAGVIT-v2 checkpoint directory (default: random init)")
    ap.add_argument("--tokenizer", default=None, help="local HF tokenizer directory (default: synthetic decimal-id tokenizer)")
    ap.add_argument("--prompts", nargs="*", default=None)
    ap.add_argument("--guidance-scale", type=float, default=5.0)
    ap.add_argument("--timesteps", type=int, default=18)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="t2i_out")
    a = ap.parse_args()
    torch.manual_seed(a.seed)
    if a.tokenizer:
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(a.tokenizer, padding_side="left")
        prompts = a.prompts or ["a photo of a cat"]
    else:
        tok = synthetic.SyntheticTokenizer()
        import numpy as np
        rs = np.random.RandomState(a.seed)
        prompts = a.prompts or [synthetic.random_text(rs, 6 + 3 * i) for i in range(4)]
    uni = UniversalPrompting(tok, max_text_len=128, special_tokens=synthetic.SPECIAL_TOKENS, ignore_id=-100, cond_dropout_prob=0.1)
    B = len(prompts)
    if a.showo:
        model = showo_amd.Showo.from_pretrained(a.showo, max_batch=2 * B, max_seq=387)
    else:
        model = synthetic.random_init_showo(max_batch=2 * B, max_seq=387).eval()
    vq = (showo_amd.MAGVITv2.from_pretrained(a.vq, max_batch=B, max_res=256) if a.vq
          else showo_amd.MAGVITv2(max_batch=B, max_res=256).cuda().eval())
    N, mask_id = model.config.num_vq_tokens, model.mask_token_id
    image_tokens = torch.full((B, N), mask_id, dtype=torch.int64, device="cuda")
    input_ids, _ = uni((list(prompts), image_tokens), 't2i_gen')
    uncond_ids, _ = uni(([''] * B, image_tokens), 't2i_gen')
    sp = uni.sptids_dict
    mask = create_attention_mask_predict_next(torch.cat([input_ids, uncond_ids], dim=0), pad_id=int(sp['<|pad|>']),
                                              soi_id=int(sp['<|soi|>']), eoi_id=int(sp['<|eoi|>']), rm_pad_in_image=True)
    gen = torch.Generator(device="cuda").manual_seed(a.seed)
    with torch.no_grad():
        ids = model.t2i_generate(input_ids=input_ids.contiguous(), uncond_input_ids=uncond_ids.contiguous(), attention_mask=mask,
                                 guidance_scale=a.guidance_scale, temperature=1.0, timesteps=a.timesteps,
                                 noise_schedule=showo_amd.get_mask_chedule("cosine"), generator=gen, config=showo_amd.gen_config())
        ids = torch.clamp(ids, max=model.config.codebook_size - 1, min=0)
        images = images_to_uint8(vq.decode_code(ids)).cpu().numpy()
    os.makedirs(a.out, exist_ok=True)
    from PIL import Image
    for i, im in enumerate(images):
        Image.fromarray(im).save(os.path.join(a.out, f"{i:02d}.png"))
    print(f"wrote {len(images)} images of shape {images.shape[1:]} to {a.out}")


if __name__ == "__main__":
    main()
