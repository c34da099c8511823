"""Sweep of the decode launches' knobs on the cfg4 shape in ONE process (the per-call hipGraph is re-captured, so a setter takes effect
   at the next call): batch-1 (`mmu_generate`) and batch-4 (`mmu_generate_batch`) decode time per step.
   A configuration = comma-separated assignments on top of the defaults: `pf=next_mb:dense:blocks` (Infinity-Cache prefetch role,
   showo_decode_set_prefetch) and `knob=value` for showo_decode_set_tuning's knobs (co_blocks, batch_co_blocks, batch_ln_blocks,
   ln_blocks, out_blocks).  Configurations are separated by ';'.
   usage: python tools/decode_sweep.py [--new 100] [--reps 3] [--configs "pf=0:0:0;co_blocks=128;batch_ln_blocks=1024,batch_co_blocks=96"]   (GPU)"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--new", type=int, default=100)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--configs", default="pf=0:0:0")
    a = ap.parse_args()
    import showo_amd
    from showo_amd import synthetic
    from showo_amd.prompting_utils import create_attention_mask_for_mmu_vit
    L = showo_amd._lib
    torch.manual_seed(0)
    model = synthetic.random_init_showo(max_batch=1, max_seq=768, w_clip_vit=True).eval()
    Lp, NB = 1 + 28 + 1 + 576 + 1 + 24, 4
    embs, masks = [], []
    for b in range(NB):
        gg = torch.Generator(device="cuda").manual_seed(100 + b)
        embs.append((torch.randn(1, Lp, 2048, device="cuda", generator=gg) * 0.05).contiguous())
        masks.append(create_attention_mask_for_mmu_vit(embs[-1], system_prompt_len=28)[0])

    def t_call(fn):
        best = 1e9
        for _ in range(a.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best, out

    ref1 = refB = None
    DEFAULTS = {"co_blocks": 128, "batch_co_blocks": 128, "batch_ln_blocks": 1024, "ln_blocks": 1024, "out_blocks": 256}  # engine.h DecodeTuning
    print("# configuration | batch-1 ms/step tok/s | batch-4 ms/step agg tok/s (decode steps only)")
    for st in a.configs.split(";"):
        knobs, pf = dict(DEFAULTS), (0, 0, 0)
        for item in st.split(","):
            k, v = item.split("=")
            if k == "pf":
                pf = tuple(int(t) for t in v.split(":"))
            else:
                knobs[k] = int(v)
        L.call("showo_decode_set_prefetch", *pf)
        for k, v in knobs.items():
            L.call("showo_decode_set_tuning", k.encode(), v)
        f1, _ = t_call(lambda: model.mmu_generate(input_embeddings=embs[0], attention_mask=masks[0], max_new_tokens=1, top_k=1))
        t1, o1 = t_call(lambda: model.mmu_generate(input_embeddings=embs[0], attention_mask=masks[0], max_new_tokens=a.new, top_k=1))
        fB, _ = t_call(lambda: model.mmu_generate_batch(input_embeddings=embs, attention_mask=masks, max_new_tokens=1, top_k=1))
        tB, oB = t_call(lambda: model.mmu_generate_batch(input_embeddings=embs, attention_mask=masks, max_new_tokens=a.new, top_k=1))
        o1 = [int(t) for t in o1]
        oB = [[int(t) for t in r] for r in oB]
        if ref1 is None:
            ref1, refB = o1, oB
        same = (o1 == ref1) and (oB == refB)
        s1 = (t1 - f1) / (a.new - 1)
        sB = (tB - fB) / (a.new - 1)
        print(f"{st:44s} | {s1 * 1e3:.4f} {1 / s1:7.1f} | {sB * 1e3:.4f} {NB / sB:7.1f} | tokens_equal={same}", flush=True)


if __name__ == "__main__":
    main()
