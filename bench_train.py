"""bench_train.py — stage-1 training step time (BASELINE.json configs[4]: configs/showo_pretraining_stage1.yaml): per GPU
15 t2i + 4 lm + 10 mmu sequences of 387 tokens, forward + backward + gradient exchange + AdamW on the HIP path, plus
the frozen MAGVIT-v2 encode of the 25 images of the batch and the on-device batch construction (MLM corruption, sequence
layouts, omni-mask intervals).  One process per GPU (torchrun); data parallel, weak scaling.
Prints ONE JSON line (rank 0).  `python bench.py --workload train ...` forwards here."""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def synthetic_texts(rs, b_t2i, b_lm, b_mmu):
    """captions of 2..37 words, LM documents longer than the sequence (cut to 387 ids by lm_prompt), 126-word answers"""
    from showo_amd.synthetic import random_text
    t2i = [random_text(rs, 2 + (i * 5) % 36) for i in range(b_t2i)]
    lm = [random_text(rs, 400) for _ in range(b_lm)]
    mmu = [random_text(rs, 126) for _ in range(b_mmu)]
    return t2i, lm, mmu


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-vq", action="store_true", help="leave the frozen VQ encode of the 25 images out of the step")
    ap.add_argument("--workload", default="train")
    a = ap.parse_args(argv)
    rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    import showo_amd
    from showo_amd import synthetic
    from showo_amd.training_utils import build_training_batch
    bt, bl, bm = 15, 4, 10
    torch.manual_seed(0)  # same initial weights on every rank (data parallel replicas)
    model = synthetic.random_init_showo(max_batch=bt + bl + bm, max_seq=387).train()
    trainer = showo_amd.Trainer(model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, coeffs=(1.0, 0.1, 1.0))
    vq = None if a.no_vq else showo_amd.MAGVITv2(max_batch=bt + bm, max_res=256).cuda().eval()
    uni = synthetic.prompting(max_text_len=128, cond_dropout_prob=0.1)
    off = len(uni.text_tokenizer)  # image-token offset (training/train.py:476)
    N, codebook = synthetic.SHOWO_DEMO["num_vq_tokens"], synthetic.SHOWO_DEMO["codebook_size"]
    cfg = type("Cfg", (), {"training": type("Training", (dict,), {"__getattr__": dict.__getitem__})(min_masking_rate=0.0)})
    rs = np.random.RandomState(4 + rank)  # every rank draws its own batch
    torch.manual_seed(4 + rank)
    random.seed(4 + rank)
    texts_t2i, texts_lm, texts_mmu = synthetic_texts(rs, bt, bl, bm)
    images = torch.rand(bt + bm, 3, 256, 256, device="cuda") * 2 - 1
    fixed_codes = torch.randint(0, codebook, (bt + bm, N), device="cuda")  # --no-vq: the tokenizer is left out of the step

    def step():
        # the body of training/train.py:510-628: frozen tokenizer on the 25 images, MLM corruption + sequence layout + omni
        # mask intervals on the device, forward, backward with overlapped gradient exchange, AdamW
        codes = vq.get_code(images) if vq is not None else fixed_codes
        tokens = codes + off
        ids, labels, mask, _, (b1, b2, b3) = build_training_batch(uni, cfg, model.mask_token_id, showo_amd.cosine_schedule,
                                                                  tokens[:bt], list(texts_t2i), list(texts_lm), tokens[bt:],
                                                                  list(texts_mmu))
        return trainer.step(ids, mask, labels, b1, b2, b3, 128)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    losses = None
    for _ in range(a.warmup):
        losses = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        T = (bt + bl + bm) * 387
        ms = dt / a.steps * 1e3
        flop = 3 * T * 2.732e9 + (0 if vq is None else (bt + bm) * 0.355e12)  # 3 x 11 223 x F(387) + encoder
        print(json.dumps({
            "metric": "train step-time (stage-1 mixed batch, fwd+bwd+AdamW" + ("" if vq is None else "+VQ encode") + ")",
            "value": ms, "unit": "ms/step", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
            "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE cfg5: showo_pretraining_stage1 per-GPU batch 15 t2i + 4 lm + 10 mmu x 387 tokens, "
                                   "random-init Show-o 1.45B, AdamW lr 1e-4", "global_batch": (bt + bl + bm) * world, "seq_len": 387,
                       "parallelism": f"dp{world}", "tokens_per_s": T * world / (ms * 1e-3),
                       "algorithmic_tflops_per_gpu": flop / (ms * 1e-3) / 1e12,
                       "losses_last_step": [float(x) for x in losses.cpu()]},
        }))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
