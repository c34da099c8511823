"""bench_train.py — stage-1 training step time (BASELINE.json configs[4]: configs/showo_pretraining_stage1.yaml): per GPU
15 t2i + 4 lm + 10 mmu sequences of 387 tokens, forward + backward + gradient exchange + AdamW on the HIP path, plus
the frozen MAGVIT-v2 encode of the 25 images of the batch.  One process per GPU (torchrun); data parallel, weak scaling.
Prints ONE JSON line (rank 0).  `python bench.py --workload train ...` forwards here."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def build_batch(d, O, rs, b_t2i=15, b_lm=4, b_mmu=10):
    """synthetic stage-1 batch: ids, labels, mask; image tokens are random codes here (the VQ encode of
    the synthetic images is timed separately in the step, its ids do not change the arithmetic)"""
    N, off = d.num_vq_tokens, d.image_offset
    L = 129 + 1 + N + 1
    ids, labels = [], []
    for i in range(b_t2i):
        k = 5 + (i * 5) % 36
        text = [d.t2i_id, 50256] + rs.randint(0, 50256, size=k - 3).tolist() + [50256]
        if rs.rand() < 0.1:
            text = [d.t2i_id, 50256, 50256]  # condition dropout (prompting_utils.py:56-57)
            k = 3
        gt = rs.randint(0, d.codebook, size=N) + off
        ratio = max(np.cos(np.pi / 2 * rs.rand()), 1.0 / N)
        masked = rs.rand(N) < ratio
        img = np.where(masked, d.mask_token_id, gt)
        row = [d.pad_id] * (129 - k) + text + [d.soi_id] + img.tolist() + [d.eoi_id]
        lab = [-100] * 130 + np.where(masked, gt, -100).tolist() + [-100]
        ids.append(row); labels.append(lab)
    for i in range(b_lm):
        row = rs.randint(0, 50256, size=L).tolist()
        ids.append(row); labels.append(list(row))
    for i in range(b_mmu):
        q = rs.randint(0, 50256, size=127).tolist()
        row = [d.mmu_id, d.soi_id] + (rs.randint(0, d.codebook, size=N) + off).tolist() + [d.eoi_id, 50256] + q
        lab = [-100] * (N + 3) + [50256] + q
        ids.append(row[:L]); labels.append(lab[:L])
    ids, labels = torch.tensor(ids), torch.tensor(labels)
    m_t = O.mask_t2i(ids[:b_t2i], d.pad_id, d.soi_id, d.eoi_id)
    m_l = O.mask_t2i(ids[b_t2i:b_t2i + b_lm], d.pad_id, d.soi_id, d.eoi_id, rm_pad_in_image=False)
    m_u = O.mask_mmu(ids[b_t2i + b_lm:], d.eoi_id)
    return ids, labels, torch.cat([m_t, m_l, m_u])


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
    import showo_oracle as O
    import weights as Wt
    d = Wt.ShowoDims()
    bt, bl, bm = 15, 4, 10
    torch.manual_seed(0)  # same initial weights on every rank (data parallel replicas)
    with torch.device("meta"):
        model = showo_amd.Showo(False, d.vocab, d.llm_vocab, codebook_size=d.codebook, num_vq_tokens=d.num_vq_tokens,
                                max_batch=bt + bl + bm, max_seq=387)
    model = model.to_empty(device="cuda").train()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "layernorm" in n and n.endswith("weight"):
                p.fill_(1.0)
            elif n.endswith("bias"):
                p.zero_()
            else:
                p.normal_(0.0, 0.02)
    trainer = showo_amd.Trainer(model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, coeffs=(1.0, 0.1, 1.0))
    vq = None if a.no_vq else showo_amd.MAGVITv2(max_batch=bt + bm, max_res=256).cuda().eval()
    rs = np.random.RandomState(4 + rank)  # every rank draws its own batch
    ids, labels, mask = build_batch(d, O, rs, bt, bl, bm)
    ids, labels, mask = ids.cuda(), labels.cuda(), mask.cuda()
    images = torch.rand(bt + bm, 3, 256, 256, device="cuda") * 2 - 1

    def step():
        if vq is not None:
            vq.get_code(images)  # frozen tokenizer: 25 images per step (training/train.py:547,573)
        return trainer.step(ids, mask, labels, bt, bl, bm, 128)

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
