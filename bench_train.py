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


def cpu_baseline(n_seq, budget_s=40.0, threads=None):
    """oracle (CPU restatement of the reference, fp32, torch autograd) forward + backward of a 3-sequence micro-batch (1 t2i +
    1 lm + 1 mmu x 387 tokens, the three losses of models/modeling_showo.py:80-98 weighted 1.0 / 0.1 / 1.0) + one torch AdamW
    step over the 1.45 B parameters; the forward + backward time is scaled linearly to the n_seq sequences of the stage-1
    per-GPU batch (the full batch does not fit host memory with fp32 activations: BASELINE.md section 3).  The frozen VQ encode
    is left out of the CPU estimate (favouring the CPU).  The ONLY place in this file that touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bench
    import showo_oracle as O
    import weights as Wt
    d = Wt.ShowoDims()
    cand = None
    if threads is None:
        threads, cand = bench._pick_threads()
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    sd = {k: (torch.randn(shape, generator=g) * std + mean).requires_grad_(True) for k, (shape, std, mean) in Wt.showo_state_spec(d).items()}
    L = 387
    rs = np.random.RandomState(0)
    t2i = [d.pad_id] * 100 + [d.t2i_id, 50256] + rs.randint(0, 50256, size=26).tolist() + [50256, d.soi_id] + [d.mask_token_id] * 256 + [d.eoi_id]
    lm = rs.randint(0, 50256, size=L).tolist()
    mmu = [d.mmu_id, d.soi_id] + (rs.randint(0, 8192, size=256) + d.llm_vocab + d.num_new_special_tokens).tolist() + [d.eoi_id] + rs.randint(0, 50256, size=128).tolist()
    ids = torch.tensor([t2i, lm, mmu])
    causal = torch.zeros(1, 1, L, L)
    causal[0, 0][torch.triu(torch.ones(L, L, dtype=torch.bool), 1)] = O.NEG_MASK
    mask = torch.cat([O.mask_t2i(ids[:1], d.pad_id, d.soi_id, d.eoi_id), causal, O.mask_mmu(ids[2:], d.eoi_id)])
    labels = ids.clone()
    labels[0, :131] = -100

    def fwd_bwd():
        _, l1, l2, l3 = O.showo_forward(sd, d, ids, attention_mask=mask, labels=labels, batch_size_t2i=1, batch_size_lm=1, batch_size_mmu=1,
                                        max_seq_length=128)
        (l1 + 0.1 * l2 + l3).backward()

    t0 = time.time()
    fwd_bwd()  # warm-up (page-in, thread pool)
    t_warm = time.time() - t0
    tried = {threads: round(t_warm, 1)}
    if cand is not None and threads != 8:  # stand-alone run: the GEMM microbenchmark's pick must not lose to the 8-thread pool on a real step
        torch.set_num_threads(8)
        t0 = time.time()
        fwd_bwd()
        tried[8] = round(time.time() - t0, 1)
        if tried[8] < t_warm:
            threads, t_warm = 8, tried[8]
        torch.set_num_threads(threads)
    n = int(max(1, min(3, budget_s // max(t_warm, 1e-3))))
    t0 = time.time()
    for _ in range(n):
        fwd_bwd()
    t_fb = (time.time() - t0) / n
    opt = torch.optim.AdamW(list(sd.values()), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    t0 = time.time()
    opt.step()
    t_opt = time.time() - t0
    est = t_fb * n_seq / 3.0 + t_opt
    return {"value": est * 1e3, "unit": "ms/step", "cores": threads, "kind": "port", "step_seconds_by_threads": {str(k): v for k, v in tried.items()},
            "sample": f"{n} x forward+backward of a 3-sequence micro-batch (1 t2i + 1 lm + 1 mmu x 387 tokens, fp32 oracle autograd) = "
                      f"{t_fb:.1f}s each, scaled x{n_seq}/3, + one torch AdamW step over 1.45 B parameters = {t_opt:.1f}s; VQ encode omitted"}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-vq", action="store_true", help="leave the frozen VQ encode of the 25 images out of the step")
    ap.add_argument("--workload", default="train")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="skip the per-launch HIP events (A/B of their overhead)")
    ap.add_argument("--event-stride", type=int, default=5)
    ap.add_argument("--roofline-steps", type=int, default=1, help="steps of the separate HIP-event leg after the timed region that feeds `roofline` (0: skip)")
    ap.add_argument("--wire", default="bf16", help="gradient wire of the data-parallel exchange: bf16 (default) | fp32")
    a = ap.parse_args(argv)
    import bench
    bench.self_launch(a.gpus, sys.argv[0])  # `--gpus N` without a launcher: re-exec as N ranks under torch.distributed.run
    rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if bench.dry_run():  # launch plumbing only (bench.dry_run): rendezvous of the ranks over gloo, one all-reduce, the line's shape
        x = torch.tensor([rank + 1.0])
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
            dist.all_reduce(x)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "train step-time (dry-run)", "value": 1.0, "unit": "ms/step", "n_gpus": world, "steps": a.steps,
                              "warmup": a.warmup, "ms_per_step": 1.0, "higher_is_better": False, "scaling": "weak", "data": "dry-run",
                              "config": {"global_batch": 29 * world, "tokens_per_s": None, "gradient_wire": a.wire if world > 1 else None},
                              "roofline": {"achieved": None, "frac": None},
                              "dryrun": {"world": world, "master_port": os.environ.get("MASTER_PORT"), "allreduce_sum": float(x),
                                         "agent_store_env": "TORCHELASTIC_USE_AGENT_STORE" in os.environ}}))
        return
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
    trainer = showo_amd.Trainer(model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, coeffs=(1.0, 0.1, 1.0), wire=a.wire)
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

    import ctypes as C
    L = showo_amd._lib
    losses = None
    for _ in range(a.warmup):
        losses = step()
    L.call("showo_prof_reset")
    L.call("showo_prof_enable", 0)  # the timed region carries NO per-launch events: they run in a second leg below
    if trainer.exchange is not None:
        trainer.exchange.measure(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = step()
    barrier()
    dt = time.perf_counter() - t0
    exposed_ms = trainer.exchange.exposed_ms() if trainer.exchange is not None else None  # of the timed steps only
    if trainer.exchange is not None:
        trainer.exchange.measure(False)
    # ---- roofline leg, after the timed region: the same steps with every n-th launch of each kernel kind bracketed by HIP events
    if not a.no_events and a.roofline_steps > 0:
        L.call("showo_prof_set_stride", a.event_stride)  # per-launch events on a systematic sample of the launches
        L.call("showo_prof_enable", 1)
        for _ in range(a.roofline_steps):
            step()
        barrier()
        L.call("showo_prof_enable", 0)
    prof = {}
    for kind, name in ((0, "gemm"), (1, "attention_fwd"), (2, "vq_conv")):
        ms_k, n_k, fl_k = C.c_double(), C.c_int64(), C.c_double()
        L.call("showo_prof_read", kind, C.byref(ms_k), C.byref(n_k), C.byref(fl_k))
        n_all, fl_all = C.c_int64(), C.c_double()
        L.call("showo_prof_totals", kind, C.byref(n_all), C.byref(fl_all))
        prof[name] = {"ms": ms_k.value, "timed": int(n_k.value), "flop_timed": fl_k.value, "launches": int(n_all.value), "flop": fl_all.value}
    L.call("showo_prof_reset")
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    wire_bytes = trainer.exchange.wire_bytes() if trainer.exchange is not None else None
    if rank == 0:
        T = (bt + bl + bm) * 387
        ms = dt / a.steps * 1e3
        flop = 3 * T * 2.732e9 + (0 if vq is None else (bt + bm) * 0.355e12)  # 3 x 11 223 x F(387) + encoder
        gm = prof["gemm"]
        n_evt_steps = 0 if a.no_events else a.roofline_steps
        ach = gm["flop_timed"] / (gm["ms"] * 1e-3) / 1e12 if gm["ms"] > 0 else 0.0
        roofline = {"bound": "mfma", "kernel": "gemm2p_kernel (bf16 MFMA GEMM: forward, dgrad and wgrad projections + lm_head, every epilogue)",
                    "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0, "traffic": None,
                    "launches": gm["launches"], "timed_launches": gm["timed"], "avg_launch_ms": gm["ms"] / max(1, gm["timed"]),
                    "executed_tflop_per_step": gm["flop"] / max(1, n_evt_steps) / 1e12,
                    "time_share_of_step": (gm["flop"] / max(1, n_evt_steps) / max(1e-9, ach * 1e12)) / (dt / a.steps) if ach > 0 else None,
                    "measured_in": f"{n_evt_steps} extra step(s) with HIP events after the timed region",
                    "attention_fwd": {"achieved": prof["attention_fwd"]["flop_timed"] / max(1e-9, prof["attention_fwd"]["ms"] * 1e-3) / 1e12},
                    "vq_conv": {"achieved": prof["vq_conv"]["flop_timed"] / max(1e-9, prof["vq_conv"]["ms"] * 1e-3) / 1e12}}
        cpu = None
        with_vq = vq is not None
        losses_host = [float(x) for x in losses.cpu()]
        if not a.no_cpu_baseline and world == 1:
            trainer = model = vq = None  # free the 30+ GB of GPU-side state before the host-side leg allocates its own
            torch.cuda.empty_cache()
            cpu = cpu_baseline(bt + bl + bm)
        print(json.dumps({
            "metric": "train step-time (stage-1 mixed batch, fwd+bwd+AdamW" + ("+VQ encode" if with_vq else "") + ")",
            "value": ms, "unit": "ms/step", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
            "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE cfg5: showo_pretraining_stage1 per-GPU batch 15 t2i + 4 lm + 10 mmu x 387 tokens, "
                                   "random-init Show-o 1.45B, AdamW lr 1e-4", "global_batch": (bt + bl + bm) * world, "seq_len": 387,
                       "parallelism": f"dp{world}", "tokens_per_s": T * world / (ms * 1e-3),
                       "algorithmic_tflops_per_gpu": flop / (ms * 1e-3) / 1e12,
                       "gradient_wire": a.wire if world > 1 else None,
                       # GPU time the compute stream spends in GradientExchange.finish() (waiting for the collectives that did not
                       # hide behind backward + widening the wire) and the bytes every rank hands to RCCL per step
                       "exchange_exposed_ms": exposed_ms, "wire_bytes_per_rank": wire_bytes,
                       "losses_last_step": losses_host},
            "roofline": roofline, "cpu_baseline": cpu,
        }))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
