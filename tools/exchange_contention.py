"""what does leaving CUs to RCCL cost the compute, and what does the exchange expose, on ONE GPU?

Runs the stage-1 training step (bench_train.py's batch: 15 t2i + 4 lm + 10 mmu x 387 tokens, VQ encode included) with the gradient
exchange forced on in a one-rank RCCL group (26 all-reduces of ~100 MB bf16 per step through torch.distributed; with one rank RCCL's
kernels move no data over xGMI, so this measures launch / stream / CU-occupancy interaction, not link time) and the compute stream
masked to 256 - r CUs (Trainer(reserve_cus=r), showo_stream_create_cu_mask) for r in {0, 8, 16}; baseline = no exchange, no mask.
Prints one table: step ms, GPU ms the compute stream spent in finish() (total and the five most exposed buckets).

    python tools/exchange_contention.py > profiles/r5_exchange_contention.txt"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import showo_amd
    import bench_train
    from showo_amd import synthetic
    from showo_amd.training_utils import build_training_batch
    bt, bl, bm = 15, 4, 10
    torch.manual_seed(0)
    model = synthetic.random_init_showo(max_batch=bt + bl + bm, max_seq=387).train()
    vq = showo_amd.MAGVITv2(max_batch=bt + bm, max_res=256).cuda().eval()
    uni = synthetic.prompting(max_text_len=128, cond_dropout_prob=0.1)
    off = len(uni.text_tokenizer)
    cfg = type("Cfg", (), {"training": type("Training", (dict,), {"__getattr__": dict.__getitem__})(min_masking_rate=0.0)})
    rs = np.random.RandomState(4)
    torch.manual_seed(4)
    random.seed(4)
    tt, tl, tm = bench_train.synthetic_texts(rs, bt, bl, bm)
    images = torch.rand(bt + bm, 3, 256, 256, device="cuda") * 2 - 1

    def make(force, r):
        return showo_amd.Trainer(model, lr=1e-4, coeffs=(1.0, 0.1, 1.0), wire="bf16", force_exchange=force, reserve_cus=r)

    def run(tr, steps=4, warm=2):
        def step():
            tokens = vq.get_code(images) + off
            ids, labels, mask, _, (b1, b2, b3) = build_training_batch(uni, cfg, model.mask_token_id, showo_amd.cosine_schedule, tokens[:bt],
                                                                      list(tt), list(tl), tokens[bt:], list(tm))
            return tr.step(ids, mask, labels, b1, b2, b3, 128)
        for _ in range(warm):
            step()
        if tr.exchange is not None:
            tr.exchange.measure(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        exp = tr.exchange.exposed_ms() if tr.exchange is not None else None
        per = tr.exchange.exposed_ms_per_bucket() if tr.exchange is not None else None
        return ms, exp, per

    print("# stage-1 training step on ONE MI355X, gradient exchange forced on in a one-rank RCCL group, bf16 wire (2.9 GB per step in 26 buckets)")
    print("# reserve = CUs kept out of the compute stream's kernels (showo_stream_create_cu_mask); exposed = GPU ms of the compute stream inside finish()")
    ms, _, _ = run(make(False, 0))
    print(f"no exchange, no mask        : {ms:7.2f} ms/step")
    for rep in range(2):
        for r in (0, 8, 16):  # multiples of 8: the same number of CUs from every XCD (round 5)
            tr_ = make(True, r)
            ms, exp, per = run(tr_)
            tr_.close()
            top = sorted(per.items(), key=lambda kv: -kv[1])[:5] if per else []
            print(f"exchange on, reserve {r:3d} CUs: {ms:7.2f} ms/step   exposed {exp:6.2f} ms   most exposed buckets (index: ms) "
                  + " ".join(f"{b}:{v:.2f}" for b, v in top) + f"   [pass {rep}]", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
