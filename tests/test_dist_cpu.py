"""world_size-2 tests of the multi-process paths on CPU (gloo, 127.0.0.1): the replica aggregation of bench.py and the
gradient exchange of the data-parallel Trainer -- the SAME `GradientExchange` object `Trainer.step` drives (launch per bucket
in backward order, finish before the optimizer), with the wire staging done by torch ops instead of the HIP kernels because the
buckets live on the CPU here.  The GPU job runs the same code over RCCL (tests/test_train_gpu.py covers the HIP staging)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class TorchWireOps:
    """CPU stand-in of show-o_amd.training.HipWireOps (same three operations, same arithmetic)"""

    @staticmethod
    def pack(grad, wire, scale):
        wire.copy_((grad * scale).to(torch.bfloat16))

    @staticmethod
    def unpack(wire, grad):
        grad.copy_(wire.float())

    @staticmethod
    def scale(grad, s):
        grad.mul_(s)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import showo_amd
    # replica aggregation: slowest rank defines the time, units add up
    dt, units = bench.aggregate(1.0 + rank, 24, dist, "cpu")
    ok = dt == float(world) and units == 24 * world
    # gradient exchange: every rank holds different gradients, all end with the mean; buckets are launched in the order
    # Trainer.step uses (head first, embedding last) and completed by finish()
    g = torch.Generator().manual_seed(7)
    base = [torch.randn(n, generator=g) for n in (5, 4099, 64)]
    # bf16 wire (unit roundoff 2^-8): each rank's x_r / world is rounded once and the sum once more: |err| <= 2^-7 * mean_r |x_r|
    for wire, tol in (("fp32", 1e-6), ("bf16", 2.0 ** -7)):
        buckets = [b * (rank + 1) + rank for b in base]
        ex = showo_amd.training.GradientExchange(buckets, dist, None, wire=wire, ops=TorchWireOps)
        for b in (2, 1, 0):
            ex.launch(b)
        ex.finish()
        assert ex.works == []
        for got, b in zip(buckets, base):
            want = sum(b * (r + 1) + r for r in range(world)) / world
            mag = sum((b * (r + 1) + r).abs() for r in range(world)) / world
            ok = ok and bool(((got - want).abs() <= tol * mag + 1e-7).all())
        ok = ok and ex.wire_bytes() == sum(b.numel() for b in base) * (4 if wire == "fp32" else 2)
        # every rank ends with bit-identical buckets (the replicas must not drift apart)
        mine = torch.cat(buckets)
        both = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        ok = ok and all(torch.equal(both[0], t) for t in both)
        # measurement hooks on CPU buckets: nothing is timed (device events only), the accessors say so instead of failing
        ex.measure(True)
        ok = ok and ex.exposed_ms() is None and ex.exposed_ms_per_bucket() is None
    # Trainer.logging_means: the reference's four logging gathers (training/train.py:603-610) as ONE all-reduce of a 4-vector
    import types
    fake = types.SimpleNamespace(exchange=showo_amd.training.GradientExchange([torch.zeros(1)], dist, None, wire="fp32", ops=TorchWireOps))
    losses = torch.tensor([1.0, 2.0, 3.0]) * (rank + 1)
    means = showo_amd.Trainer.logging_means(fake, losses, torch.full((5,), 0.25 * (rank + 1)))
    want = torch.tensor([1.0, 2.0, 3.0, 0.25]) * (sum(range(1, world + 1)) / world)
    ok = ok and bool(torch.allclose(means, want))
    fake1 = types.SimpleNamespace(exchange=None)
    ok = ok and bool(torch.equal(showo_amd.Trainer.logging_means(fake1, losses)[:3], losses))
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo_aggregate_and_gradient_exchange():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_no_decay_rule_is_the_reference_rule():
    """training/train.py:211: only names containing these substrings skip weight decay -- with Phi's parameter names
    that means biases only (layernorm / embed_tokens weights DO decay, a quirk kept on purpose)"""
    sys.path.insert(0, ROOT)
    import showo_amd
    nd = showo_amd.training.NO_DECAY
    assert any(x in "showo.model.layers.0.mlp.fc1.bias" for x in nd)
    assert not any(x in "showo.model.layers.0.input_layernorm.weight" for x in nd)
    assert not any(x in "showo.model.embed_tokens.weight" for x in nd)


def test_bench_gpus_2_launch_plumbing_with_two_real_processes():
    """`python bench.py --gpus 2` end to end with REAL processes (no monkeypatching): bench.self_launch re-execs the
    command as 2 ranks under torch.distributed.run, the ranks rendezvous (gloo here, RCCL on the GPU box), aggregate MAX time / SUM
    units, and every rank starts bench_train.py as a child whose ranks rendezvous among themselves on MASTER_PORT + 101 without the
    launcher's agent store.  SHOWO_BENCH_DRYRUN=1 removes the GPU work only; launch code, environment handling and aggregation are the
    product's."""
    import json
    import subprocess
    env = dict(os.environ, SHOWO_BENCH_DRYRUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-1500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, r.stdout[-800:]  # only rank 0 prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dryrun"]["ranks"] == 2
    assert d["dryrun"]["units"] == 2 * 8 * 3                      # SUM of the units of both ranks
    assert abs(d["dryrun"]["max_dt"] - 0.02) < 1e-9               # MAX over ranks (rank 1 reported 0.02)
    ts = d["train_step"]
    assert ts is not None and "error" not in ts, ts
    assert ts["n_gpus"] == 2 and ts["global_batch"] == 58 and ts["gradient_wire"] == "bf16"
    child = ts["dryrun"]
    assert child["world"] == 2 and child["allreduce_sum"] == 3.0  # the two CHILD ranks found each other: 1 + 2
    assert int(child["master_port"]) == int(d["dryrun"]["master_port"]) + 101
    assert child["agent_store_env"] is False
    # the training line on its own, launched the same way
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_train.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-1500:])
    d2 = json.loads([l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l][0])
    assert d2["n_gpus"] == 2 and d2["dryrun"]["allreduce_sum"] == 3.0
