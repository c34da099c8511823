"""world_size-2 tests of the multi-process paths on CPU (gloo, 127.0.0.1): the replica aggregation of bench.py and the
gradient-bucket averaging of the data-parallel Trainer.  The GPU job runs the same code over RCCL."""
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


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import showo_amd
    # replica aggregation: slowest rank defines the time, units add up
    dt, units = bench.aggregate(1.0 + rank, 24, dist, "cpu")
    # gradient buckets: every rank holds different gradients, all end with the mean (fixed bucket order)
    buckets = [torch.full((5,), float(rank + 1)), torch.arange(7, dtype=torch.float32) * (rank + 1)]
    showo_amd.training.average_buckets(buckets, async_op=True)
    ok = dt == float(world) and units == 24 * world
    ok = ok and torch.allclose(buckets[0], torch.full((5,), (1 + world) / 2.0))
    ok = ok and torch.allclose(buckets[1], torch.arange(7, dtype=torch.float32) * (1 + world) / 2.0)
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo_aggregate_and_bucket_average():
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
