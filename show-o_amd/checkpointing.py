"""Checkpoint rotation / resume of a training run -- the reference's `save_checkpoint` (training/train.py:851-889) and resume
block (:429-443) as two functions with the same on-disk result:

    <output_dir>/checkpoint-<global_step>/unwrapped_model/{config.json, pytorch_model.bin}
    <output_dir>/checkpoint-<global_step>/metadata.json            {"global_step": N}
    <output_dir>/checkpoint-<global_step>/optimizer.bin            (extra: `Trainer.state_dict()`; the reference does not
                                                                    restore optimizer state on resume)

Oldest checkpoints are removed BEFORE saving so that at most `checkpoints_total_limit` exist afterwards; only the main process
writes (every rank holds the same replica in data-parallel training)."""
import json
import os
import shutil

import torch


def _steps(output_dir):
    if not os.path.isdir(output_dir):
        return []
    ds = [d for d in os.listdir(output_dir) if d.startswith("checkpoint")]
    return sorted(ds, key=lambda x: int(x.split("-")[1]))


def save_checkpoint(model, output_dir, global_step, checkpoints_total_limit=None, trainer=None, is_main_process=True):
    if not is_main_process:
        return None
    os.makedirs(output_dir, exist_ok=True)
    if checkpoints_total_limit is not None:
        existing = _steps(output_dir)
        if len(existing) >= checkpoints_total_limit:
            for d in existing[:len(existing) - checkpoints_total_limit + 1]:
                shutil.rmtree(os.path.join(output_dir, d))
    save_path = os.path.join(output_dir, f"checkpoint-{global_step}")
    model.save_pretrained(os.path.join(save_path, "unwrapped_model"), safe_serialization=False)
    with open(os.path.join(save_path, "metadata.json"), "w+") as f:
        json.dump({"global_step": global_step}, f)
    if trainer is not None:
        torch.save(trainer.state_dict(), os.path.join(save_path, "optimizer.bin"))
    return save_path


def latest_checkpoint(output_dir):
    """(path, global_step) of the highest-numbered checkpoint, or (None, 0)"""
    ds = _steps(output_dir)
    if not ds:
        return None, 0
    return os.path.join(output_dir, ds[-1]), int(ds[-1].split("-")[1])


def resume_from_checkpoint(model, output_dir, trainer=None):
    """loads the newest checkpoint into `model` (strict, like the reference) and, when present, the optimizer state into
    `trainer`; returns global_step (0 when there is nothing to resume)"""
    path, step = latest_checkpoint(output_dir)
    if path is None:
        return 0
    wdir = os.path.join(path, "unwrapped_model")
    if os.path.isfile(os.path.join(wdir, "pytorch_model.bin")):
        sd = torch.load(os.path.join(wdir, "pytorch_model.bin"), map_location="cpu", weights_only=True)
    else:
        from safetensors.torch import load_file
        sd = load_file(os.path.join(wdir, "pytorch_model.safetensors"))
    model.load_state_dict(sd, strict=True)
    if hasattr(model, "_weights_changed"):
        model._weights_changed = True
    opt = os.path.join(path, "optimizer.bin")
    if trainer is not None and os.path.isfile(opt):
        trainer.load_state_dict(torch.load(opt, map_location="cpu", weights_only=False))
    return step
