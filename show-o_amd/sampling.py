"""Mask schedules for the mask-predict sampler — host-side mirror of the reference's `models/sampling.py`
public names (`get_mask_chedule` [sic], `cosine_schedule`, ...; reference models/sampling.py:39-78).

These are tiny scalar functions evaluated once per denoise step on the HOST with torch CPU ops, exactly as
the reference evaluates them (`noise_schedule(torch.tensor(ratio))`, reference models/modeling_showo.py:157-158),
so the per-step constants handed to the HIP sampler are bit-identical to the reference's.
The per-token work of the sampler (softmax / multinomial / Gumbel top-k) lives in csrc/sampler.hip.
"""
import math
from functools import partial

import torch


def cosine_schedule(t):
    return torch.cos(t * (math.pi * 0.5))


def linear_schedule(t):
    return (1 - t).clamp(min=1e-6, max=1.0)


def pow_schedule(t, method):
    exponent = float(method.replace("pow", ""))
    return (1.0 - t ** exponent).clamp(min=1e-6, max=1.0)


def sigmoid_schedule(t, start=-3, end=3, tau=1.0, clip_min=1e-6):
    lo = torch.sigmoid(torch.tensor(start / tau))
    hi = torch.sigmoid(torch.tensor(end / tau))
    cur = torch.sigmoid((t * (end - start) + start) / tau)
    return torch.clip((hi - cur) / (hi - lo), clip_min, 1.0)


def get_mask_chedule(method, **schedule_kwargs):
    """Same (misspelled) name and return contract as the reference (models/sampling.py:68-78)."""
    if method == "cosine":
        return cosine_schedule
    if method == "linear":
        return linear_schedule
    if "pow" in method:
        return partial(pow_schedule, method=method)
    if method == "sigmoid":
        return partial(sigmoid_schedule, **schedule_kwargs)
    raise ValueError("Unknown schedule method: {}".format(method))


def t2i_step_constants(timesteps, num_vq_tokens, temperature=1.0, noise_schedule=cosine_schedule):
    """Per-step host constants of Showo.t2i_generate (reference models/modeling_showo.py:157-173):
    floor(N * schedule((k+1)/T)) evaluated in fp32 on the CPU and the compounding temperature."""
    mask_len, temps = [], []
    temp = temperature
    for step in range(timesteps):
        ratio = 1.0 * (step + 1) / timesteps
        mask_ratio = noise_schedule(torch.tensor(ratio))
        mask_len.append(float((num_vq_tokens * mask_ratio).floor()))
        temp = temp * (1.0 - ratio)
        temps.append(float(torch.tensor(temp, dtype=torch.float32)))
    return mask_len, temps
