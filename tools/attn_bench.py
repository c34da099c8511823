"""Micro-benchmark of the prefill / t2i attention launch at the shape the 18-step t2i loop runs it (BASELINE cfg2 with prefix reuse:
B = 16 CFG-doubled rows, 32 heads, 258 query rows <soi> + 256 image tokens + <eoi>, 387 keys; per-row visibility interval as
synthetic.t2i_inputs builds it: conditional rows see their 2..37 text tokens + the image block, unconditional rows the image block).
usage: python tools/attn_bench.py [--variants 0,1,2] [--reps 200] [--op 0|1] [--B 16 --Lq 258 --Lk 387]
Prints us per launch and TF/s (4 B nH Lq Lk 64 flops) per implementation (showo_attn_set_impl: 0 = by shape, 1 = gather form,
2 = LDS-tiled, 3 = LDS-tiled at 3 waves / SIMD), and the max |difference| of every implementation's output against the first."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import showo_amd  # noqa: E402

L = showo_amd._lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0")
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--op", type=int, default=0)
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--nH", type=int, default=32)
    ap.add_argument("--Lq", type=int, default=258)
    ap.add_argument("--Lk", type=int, default=387)
    ap.add_argument("--once", action="store_true", help="one launch per variant (for rocprofv3 --pmc passes)")
    a = ap.parse_args()
    B, nH, Lq, Lk = a.B, a.nH, a.Lq, a.Lk
    Lp = (Lk + 63) // 64 * 64
    torch.manual_seed(0)
    dt = torch.float16 if a.op else torch.bfloat16
    Q = (torch.randn(B, nH, Lq, 64, device="cuda") * 0.25).to(dt).view(torch.int16).contiguous()
    K = torch.randn(B, nH, Lk, 64, device="cuda").to(dt).view(torch.int16).contiguous()
    Vt = torch.zeros(B, nH, 64, Lp, device="cuda", dtype=dt)
    Vt[..., :Lk] = torch.randn(B, nH, 64, Lk, device="cuda").to(dt)
    Vt = Vt.view(torch.int16).contiguous()
    T = Lk - Lq  # text block (129 at cfg2)
    iv = torch.zeros(B, Lq, 4, dtype=torch.int32)
    for b in range(B):
        words = (2 + (b * 5) % 36) if b < B // 2 else 0
        iv[b, :, 0] = max(0, T - 1 - words - 1)  # task token + words sit right of the pad run
        iv[b, :, 1] = Lk
    iv = iv.cuda()
    O = torch.zeros(B, Lq, nH * 64, dtype=torch.int16, device="cuda")
    s = L.stream()
    fn = "showo_attn_fwd_op16" if a.op else "showo_attn_fwd"

    def launch():
        args = [L.ptr(Q), L.ptr(K), L.ptr(Vt), L.ptr(iv), None, None, L.ptr(O), B, nH, Lq, Lk, Lk, Lp, nH * 64]
        L.call(fn, *(args + ([1, s] if a.op else [s])))

    flops = 4.0 * B * nH * Lq * Lk * 64
    ref = None
    for v in [int(x) for x in a.variants.split(",")]:
        L.call("showo_attn_set_impl", v)
        O.zero_()
        launch()
        torch.cuda.synchronize()
        out = O.view(dt).float().clone()
        if ref is None:
            ref = out
        diff = float((out - ref).abs().max())
        if a.once:
            print(f"variant {v}: one launch; max |o - o_variant0| = {diff:.3e}")
            continue
        for _ in range(300):  # steady state (clocks, L2 / Infinity-Cache contents): the first ~100 launches of a process run 8-10 % slower
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(a.reps):
                launch()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / a.reps * 1e3)
        print(f"variant {v}: {best:7.2f} us per launch = {flops / best / 1e6:6.1f} TF/s; max |o - o_variant0| = {diff:.3e} (|o| max {float(ref.abs().max()):.2f})")
    L.call("showo_attn_set_impl", 0)


if __name__ == "__main__":
    main()
