"""Times the split-precision 3x3 convolution (showo_conv3x3_bf16x3) on the VQGAN's large launches.  The kernel choice is an
environment switch read once per process, so an A/B is two runs: SHOWO_CONV_3TAP=1 (default: conv3t_split_kernel where it applies)
and SHOWO_CONV_3TAP=0 (conv2p_split_kernel).   usage: python tools/conv_ab.py [--batch 8]   (GPU)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    import showo_amd
    L = showo_amd._lib
    lib = L.load()
    B = a.batch
    shapes = [(256, 256, 128, 128, 0), (128, 128, 128, 128, 0), (128, 128, 256, 128, 0), (64, 64, 256, 256, 0), (128, 128, 128, 128, 1),
              (64, 64, 256, 256, 1), (32, 32, 512, 256, 0), (256, 256, 128, 128, 2)]
    print(f"# SHOWO_CONV_3TAP={os.environ.get('SHOWO_CONV_3TAP', '1')}  batch {B}:  H W Cin Cout mode | us | algorithmic TF/s | launches on conv3t")
    tot = 0.0
    for (H, W, Cin, Cout, mode) in shapes:
        Ho, Wo = (H, W) if mode == 0 else (2 * H, 2 * W) if mode == 1 else (H // 2, W // 2)
        x = torch.randn(B, H, W, Cin, device="cuda")
        w = torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.03
        bias = torch.randn(Cout, device="cuda")
        def split(t):
            hi = torch.empty(t.shape, dtype=torch.int16, device="cuda"); lo = torch.empty_like(hi)
            L.call("showo_split_f32_bf16", L.ptr(t), L.ptr(hi), L.ptr(lo), t.numel(), L.stream())
            return hi, lo
        xh, xl = split(x.contiguous()); wh, wl = split(w.contiguous())
        out = torch.empty(B, Ho * Wo, Cout, device="cuda")
        n0 = lib.showo_conv3t_launches()
        def run():
            L.call("showo_conv3x3_bf16x3", L.ptr(xh), L.ptr(xl), L.ptr(wh), L.ptr(wl), L.ptr(bias), None, L.ptr(out), B, H, W, Cin, Cout, mode, L.stream())
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
        fl = 2.0 * B * Ho * Wo * Cout * 9 * Cin
        tot += us
        print(f"{H:4d} {W:4d} {Cin:4d} {Cout:4d} {mode} | {us:9.1f} | {fl / us / 1e6:7.1f} | {lib.showo_conv3t_launches() - n0}", flush=True)
    print(f"# sum {tot:.1f} us")


if __name__ == "__main__":
    main()
