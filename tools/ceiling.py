"""Measured ceilings of the box, printed next to the spec figures ("re-measure with a hipBLASLt GEMM and a copy
kernel and print both").  One process, interleaved rounds:

  * torch.matmul bf16 (hipBLASLt / rocBLAS, comparison only -- the product never calls it) and our gemm2p kernel on the bench
    shapes, random-normal and zero-filled operands (zero operands show the DVFS give-back: the chip clocks to its power budget);
  * the float4 copy kernel (showo_copy_b128) and torch's device copy on a 2 GiB buffer (HBM) and a 64 MiB buffer (Infinity Cache);
  * rocm-smi clock / power samples taken while each arm runs.

usage: python tools/ceiling.py [out.json]
"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import showo_amd  # noqa: E402

L = showo_amd._lib


class Smi(threading.Thread):
    """samples `rocm-smi --showclocks --showpower --json` every 0.4 s while an arm runs"""

    def __init__(self):
        super().__init__(daemon=True)
        self.samples, self.stop_flag = [], False

    def run(self):
        while not self.stop_flag:
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
                d = json.loads(out)
                card = d.get("card0", {})
                rec = {}
                for k, v in card.items():
                    kl = k.lower()
                    if "sclk" in kl and "clock" in kl:
                        rec["sclk"] = v
                    if "power" in kl and ("socket" in kl or "average" in kl or "current" in kl):
                        rec["power_w"] = v
                if rec:
                    self.samples.append(rec)
            except Exception:  # rocm-smi missing or unparsable: the timing still stands
                pass
            time.sleep(0.4)


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def ours(A, W, out, M, N, K):
    L.call("showo_gemm_bf16", L.ptr(A), K, L.ptr(W), K, None, 0, L.ptr(out), N, None, 0, M, N, K, L.EPI_BF16, L.stream())


def main():
    dev = torch.device("cuda:0")
    res = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "gemm": [], "copy": []}
    try:
        torch.backends.cuda.preferred_blas_library("hipblaslt")
        res["blas"] = "hipblaslt (torch.backends.cuda.preferred_blas_library)"
    except Exception as ex:  # older torch: whatever the default is
        res["blas"] = f"default ({ex})"
    shapes = [(4128, 6144, 2048), (4128, 2048, 2048), (4128, 8192, 2048), (4128, 2048, 8192), (4128, 14336, 2048), (4128, 2048, 10240),
              (6192, 6144, 2048), (6192, 2048, 2048), (6192, 8192, 2048), (6192, 2048, 8192), (4096, 4096, 4096), (8192, 8192, 8192)]
    g = torch.Generator(device=dev).manual_seed(0)
    for (M, N, K) in shapes:
        flops = 2.0 * M * N * K
        iters = 20 if flops < 3e11 else 6
        for fill in ("random", "zero"):
            if fill == "random":
                A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
                W = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
            else:
                A = torch.zeros(M, K, device=dev, dtype=torch.bfloat16)
                W = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            Wt = W.t()
            rec = {"M": M, "N": N, "K": K, "fill": fill}
            smi = Smi()
            smi.start()
            best = {"blas": 1e9, "ours": 1e9}
            for _ in range(3):  # interleaved rounds, minimum per arm
                best["blas"] = min(best["blas"], timed(lambda: torch.matmul(A, Wt, out=out), iters))
                best["ours"] = min(best["ours"], timed(lambda: ours(A, W, out, M, N, K), iters))
            smi.stop_flag = True
            smi.join()
            rec["blas_tflops"] = flops / best["blas"] / 1e12
            rec["ours_tflops"] = flops / best["ours"] / 1e12
            rec["smi"] = smi.samples[-3:]
            res["gemm"].append(rec)
            print(f"gemm M={M} N={N} K={K} {fill:6s}: blas {rec['blas_tflops']:7.1f} TF/s   gemm2p {rec['ours_tflops']:7.1f} TF/s   smi {rec['smi'][-1:] }", flush=True)
    for nbytes in (2 << 30, 64 << 20):
        src = torch.empty(nbytes // 4, device=dev, dtype=torch.float32).normal_()
        dst = torch.empty_like(src)
        smi = Smi()
        smi.start()
        t_k = min(timed(lambda: L.call("showo_copy_b128", L.ptr(src), L.ptr(dst), nbytes, L.stream()), 20) for _ in range(3))
        t_t = min(timed(lambda: dst.copy_(src), 20) for _ in range(3))
        smi.stop_flag = True
        smi.join()
        assert torch.equal(src, dst)
        rec = {"bytes": nbytes, "copy_b128_TBps": 2 * nbytes / t_k / 1e12, "torch_copy_TBps": 2 * nbytes / t_t / 1e12, "smi": smi.samples[-3:]}
        res["copy"].append(rec)
        print(f"copy {nbytes >> 20} MiB: showo_copy_b128 {rec['copy_b128_TBps']:.2f} TB/s (read + write)   torch {rec['torch_copy_TBps']:.2f} TB/s", flush=True)
    res["measured_peak"] = {
        "bf16_tflops_blas_random": max(r["blas_tflops"] for r in res["gemm"] if r["fill"] == "random"),
        "bf16_tflops_blas_zero": max(r["blas_tflops"] for r in res["gemm"] if r["fill"] == "zero"),
        "bf16_tflops_gemm2p_random": max(r["ours_tflops"] for r in res["gemm"] if r["fill"] == "random"),
        "bf16_tflops_gemm2p_zero": max(r["ours_tflops"] for r in res["gemm"] if r["fill"] == "zero"),
        "hbm_TBps_copy": max(r["copy_b128_TBps"] for r in res["copy"] if r["bytes"] >= (1 << 30)),
    }
    print(json.dumps(res["measured_peak"]))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
