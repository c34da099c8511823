"""Prefill (631 embeddings -> first token) with the two GEMM kernel families forced, same process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import showo_amd
from showo_amd import synthetic
L = showo_amd._lib
torch.manual_seed(0)
m = synthetic.random_init_showo(max_batch=1, max_seq=768, w_clip_vit=True).eval()
for Lp in (631, 387, 256):
    emb = torch.randn(1, Lp, 2048, device="cuda") * 0.02
    for impl in (0, 5, 1, 0, 5):
        L.call("showo_gemm_set_impl", impl)
        m.mmu_generate(input_embeddings=emb, attention_mask=None, max_new_tokens=1, top_k=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            m.mmu_generate(input_embeddings=emb, attention_mask=None, max_new_tokens=1, top_k=1)
        torch.cuda.synchronize()
        print(f"L={Lp} gemm impl={impl} (0 = by shape, 5 = phase-split 256-wide, 1 = 128x128): prefill {1e3 * (time.perf_counter() - t0) / 3:.2f} ms", flush=True)
L.call("showo_gemm_set_impl", 0)
