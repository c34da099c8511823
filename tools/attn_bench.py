"""A/B timing of the two attention kernels at the bench shape (B=16 CFG rows, 32 heads, L=387, t2i masks)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
import showo_amd, showo_oracle as O, weights as Wt
L = showo_amd._lib
d = Wt.ShowoDims()
B, nH, Lq = 16, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 387
Lp = (Lq + 63) // 64 * 64
torch.manual_seed(0)
rows = []
for i in range(B):
    k = 5 + (i * 5) % 36
    rows.append([d.pad_id] * (129 - k) + [d.t2i_id] + [7] * (k - 1) + [d.soi_id] + [d.mask_token_id] * (Lq - 131) + [d.eoi_id])
mask = O.mask_t2i(torch.tensor(rows), d.pad_id, d.soi_id, d.eoi_id).cuda()
iv = torch.empty((B, Lq, 4), dtype=torch.int32, device="cuda"); flag = torch.zeros(4, dtype=torch.int32, device="cuda")
L.call("showo_mask_compress", L.ptr(mask), L.ptr(iv), L.ptr(flag), B, Lq, Lq, L.stream())
Q = (torch.randn(B, nH, Lq, 64, device="cuda") * 0.3).to(torch.bfloat16)
K = torch.randn(B, nH, Lq, 64, device="cuda").to(torch.bfloat16)
Vt = torch.zeros(B, nH, 64, Lp, device="cuda", dtype=torch.bfloat16); Vt[..., :Lq] = torch.randn(B, nH, 64, Lq, device="cuda").to(torch.bfloat16)
outs = {}
for impl in (1, 2, 3):
    L.call("showo_attn_set_impl", impl)
    out = torch.zeros(B * Lq, nH * 64, device="cuda", dtype=torch.bfloat16)
    f = lambda: L.call("showo_attn_fwd", L.ptr(Q), L.ptr(K), L.ptr(Vt), L.ptr(iv), L.ptr(flag), None, L.ptr(out), B, nH, Lq, Lq, Lq, Lp, nH * 64, L.stream())
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"impl {impl}: {us:8.1f} us  {4.0 * B * nH * Lq * Lq * 64 / us / 1e6:7.1f} TF (dense flops)")
    outs[impl] = out.float()
print("max |impl1 - impl2| =", float((outs[1] - outs[2]).abs().max()), "max |impl2 - impl3| =", float((outs[2] - outs[3]).abs().max()), " max|out| =", float(outs[1].abs().max()))
