"""timing of the MAGVIT-v2 tokenizer paths: get_code of 25 images (one training step) and decode_code of 8 (one t2i batch)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import showo_amd
for prec in (1, 0):
    vq = showo_amd.MAGVITv2(max_batch=25, max_res=256, precision=prec).cuda().eval()
    x = torch.rand(25, 3, 256, 256, device="cuda") * 2 - 1
    ids = torch.randint(0, 8192, (8, 256), device="cuda")
    for name, f in (("get_code x25", lambda: vq.get_code(x)), ("decode_code x8", lambda: vq.decode_code(ids))):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): f()
        torch.cuda.synchronize()
        print(f"precision {prec} {name}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms")
    del vq
