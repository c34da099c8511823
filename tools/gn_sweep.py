"""Same-process sweep of the GEMM tile-group width gn (n-panels per XCD tile group) on the bench workload: images/s per setting."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import showo_amd
from showo_amd import synthetic
L = showo_amd._lib
torch.manual_seed(0)
B = 8
model = synthetic.random_init_showo(max_batch=2 * B, max_seq=387, ln_jitter=True).eval()
vq = showo_amd.MAGVITv2(max_batch=B, max_res=256).cuda().eval()
uni = synthetic.prompting(128)
ic, iu, mask = synthetic.t2i_inputs(uni, B, 256, model.mask_token_id)
cfg = showo_amd.gen_config()
gen = torch.Generator(device="cuda").manual_seed(1)


def step():
    toks = model.t2i_generate(input_ids=ic.clone(), uncond_input_ids=iu, attention_mask=mask, temperature=1.0, timesteps=18,
                              guidance_scale=5.0, generator=gen, config=cfg)
    return vq.decode_code(torch.clamp(toks, max=8191, min=0))


for _ in range(2):
    step()
torch.cuda.synchronize()
def run(label):
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    print(f"{label}: {3 * B / (time.perf_counter() - t0):.2f} images/s", flush=True)


for gn in [4, 8, 4]:
    L.call("showo_gemm_tune", gn, 0, None)
    run(f"gn={gn}")
L.call("showo_gemm_tune", 4, 0, None)
for impl in [0, 3, 2, 0, 3]:
    L.call("showo_attn_set_impl", impl)
    run(f"attn_impl={impl}")
L.call("showo_attn_set_impl", 0)
for flags in [0, 1, 0]:
    L.call("showo_gemm_tune", 4, flags, None)
    run(f"gemm flags={flags} (1 = no stagger)")
