"""Same seeded full-size t2i_generate with SHOWO_LAYER_OVERLAP=0 and =1 (two processes): the sampled token ids must be identical
(the two-stream schedule changes which kernels run concurrently, not what they compute).  usage: python tools/overlap_check.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1:
    import torch
    import showo_amd
    from showo_amd import synthetic
    torch.manual_seed(0)
    m = synthetic.random_init_showo(max_batch=16, max_seq=387, ln_jitter=True).eval()
    uni = synthetic.prompting(128)
    ic, iu, mask = synthetic.t2i_inputs(uni, 8, 256, m.mask_token_id)
    outs = []
    for graph in (0, 1):
        gen = torch.Generator(device="cuda").manual_seed(5)
        outs.append(m.t2i_generate(input_ids=ic.clone(), uncond_input_ids=iu, attention_mask=mask, temperature=1.0, timesteps=18,
                                   guidance_scale=5.0, generator=gen, config=showo_amd.gen_config(), use_graph=graph).cpu())
    assert torch.equal(outs[0], outs[1]), "eager vs hipGraph differ"
    torch.save(outs[0], sys.argv[1])
    sys.exit(0)

paths = []
for v in ("0", "1"):
    p = f"/tmp/overlap_{v}.pt"
    subprocess.check_call([sys.executable, __file__, p], env=dict(os.environ, SHOWO_LAYER_OVERLAP=v))
    paths.append(p)
import torch
a, b = torch.load(paths[0]), torch.load(paths[1])
print("tokens identical:", bool(torch.equal(a, b)), tuple(a.shape), "distinct ids", int(a.unique().numel()))
assert torch.equal(a, b)
