"""bench_configs.py — the two BASELINE.json configs that are not the headline line, as bench workloads with their own JSON line
(`python bench.py --workload t2i512 | mmu`; same flags, same one-line contract, N = 1 only):

  t2i512 (configs[2]): configs/showo_demo_512x512.yaml, t2i 512x512, batch 4, inpainting of the centred 16x16 token block
      (inference_t2i.py:100-113), CFG 5.0 -> forward on [8,1155], 18 mask-predict steps, including MAGVITv2.get_code of the
      input image and decode_code of the result.  MFMA-bound like the headline; roofline kernel = gemm2p_kernel.
  mmu (configs[3]): inference_mmu.py w_clip_vit, 512x512: per image CLIP ViT-L/14-336 tower + mm_projector + splice, prefill of
      the 631 prompt embeddings, 100 new tokens (top_k = 1) against the KV cache; 4 images = 4 independent batch-1 decodes
      (modeling_showo.py:204,229).  HBM-bound: every decoded token streams the 2.66 GB of bf16 weights once
      (24 x 100.7 MB + 240 MB lm_head), so roofline.achieved = 2.66 GB x tokens / decode time.

  vq: the HBM-bound kernels of the MAGVIT-v2 path as bandwidth lines (BASELINE.json north_star: "rocprof showing achieved HBM GB/s on
      the VQ/argmin path"): LFQ pack (sign test + bit pack, models/modeling_magvitv2.py:201-206) / unpack (:208-221) at >= 64 MB and
      the GroupNorm(32) + swish passes (models/common_modules.py:16-24) at the decoder's largest activation ([8, 256x256, 128] fp32).
      achieved = algorithmic bytes (each tensor once) / launch time from HIP events on the launch stream; peak 8 TB/s.

Synthetic data, random-init weights of the true architectures (no checkpoints offline).  Nothing here touches oracle/."""
import argparse
import ctypes as C
import json
import sys
import time

import numpy as np
import torch


def _events(L, enable, stride=5):
    L.call("showo_prof_reset")
    L.call("showo_prof_set_stride", stride)
    L.call("showo_prof_enable", 1 if enable else 0)


def _read(L, kind):
    ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
    L.call("showo_prof_read", kind, C.byref(ms), C.byref(n), C.byref(fl))
    return ms.value, int(n.value), fl.value


def t2i512(a):
    import showo_amd
    from showo_amd import synthetic
    from showo_amd.prompting_utils import intervals_predict_next
    L = showo_amd._lib
    B, N = 4, 1024
    Lseq = 129 + 1 + N + 1
    torch.manual_seed(0)
    model = synthetic.random_init_showo(max_batch=2 * B, max_seq=Lseq, ln_jitter=True, num_vq_tokens=N).eval()
    if a.precision:
        model.set_precision(a.precision)
    vq = showo_amd.MAGVITv2(max_batch=B, max_res=512).cuda().eval()
    uni = synthetic.prompting(max_text_len=128)
    sp = uni.sptids_dict
    off = len(uni.text_tokenizer)
    x = (torch.rand(1, 3, 512, 512, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)) * 2 - 1).expand(B, -1, -1, -1).contiguous()
    grid = torch.zeros(32, 32, dtype=torch.bool)
    grid[8:24, 8:24] = True  # the centred 16x16 block of the 32x32 token grid is generated, the rest is kept
    hole = grid.reshape(-1).cuda()
    rs = np.random.RandomState(0)
    prompts = [synthetic.random_text(rs, 3 + 7 * i) for i in range(B)]
    cfg = showo_amd.gen_config(num_vq_tokens=N)
    gen = torch.Generator(device="cuda").manual_seed(1)

    def step():
        codes = vq.get_code(x)
        img = torch.where(hole[None], torch.full_like(codes, model.mask_token_id), codes + off)
        ic, _ = uni((prompts, img), 't2i_gen')
        iu, _ = uni(([''] * B, img), 't2i_gen')
        mask = intervals_predict_next(torch.cat([ic, iu]), pad_id=int(sp['<|pad|>']), soi_id=int(sp['<|soi|>']), eoi_id=int(sp['<|eoi|>']),
                                      rm_pad_in_image=True)
        toks = model.t2i_generate(input_ids=ic.contiguous(), uncond_input_ids=iu.contiguous(), attention_mask=mask, timesteps=18,
                                  guidance_scale=5.0, generator=gen, config=cfg, use_graph=a.graph)
        return codes, toks, vq.decode_code(toks)

    for _ in range(a.warmup):
        codes, toks, img = step()
    torch.cuda.synchronize()
    assert tuple(img.shape) == (B, 3, 512, 512) and torch.isfinite(img).all()
    assert torch.equal(toks[:, ~hole], codes[:, ~hole])  # known tokens come back untouched
    _events(L, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _events(L, True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    dt_evt = time.perf_counter() - t1
    L.call("showo_prof_enable", 0)
    ms_g, n_g, fl_g = _read(L, 0)
    ms_a, n_a, fl_a = _read(L, 1)
    ms_c, n_c, fl_c = _read(L, 2)
    ach = fl_g / (ms_g * 1e-3) / 1e12 if ms_g > 0 else 0.0
    value = B * a.steps / dt
    return {"metric": "t2i images/sec @512x512 (18 denoise steps, inpainting, incl. get_code + decode_code)", "value": value, "unit": "images/s",
            "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE cfg3: configs/showo_demo_512x512.yaml t2i 512x512 batch 4, centred 16x16-token inpainting mask, CFG 5.0 "
                                   "(forward on [8,1155]), 18 steps, MAGVITv2.get_code + decode_code; random-init Show-o 1.45B + MAGVIT-v2",
                       "global_batch": B, "seq_len": Lseq, "parallelism": "replicas x1", "algorithmic_tflop_per_image": 122.5,
                       "end_to_end_algorithmic_tflops": value * 122.5},
            "roofline": {"bound": "mfma", "kernel": "gemm2p_kernel", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0,
                         "traffic": None, "timed_launches": n_g, "avg_launch_ms": ms_g / max(1, n_g),
                         "measured_in": f"1 extra eager step with HIP events ({dt_evt * 1e3:.0f} ms)",
                         "attention": {"achieved": fl_a / max(1e-9, ms_a * 1e-3) / 1e12}, "vq_conv": {"achieved": fl_c / max(1e-9, ms_c * 1e-3) / 1e12}},
            "cpu_baseline": None}


def mmu(a):
    import showo_amd
    from showo_amd import synthetic
    from showo_amd.clip_encoder import CLIP_VIT_L_14_336, vision_state_spec
    from showo_amd.prompting_utils import create_attention_mask_for_mmu_vit
    torch.manual_seed(0)
    model = synthetic.random_init_showo(max_batch=1, max_seq=768, w_clip_vit=True).eval()
    if a.precision:
        model.set_precision(a.precision)  # (2: the batched entry point then serves the 4 sequences one by one)
    g = torch.Generator().manual_seed(22)
    sd = {}
    for k, shape in vision_state_spec(CLIP_VIT_L_14_336).items():  # random-init weights of the true ViT-L/14-336 architecture
        if "layer_norm" in k or "layernorm" in k or "layrnorm" in k:
            sd[k] = torch.ones(shape) if k.endswith("weight") else torch.zeros(shape)
        else:
            sd[k] = torch.randn(shape, generator=g) * 0.02
    tower = showo_amd.CLIPVisionTower("synthetic", config=CLIP_VIT_L_14_336, state_dict=sd, max_batch=1).cuda()
    emb_tab = model.showo.model.embed_tokens.weight
    Lp, NEW = 1 + 28 + 1 + 576 + 1 + 24, 100
    n_img = 4 * max(1, a.steps)
    t_clip, t_first, t_dec = [], [], []
    for i in range(a.warmup + n_img):
        gg = torch.Generator(device="cuda").manual_seed(3 + i)
        pixels = torch.randn(1, 3, 336, 336, device="cuda", generator=gg)  # CLIPImageProcessor output (host side, not timed)
        ids = torch.randint(0, 50256, (1, Lp - 576), device="cuda", generator=gg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            img_emb = model.mm_projector(tower(pixels))
            txt = emb_tab[ids]
            emb = torch.cat([txt[:, :30], img_emb, txt[:, 30:]], dim=1)
        am = create_attention_mask_for_mmu_vit(emb, system_prompt_len=28)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        first = model.mmu_generate(input_embeddings=emb, attention_mask=am[0], max_new_tokens=1, top_k=1)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        toks = model.mmu_generate(input_embeddings=emb, attention_mask=am[0], max_new_tokens=NEW, top_k=1)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        assert len(toks) == NEW and int(toks[0]) == int(first[0])
        if i >= a.warmup:
            t_clip.append(t1 - t0), t_first.append(t2 - t1), t_dec.append(t3 - t2)
    tc, tf, td = float(np.mean(t_clip)), float(np.mean(t_first)), float(np.mean(t_dec))
    # decode time of the NEW - 1 cached steps = whole call minus the prefill + first token (measured on the same prompt)
    t_tok = max(1e-9, (td - tf) / (NEW - 1))
    bytes_per_token = 2.0 * (24 * (4 * 2048 * 2048 + 2 * 2048 * 8192) + 58498 * 2048)  # bf16 weights streamed once per token
    ach = bytes_per_token / t_tok / 1e9
    # ---- the config as BASELINE.json writes it: "batch=4 images".  Four prompts (own CLIP features, own question) served TOGETHER by
    # Showo.mmu_generate_batch (csrc/decode_batch.hip: 4 KV caches, one weight stream per token step); each sequence must reproduce its
    # own batch-1 tokens.  Timed: the whole call (4 prefills + first tokens + 99 batched steps) and, separately, the 4 prefills alone.
    NB = 4
    embs, masks = [], []
    for b in range(NB):
        gg = torch.Generator(device="cuda").manual_seed(100 + b)
        pixels = torch.randn(1, 3, 336, 336, device="cuda", generator=gg)
        ids = torch.randint(0, 50256, (1, Lp - 576), device="cuda", generator=gg)
        with torch.no_grad():
            img_emb = model.mm_projector(tower(pixels))
            txt = emb_tab[ids]
            embs.append(torch.cat([txt[:, :30], img_emb, txt[:, 30:]], dim=1).contiguous())
        masks.append(create_attention_mask_for_mmu_vit(embs[-1], system_prompt_len=28)[0])
    single = [[int(t) for t in model.mmu_generate(input_embeddings=embs[b], attention_mask=masks[b], max_new_tokens=NEW, top_k=1)] for b in range(NB)]
    tb_first, tb_all = [], []
    for i in range(a.warmup + max(1, a.steps)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.mmu_generate_batch(input_embeddings=embs, attention_mask=masks, max_new_tokens=1, top_k=1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        got = model.mmu_generate_batch(input_embeddings=embs, attention_mask=masks, max_new_tokens=NEW, top_k=1)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        assert [[int(t) for t in r] for r in got] == single, "batched decode must reproduce every sequence's batch-1 tokens"
        if i >= a.warmup:
            tb_first.append(t1 - t0), tb_all.append(t2 - t1)
    tbf, tba = float(np.mean(tb_first)), float(np.mean(tb_all))
    t_stepB = max(1e-9, (tba - tbf) / (NEW - 1))
    achB = bytes_per_token / t_stepB / 1e9  # the weights are streamed once per STEP of 4 tokens
    batch4 = {"sequences": NB, "new_tokens_each": NEW, "aggregate_tokens_per_s": NB * NEW / tba, "ms_whole_call": tba * 1e3,
              "ms_four_prefills_and_first_tokens": tbf * 1e3, "ms_per_step_of_4_tokens": t_stepB * 1e3,
              "decode_only_tokens_per_s": NB / t_stepB, "hbm_GBps": achB, "frac_of_8TBps": achB / 8000.0,
              "tokens_equal_batch1_runs": True, "speedup_vs_4_batch1_calls": (NB * td) / tba}
    # the copy ceiling of THIS box in THIS run (no stale file)
    L = showo_amd._lib
    src, dst = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"), torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    L.call("showo_copy_b128", L.ptr(src), L.ptr(dst), src.numel(), L.stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        L.call("showo_copy_b128", L.ptr(src), L.ptr(dst), src.numel(), L.stream())
    e1.record()
    torch.cuda.synchronize()
    copy_peak = 2 * src.numel() * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    return {"metric": "mmu AR decode tokens/sec (w_clip_vit, 631-embedding prompts, 100 new tokens each, batch 4 = BASELINE cfg4; batch 1 in config.batch1)",
            "value": NB * NEW / tba, "unit": "tokens/s",
            "n_gpus": 1, "steps": max(1, a.steps), "warmup": a.warmup, "ms_per_step": tba * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE cfg4: inference_mmu.py w_clip_vit 512x512, per image CLIP ViT-L/14-336 + mm_projector + splice, prefill of "
                                   "631 embeddings, 100 new tokens top_k=1 (KV cache, device-side loop, hipGraph per token step); value = the 4 images of "
                                   "the config decoded TOGETHER (mmu_generate_batch), batch1 = one image at a time (the reference's mmu_generate semantics)",
                       "global_batch": NB, "seq_len": Lp + NEW, "parallelism": "replicas x1", "batch4": batch4,
                       "batch1": {"tokens_per_s": NEW / td, "ms_per_image": td * 1e3, "hbm_GBps": ach, "frac_of_8TBps": ach / 8000.0},
                       "clip_projector_splice_ms": tc * 1e3, "prefill_to_first_token_ms": tf * 1e3, "time_to_first_token_ms": (tc + tf) * 1e3,
                       "ms_per_decoded_token": t_tok * 1e3},
            # the roofline describes the batch the `value` is for (the batch-1 fraction was attached to the batch-4 value):
            # one decode STEP of 4 tokens streams the same 2.66 GB of weights once; the batch-1 step is next to it
            "roofline": {"bound": "hbm", "kernel": "batch-4 decode step = 24 x (ln_gemvB<4> + attn_decode_coB<4> + out_dense_y2B<4>) + lm_head + per-sequence arg-max",
                         "achieved": achB, "peak": 8000.0, "unit": "GB/s", "frac": achB / 8000.0, "traffic": None,
                         "algorithmic_bytes_per_step_of_4_tokens": bytes_per_token, "measured_peak_copy_GBps": copy_peak,
                         "whole_call_incl_prefills_GBps": bytes_per_token * NEW / tba / 1e9,
                         "batch1_step": {"kernel": "24 x (ln_gemv2 + attn_decode_co + out_gemv2) + lm_head GEMV + arg-max", "achieved": ach, "frac": ach / 8000.0,
                                         "algorithmic_bytes_per_token": bytes_per_token}},
            "metric_history": "rounds 1-4 quoted batch-1 tokens/s under this workload (now config.batch1.tokens_per_s); since round 5 `value` is the aggregate of the "
                              "4 sequences of BASELINE cfg4 decoded together, whole call incl. the 4 prefills (ADVICE r5): compare like with like",
            "cpu_baseline": None}


def vq(a):
    """HBM GB/s of the VQ path's bandwidth-bound kernels at sizes where bandwidth, not launch latency, decides"""
    import showo_amd
    L = showo_amd._lib
    s = L.stream
    torch.manual_seed(0)
    reps = max(3, a.steps)

    def timed(fn):
        for _ in range(max(1, a.warmup)):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)  # torch's current stream IS the launch stream
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    rows = {}
    # ---- LFQ: 16 x 131 072 tokens x 13 channels fp32 = 109 MB in, 16.8 MB of int64 ids out (60 B per token)
    B, Cz, hw = 16, 13, 131072
    z = torch.randn(B, Cz, hw, device="cuda")
    ids = torch.empty(B, hw, dtype=torch.int64, device="cuda")
    t = timed(lambda: L.call("showo_lfq_pack_nchw", L.ptr(z), L.ptr(ids), B, Cz, hw, s()))
    by = B * hw * (Cz * 4 + 8)
    rows["lfq_pack_nchw"] = {"bytes": by, "us": t * 1e6, "GBps": by / t / 1e9}
    zq = torch.empty_like(z)
    t = timed(lambda: L.call("showo_lfq_unpack_nchw", L.ptr(ids), L.ptr(zq), B, Cz, hw, s()))
    rows["lfq_unpack_nchw"] = {"bytes": by, "us": t * 1e6, "GBps": by / t / 1e9}
    assert torch.equal(zq > 0, z > 0)  # pack -> unpack round trip keeps every sign (size-independent property)
    zl = torch.randn(B, hw, 16, device="cuda")
    t = timed(lambda: L.call("showo_lfq_pack_nhwc", L.ptr(zl), L.ptr(ids), B, Cz, hw, 16, s()))
    by2 = B * hw * (16 * 4 + 8)
    rows["lfq_pack_nhwc"] = {"bytes": by2, "us": t * 1e6, "GBps": by2 / t / 1e9}
    del z, zq, zl, ids
    # ---- GroupNorm(32, eps 1e-6) + swish on the decoder's largest activation: [8, 65 536, 128] fp32 NHWC = 268 MB
    Bn, HW, Cc = 8, 65536, 128
    x = torch.randn(Bn, HW, Cc, device="cuda")
    nd = L.load().showo_gn_stats_doubles(Bn, HW)
    stats = torch.empty(nd, dtype=torch.float64, device="cuda")
    gam, bet = torch.randn(Cc, device="cuda"), torch.randn(Cc, device="cuda")
    y = torch.empty(Bn, HW, Cc, dtype=torch.int16, device="cuda")
    ylo = torch.empty_like(y)
    t = timed(lambda: L.call("showo_gn_stats", L.ptr(x), L.ptr(stats), Bn, HW, Cc, s()))
    rows["gn_stats"] = {"bytes": x.numel() * 4, "us": t * 1e6, "GBps": x.numel() * 4 / t / 1e9}
    t = timed(lambda: L.call("showo_gn_apply", L.ptr(x), L.ptr(stats), L.ptr(gam), L.ptr(bet), L.ptr(y), None, Bn, HW, Cc, 1e-6, 1, s()))
    rows["gn_apply_swish_bf16"] = {"bytes": x.numel() * 6, "us": t * 1e6, "GBps": x.numel() * 6 / t / 1e9}
    t = timed(lambda: L.call("showo_gn_apply", L.ptr(x), L.ptr(stats), L.ptr(gam), L.ptr(bet), L.ptr(y), L.ptr(ylo), Bn, HW, Cc, 1e-6, 1, s()))
    rows["gn_apply_swish_split"] = {"bytes": x.numel() * 8, "us": t * 1e6, "GBps": x.numel() * 8 / t / 1e9}
    # the copy ceiling of this box, same process
    src, dst = torch.empty(1 << 28, dtype=torch.uint8, device="cuda"), torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    t = timed(lambda: L.call("showo_copy_b128", L.ptr(src), L.ptr(dst), src.numel(), s()))
    copy = 2 * src.numel() / t / 1e9
    dom = rows["gn_apply_swish_split"]  # the pass the default (split-precision) MAGVIT-v2 runs 42 times per image
    tot_b = sum(r["bytes"] for r in rows.values())
    tot_t = sum(r["us"] for r in rows.values()) * 1e-6
    return {"metric": "MAGVIT-v2 VQ path, HBM-bound kernels: achieved GB/s (LFQ pack / unpack, GroupNorm + swish)", "value": tot_b / tot_t / 1e9,
            "unit": "GB/s", "n_gpus": 1, "steps": reps, "warmup": a.warmup, "ms_per_step": tot_t * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 -> i64 / bf16", "data": "synthetic",
            "config": {"workload": "LFQ pack / unpack on 16 x 131 072 tokens (109 MB of latents), GroupNorm(32)+swish on [8, 65 536, 128] fp32 "
                                   "(268 MB: the decoder's 128-channel 256x256 stage); value = sum of algorithmic bytes / sum of kernel times",
                       "kernels": rows, "copy_kernel_GBps_same_process": copy},
            "roofline": {"bound": "hbm", "kernel": "gn_apply_kernel (GroupNorm + swish -> (hi, lo) bf16)", "achieved": dom["GBps"], "peak": 8000.0,
                         "unit": "GB/s", "frac": dom["GBps"] / 8000.0, "traffic": None,
                         "frac_of_copy_ceiling": dom["GBps"] / copy},
            "cpu_baseline": None}


def vq_hbm_quick(L):
    """the default bench line's `vq_hbm` field (north_star: "achieved HBM GB/s on the VQ/argmin path" in a DRIVER-run record): the LFQ
    sign-pack / unpack and the GroupNorm + swish apply pass at bandwidth-deciding sizes, about 100 ms in total.  Algorithmic bytes
    (each tensor once) / launch time from HIP events on the launch stream."""
    s = L.stream

    def timed(fn, reps=5):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    B, Cz, hw = 16, 13, 131072
    z = torch.randn(B, Cz, hw, device="cuda")
    ids = torch.empty(B, hw, dtype=torch.int64, device="cuda")
    by = B * hw * (Cz * 4 + 8)
    t_pack = timed(lambda: L.call("showo_lfq_pack_nchw", L.ptr(z), L.ptr(ids), B, Cz, hw, s()))
    zq = torch.empty_like(z)
    t_unpack = timed(lambda: L.call("showo_lfq_unpack_nchw", L.ptr(ids), L.ptr(zq), B, Cz, hw, s()))
    assert torch.equal(zq > 0, z > 0)
    del z, zq, ids
    Bn, HW, Cc = 8, 65536, 128
    x = torch.randn(Bn, HW, Cc, device="cuda")
    stats = torch.empty(L.load().showo_gn_stats_doubles(Bn, HW), dtype=torch.float64, device="cuda")
    gam, bet = torch.randn(Cc, device="cuda"), torch.randn(Cc, device="cuda")
    y, ylo = torch.empty(Bn, HW, Cc, dtype=torch.int16, device="cuda"), torch.empty(Bn, HW, Cc, dtype=torch.int16, device="cuda")
    t_stats = timed(lambda: L.call("showo_gn_stats", L.ptr(x), L.ptr(stats), Bn, HW, Cc, s()))
    t_apply = timed(lambda: L.call("showo_gn_apply", L.ptr(x), L.ptr(stats), L.ptr(gam), L.ptr(bet), L.ptr(y), L.ptr(ylo), Bn, HW, Cc, 1e-6, 1, s()))
    return {"lfq_pack_GBps": by / t_pack / 1e9, "lfq_unpack_GBps": by / t_unpack / 1e9, "gn_stats_GBps": x.numel() * 4 / t_stats / 1e9,
            "gn_apply_GBps": x.numel() * 8 / t_apply / 1e9, "peak_GBps": 8000.0,
            "sizes": "LFQ: 16 x 131 072 tokens x 13 fp32 channels (60 B per token); GroupNorm + swish -> (hi, lo) bf16 on [8, 65 536, 128] fp32 (8 B per element)"}


def main(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", type=int, default=0, help="Showo.set_precision: 0 bf16 operands (default), 1 accuracy mode, 2 fp16 operands")
    a, _ = ap.parse_known_args(argv)
    if a.gpus != 1:
        raise SystemExit("bench: the t2i512 / mmu / vq workloads are single-GPU lines (replicas scale like the headline)")
    torch.cuda.set_device(0)
    out = {"t2i512": t2i512, "mmu": mmu, "vq": vq}[a.workload](a)
    if a.precision and isinstance(out.get("config"), dict):
        out["config"]["precision"] = {1: "accuracy mode (split-bf16)", 2: "fp16 operands, split-bf16 lm_head"}.get(a.precision, str(a.precision))
        out["dtype"] = {1: "bf16x3", 2: "f16"}.get(a.precision, out.get("dtype"))
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1:])
