"""GPU parity tests of the IEEE-half ("fp16") operand type (round 6; `Showo.set_precision(2)`, the `_op16` entry points with
op = SHOWO_OP_F16), through the C ABI.

Kernel level: every kernel that has an fp16 instance is compared with the fp32 / fp64 oracle evaluated on the SAME fp16-rounded
operands, so the bound is accumulation order + ONE output rounding (2^-11 relative for a 16-bit output, 1e-3 of the output scale for
fp32 outputs -- the bounds of the bf16 tests with 2^-8 replaced by 2^-11).  The fp16 instances share tiles, phase programs, split-K
and epilogue code with the bf16 ones (template parameter F16, csrc/common.h Op16), so the shapes below walk the same dispatch
paths: 128^2 kernel (M < 256), GEMV (M <= 8), production kernel with forced tile variants from each family, split-K, the
K-concatenated residual form, the fused [Wqkv ; W1] epilogue, both attention kernels and the single-query decode kernel.
Module level (reference fixtures, end to end): tests/test_modules_gpu.py (full-size fixtures) and the tiny-model tests below."""
import numpy as np
import pytest
import torch

import util
from util import O, dev

pytestmark = pytest.mark.gpu

F16 = 1  # SHOWO_OP_F16
FP16_TOL = 1e-3  # north_star: "logits within 1e-3" -- rel_max and rel_rms against the fp32 reference, end to end


def L():
    return util.lib()


def S():
    return util.lib().stream()


def h16(t):
    """fp32 -> IEEE-half bit pattern (int16 storage), saturating like the kernels' converts"""
    return t.clamp(-65504.0, 65504.0).to(torch.float16).view(torch.int16)


def f16r(t):
    return t.clamp(-65504.0, 65504.0).to(torch.float16).to(torch.float32)


def from_h16(t):
    return t.view(torch.float16).to(torch.float32)


def test_cast_layernorm_and_saturation_counter():
    torch.manual_seed(0)
    x = torch.randn(1000 * 7 + 3) * 50
    x[5], x[6], x[7], x[8] = 1e6, -1e6, 3e-6, -7e-8  # saturate; subnormals are kept, not flushed
    y = torch.empty(x.numel(), dtype=torch.int16, device="cuda")
    xd = dev(x)
    L().call("showo_cast_f32_op16", L().ptr(xd), L().ptr(y), x.numel(), F16, S())
    assert torch.equal(y.cpu(), h16(x))  # RNE + saturation: bit-exact with torch's conversion of the clamped value
    got = from_h16(y.cpu())
    assert got[5] == 65504.0 and got[6] == -65504.0 and got[7] != 0 and got[8] != 0
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    L().call("showo_count_f16_saturated", L().ptr(y), x.numel(), L().ptr(cnt), S())
    assert int(cnt) == 2
    y0 = torch.empty(x.numel(), dtype=torch.int16, device="cuda")
    L().call("showo_cast_f32_op16", L().ptr(xd), L().ptr(y0), x.numel(), 0, S())
    assert torch.equal(y0.cpu(), util.to_bf16_bits(x))  # op 0 = the bf16 entry point
    for rows, H in ((37, 2048), (5, 128), (3, 100)):
        xx = torch.randn(rows, H) * 3 + 0.5
        w, b = torch.randn(H) * 0.1 + 1, torch.randn(H) * 0.05
        yy = torch.empty((rows, H), dtype=torch.int16, device="cuda")
        L().call("showo_layernorm_f32_op16", L().ptr(dev(xx)), L().ptr(dev(w)), L().ptr(dev(b)), L().ptr(yy), None, rows, H, 1e-5, F16, S())
        want = O.layer_norm(xx, w, b, 1e-5)
        assert (from_h16(yy.cpu()) - want).abs().max() <= 2 ** -11 * want.abs().max() + 1e-6
        y16 = torch.empty((rows, H), dtype=torch.int16, device="cuda")
        L().call("showo_layernorm_f32_bf16", L().ptr(dev(xx)), L().ptr(dev(w)), L().ptr(dev(b)), L().ptr(y16), None, rows, H, 1e-5, S())
        yb = torch.empty((rows, H), dtype=torch.int16, device="cuda")
        L().call("showo_layernorm_f32_op16", L().ptr(dev(xx)), L().ptr(dev(w)), L().ptr(dev(b)), L().ptr(yb), None, rows, H, 1e-5, 0, S())
        assert torch.equal(yb, y16)


def _gemm(A, W, bias, epi, resid=None, op=F16):
    M, K = A.shape
    N = W.shape[0]
    enc = h16 if op else util.to_bf16_bits
    Ad, Wd = dev(enc(A)), dev(enc(W))
    f32 = epi in (2, 3)
    out = torch.full((M, N), float("nan") if f32 else 0, dtype=torch.float32 if f32 else torch.int16, device="cuda")
    rd = None if resid is None else dev(resid)
    L().call("showo_gemm_op16", L().ptr(Ad), K, L().ptr(Wd), K, L().ptr(dev(bias)), 0, L().ptr(out), N, L().ptr(rd), N if rd is not None else 0,
             M, N, K, epi, op, S())
    torch.cuda.synchronize()
    return (out if f32 else (from_h16(out) if op else util.from_bf16_bits(out))).cpu()


# M = 1 / 5: GEMV; 200 / 129: 128^2 kernel; 774 / 1548: production kernel; (631, 2048, 2048) and (577, 1024, 4096): split-K
@pytest.mark.parametrize("M,N,K", [(1, 512, 256), (5, 384, 512), (200, 256, 128), (129, 130, 64), (774, 2048, 2048), (1548, 768, 256),
                                   (631, 2048, 2048), (577, 1024, 4096)])
def test_gemm_fp16_epilogues(M, N, K):
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K)
    W = torch.randn(N, K) * 0.05
    bias = torch.randn(N)
    ref = f16r(A).double() @ f16r(W).double().T + bias.double()
    tol = 1e-3 * float(ref.abs().max())
    got = _gemm(A, W, bias, 2)
    err = float((got.double() - ref).abs().max())
    print(f"[parity] fp16 GEMM M={M} N={N} K={K}: fp32 epilogue max err {err:.3e} (scale {float(ref.abs().max()):.2f})")
    assert err < tol
    got = _gemm(A, W, bias, 0)
    assert (got.double() - ref).abs().max() < 2 ** -11 * float(ref.abs().max()) + tol
    got = _gemm(A, W, bias, 1)
    want = O.gelu_new(ref.float()).double()
    assert (got.double() - want).abs().max() < 2 ** -11 * float(want.abs().max()) + tol
    resid = torch.randn(M, N)
    got = _gemm(A, W, bias, 3, resid=resid)
    assert (got.double() - (ref + resid.double())).abs().max() < tol
    # op = SHOWO_OP_BF16 through the same entry point is the bf16 GEMM
    if M >= 129:
        got0 = _gemm(A, W, bias, 2, op=0)
        ref0 = util.bf16_round(A).double() @ util.bf16_round(W).double().T + bias.double()
        assert (got0.double() - ref0).abs().max() < tol


@pytest.mark.parametrize("variant", [256, 224, 176, 144, 1192, 1160, 1128, 2256, 2208, 3192, 3144, 4176])
def test_gemm_fp16_every_tile_family_gives_the_same_bits(variant):
    """forced tile variants of the production kernel (m-split, n-split, weight ring, buffer-descriptor DMAs): every fp16 instance
    computes one fp32 chain over k in order per output element, so all variants agree bit for bit (as the bf16 ones do)"""
    torch.manual_seed(3)
    M, N, K = 1290, 1024, 1024
    A, W, bias = torch.randn(M, K), torch.randn(N, K) * 0.05, torch.randn(N)
    try:
        L().call("showo_gemm_set_impl", 5)
        L().call("showo_gemm_tune", 8, 256 << 8, None)
        base = _gemm(A, W, bias, 2)
        L().call("showo_gemm_tune", 8, variant << 8, None)
        got = _gemm(A, W, bias, 2)
        gotg = _gemm(A, W, bias, 1)
    finally:
        L().call("showo_gemm_tune", 8, 0, None)
        L().call("showo_gemm_set_impl", 0)
    assert torch.equal(got, base)
    ref = f16r(A).double() @ f16r(W).double().T + bias.double()
    assert (got.double() - ref).abs().max() < 1e-3 * float(ref.abs().max())
    want = O.gelu_new(ref.float()).double()
    assert (gotg.double() - want).abs().max() < 2 ** -11 * float(want.abs().max()) + 1e-3 * float(ref.abs().max())


def _tiled(W):
    N, K = W.shape
    out = torch.zeros(int(L().load().showo_gemm_tiled_elems(N, K)), dtype=torch.int16, device="cuda")
    L().call("showo_gemm_tile_weight", L().ptr(W), K, N, K, L().ptr(out), S())
    return out


@pytest.mark.parametrize("tiled", [0, 1])
@pytest.mark.parametrize("M,K0,K1,N", [(700, 128, 512, 384), (1548, 256, 1024, 2048), (631, 2048, 8192, 2048)])
def test_gemm_kcat_residual_fp16(M, K0, K1, N, tiled):
    """x += [attn | ffn] [Wd | W2]^T + b with fp16 operands (the second launch of a layer; (631, .) = the split-K prefill shape)"""
    torch.manual_seed(M + K0)
    A0, A1 = torch.randn(M, K0), torch.randn(M, K1)
    W = torch.randn(N, K0 + K1) * 0.03
    bias, resid = torch.randn(N), torch.randn(M, N)
    Wd = dev(h16(W))
    Wuse = _tiled(Wd) if tiled else Wd  # (held in a local: the C ABI takes raw pointers)
    x = dev(resid.clone())
    L().call("showo_gemm_kcat_op16", L().ptr(dev(h16(A0))), K0, K0, L().ptr(dev(h16(A1))), K1, K1, L().ptr(Wuse), K0 + K1,
             L().ptr(dev(bias)), L().ptr(x), N, L().ptr(x), N, M, N, 3, tiled, F16, S())
    torch.cuda.synchronize()
    ref = torch.cat([f16r(A0), f16r(A1)], 1).double() @ f16r(W).double().T + bias.double() + resid.double()
    err = float((x.cpu().double() - ref).abs().max())
    print(f"[parity] fp16 K-concatenated residual GEMM M={M}: max err {err:.3e} (scale {float(ref.abs().max()):.2f})")
    assert err < 1e-3 * float(ref.abs().max())


@pytest.mark.parametrize("B,Lq,nH,F,pos0", [(2, 387, 4, 512, 0), (3, 130, 4, 1024, 0), (1, 300, 4, 256, 40)])
def test_fused_qkv_fc1_projection_fp16(B, Lq, nH, F, pos0):
    """the first launch of a layer with fp16 operands: bias + q/k LayerNorm(64) + partial RoPE + head-major relayout of Q / K / V^T from
    the fp32 accumulators, bias + gelu_new for the fc1 columns; every output rounded ONCE to fp16"""
    torch.manual_seed(B * 1000 + Lq)
    H = nH * 64
    h = torch.randn(B * Lq, H)
    W = torch.randn(3 * H + F, H) * 0.05
    bias = torch.randn(3 * H + F) * 0.1
    qw, qb, kw, kb = torch.randn(64) * .1 + 1, torch.randn(64) * .05, torch.randn(64) * .1 + 1, torch.randn(64) * .05
    cos, sin = O.rope_tables(32, 2048, 10000.0)
    Lcap = pos0 + Lq + 3
    Lp = ((pos0 + Lq + 63) // 64) * 64
    Q = torch.zeros((B, nH, Lq, 64), dtype=torch.int16, device="cuda")
    K = torch.zeros((B, nH, Lcap, 64), dtype=torch.int16, device="cuda")
    Vt = torch.zeros((B, nH, 64, Lp), dtype=torch.int16, device="cuda")
    ffn = torch.zeros((B * Lq, F), dtype=torch.int16, device="cuda")
    L().call("showo_gemm_qkv_fc1_op16", L().ptr(dev(h16(h))), H, L().ptr(dev(h16(W))), H, L().ptr(dev(bias)), L().ptr(dev(qw)), L().ptr(dev(qb)),
             L().ptr(dev(kw)), L().ptr(dev(kb)), L().ptr(dev(cos)), L().ptr(dev(sin)), L().ptr(Q), L().ptr(K), L().ptr(Vt), L().ptr(ffn), F, F,
             B, Lq, nH, 32, 1e-5, pos0, Lcap, Lp, 0, F16, S())
    torch.cuda.synchronize()
    acc = (f16r(h).double() @ f16r(W).double().T + bias.double()).float()
    x = acc[:, :3 * H].view(B, Lq, 3, nH, 64)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    cs, sn = cos[pos0:pos0 + Lq], sin[pos0:pos0 + Lq]
    q = O.apply_partial_rope(O.layer_norm(q, qw, qb, 1e-5), cs, sn, 32) * 0.125
    k = O.apply_partial_rope(O.layer_norm(k, kw, kb, 1e-5), cs, sn, 32)
    e11 = 2 ** -11
    assert (from_h16(Q.cpu()) - q).abs().max() < e11 * float(q.abs().max()) + 1e-5
    Kc = from_h16(K.cpu())
    assert (Kc[:, :, pos0:pos0 + Lq] - k).abs().max() < e11 * float(k.abs().max()) + 1e-5
    assert (Kc[:, :, :pos0] == 0).all() and (Kc[:, :, pos0 + Lq:] == 0).all()
    vt = from_h16(Vt.cpu())
    assert (vt[..., pos0:pos0 + Lq] - v.transpose(2, 3)).abs().max() < e11 * float(v.abs().max()) + 1e-5
    g = O.gelu_new(acc[:, 3 * H:])
    assert (from_h16(ffn.cpu()) - g).abs().max() < e11 * float(g.abs().max()) + 2e-4 * float(g.abs().max())  # + the fast exp / rcp of gelu_new_fast
    # the projection alone (ffn_out = NULL) writes the same Q / K / V^T bits
    Q2, K2, Vt2 = torch.zeros_like(Q), torch.zeros_like(K), torch.zeros_like(Vt)
    L().call("showo_gemm_qkv_fc1_op16", L().ptr(dev(h16(h))), H, L().ptr(dev(h16(W[:3 * H].contiguous()))), H, L().ptr(dev(bias[:3 * H].contiguous())),
             L().ptr(dev(qw)), L().ptr(dev(qb)), L().ptr(dev(kw)), L().ptr(dev(kb)), L().ptr(dev(cos)), L().ptr(dev(sin)), L().ptr(Q2), L().ptr(K2),
             L().ptr(Vt2), None, 0, 0, B, Lq, nH, 32, 1e-5, pos0, Lcap, Lp, 0, F16, S())
    assert torch.equal(Q2, Q) and torch.equal(K2, K) and torch.equal(Vt2, Vt)


def _prep16(qkv, qw, qb, kw, kb, B, Lq, nH):
    cos, sin = O.rope_tables(32, 2048, 10000.0)
    Lp = ((Lq + 63) // 64) * 64
    Q = torch.zeros((B, nH, Lq, 64), dtype=torch.int16, device="cuda")
    K = torch.zeros((B, nH, Lq, 64), dtype=torch.int16, device="cuda")
    Vt = torch.zeros((B, nH, 64, Lp), dtype=torch.int16, device="cuda")
    L().call("showo_qk_prep_op16", L().ptr(dev(h16(qkv))), L().ptr(dev(qw)), L().ptr(dev(qb)), L().ptr(dev(kw)), L().ptr(dev(kb)), L().ptr(dev(cos)),
             L().ptr(dev(sin)), L().ptr(Q), L().ptr(K), L().ptr(Vt), B, Lq, nH, 32, 1e-5, 0, Lq, Lp, F16, S())
    torch.cuda.synchronize()
    return Q, K, Vt


def _attn_oracle16(Q, K, Vt, mask, Lq, Lk):
    q, k = from_h16(Q.cpu()), from_h16(K.cpu())[:, :, :Lk]
    v = from_h16(Vt.cpu())[..., :Lk].transpose(2, 3)
    o = torch.softmax(q @ k.transpose(2, 3) + mask, dim=-1) @ v
    B, nH = o.shape[:2]
    return o.transpose(1, 2).reshape(B, Lq, nH * 64)


@pytest.mark.parametrize("impl", [1, 2], ids=["gather", "lds-tiled"])
@pytest.mark.parametrize("Lq,nH", [(27, 2), (387, 2), (1155, 1)])
def test_attention_fp16_mask_families(Lq, nH, impl):
    """qk_prep + attention with fp16 Q / K / V^T / P / O under the reference's mask families (t2i, mmu, causal lm), both kernels;
    P and O are rounded to fp16 once each: 2^-11 relative to max|V| is the expected scale (the bf16 test's bound with 2^-8 -> 2^-11)"""
    import test_kernels_gpu as TK
    d = util.tiny_dims()
    torch.manual_seed(Lq)
    L().call("showo_attn_set_impl", impl)
    try:
        for name, mask in TK._mask_cases(d, Lq):
            B = mask.shape[0]
            qkv = torch.randn(B * Lq, 3 * nH * 64) * 2
            qw, qb, kw, kb = torch.randn(64) * .1 + 1, torch.randn(64) * .05, torch.randn(64) * .1 + 1, torch.randn(64) * .05
            Q, K, Vt = _prep16(qkv, qw, qb, kw, kb, B, Lq, nH)
            q, k, v = TK._prep_oracle(qkv, qw, qb, kw, kb, B, Lq, nH)  # (rounds qkv to bf16: only used for the relayout check below)
            assert torch.equal(from_h16(Vt.cpu())[..., :Lq], f16r(qkv).view(B, Lq, 3, nH, 64)[:, :, 2].transpose(1, 2).transpose(2, 3))
            md = dev(mask)
            iv = torch.zeros((B, Lq, 4), dtype=torch.int32, device="cuda")
            flag = torch.zeros(4, dtype=torch.int32, device="cuda")
            Od = torch.zeros((B, Lq, nH * 64), dtype=torch.int16, device="cuda")
            L().call("showo_mask_compress", L().ptr(md), L().ptr(iv), L().ptr(flag), B, Lq, Lq, S())
            L().call("showo_attn_fwd_op16", L().ptr(Q), L().ptr(K), L().ptr(Vt), L().ptr(iv), L().ptr(flag), L().ptr(md), L().ptr(Od), B, nH, Lq, Lq, Lq,
                     Vt.shape[-1], nH * 64, F16, S())
            torch.cuda.synchronize()
            want = _attn_oracle16(Q, K, Vt, mask, Lq, Lq)
            err = float((from_h16(Od.cpu()) - want).abs().max())
            assert err < 2.5 * 2 ** -11 * float(want.abs().max()) + 2e-4, (name, err)
        # dense-mask fallback (a mask no two intervals represent) and the single-query decode form against a longer cache
        B, Lk = 2, 200
        mask = torch.where(torch.rand(B, 1, Lq if Lq < 100 else 100, Lk) < 0.6, 0.0, O.NEG_MASK)
        mask[..., 0] = 0
        Lq2 = mask.shape[2]
        qkv = torch.randn(B * Lk, 3 * nH * 64) * 2
        qw, qb, kw, kb = torch.ones(64), torch.zeros(64), torch.ones(64), torch.zeros(64)
        Q, K, Vt = _prep16(qkv, qw, qb, kw, kb, B, Lk, nH)
        Qs = Q[:, :, :Lq2].contiguous()
        for rows in (Lq2, 1):
            Qr = Qs[:, :, :rows].contiguous()
            mk = mask[:, :, :rows].contiguous()
            md = dev(mk)
            iv = torch.zeros((B, rows, 4), dtype=torch.int32, device="cuda")
            flag = torch.zeros(4, dtype=torch.int32, device="cuda")
            Od = torch.zeros((B, rows, nH * 64), dtype=torch.int16, device="cuda")
            L().call("showo_mask_compress", L().ptr(md), L().ptr(iv), L().ptr(flag), B, rows, Lk, S())
            L().call("showo_attn_fwd_op16", L().ptr(Qr), L().ptr(K), L().ptr(Vt), L().ptr(iv), L().ptr(flag), L().ptr(md), L().ptr(Od), B, nH, rows, Lk, Lk,
                     Vt.shape[-1], nH * 64, F16, S())
            torch.cuda.synchronize()
            assert int(flag[0]) == 1
            want = _attn_oracle16(Qr, K, Vt, mk, rows, Lk)
            err = float((from_h16(Od.cpu()) - want).abs().max())
            assert err < 2.5 * 2 ** -11 * float(want.abs().max()) + 2e-4, (rows, err)
    finally:
        L().call("showo_attn_set_impl", 0)


# ------------------------------------------------------------------------------------------------- module level, tiny model
def _check(got, ref, what, tol=FP16_TOL):
    rmax, rrms = util.relerr(got, ref)
    print(f"[parity] precision 2 (fp16 operands), {what}: rel_max={rmax:.3e} rel_rms={rrms:.3e}")
    assert rmax <= tol and rrms <= tol, (what, rmax, rrms)
    return rmax


def test_tiny_fp16_mode_logits_trajectory_decode_and_switching():
    """precision 2 on the tiny test model (K = 128: the generic layer path, 128^2 GEMM, gather attention, split-bf16 head) against the
    REFERENCE's fp32 outputs: logits of every mask family, the 6-step t2i trajectory under the reference's noise, greedy and stochastic
    mmu tokens through the KV cache, the range check, and switching 2 -> 0 -> 1 -> 2 on one engine (weight images change type)."""
    g = util.golden("showo_tiny_forward.npz")
    d, sd = util.tiny_state()
    m = util.build_showo(d, sd).set_precision(2)
    for key in ("t2i", "mmu", "train"):
        lg = m(dev(g[f"{key}_ids"]), attention_mask=dev(g[f"{key}_mask"]))
        assert lg.dtype == torch.float32
        _check(lg, torch.from_numpy(g[f"{key}_logits"]), f"tiny {key} logits vs the fp32 reference")
    assert util.lib().load().showo_engine_get_precision(m.engine()) == 2
    n_sat = m.range_check(lambda: m(dev(g["t2i_ids"]), attention_mask=dev(g["t2i_mask"])))
    print(f"[parity] precision 2 range check on the tiny model: {n_sat} saturated fp16 activations")
    assert n_sat == 0
    # interval masks built on the device give the same bits as the dense reference masks
    P = util.pkg().prompting_utils
    ids = dev(g["t2i_ids"])
    iv = P.intervals_predict_next(ids, pad_id=d.pad_id, soi_id=d.soi_id, eoi_id=d.eoi_id, rm_pad_in_image=True)
    lg2 = m(ids, attention_mask=dev(g["t2i_mask"]))
    assert torch.equal(m(ids, attention_mask=iv), lg2)
    # t2i trajectory with the reference's recorded noise: identical tokens, per-step logits within the tolerance
    g2 = util.golden("showo_tiny_t2i.npz")
    steps, B = int(g2["steps"]), g2["ids_cond"].shape[0]
    N, V = d.num_vq_tokens, d.codebook
    ids_c = dev(g2["ids_cond"]).clone()
    out = m.t2i_generate(input_ids=ids_c, uncond_input_ids=dev(g2["ids_uncond"]), attention_mask=dev(g2["mask"]), timesteps=steps,
                         guidance_scale=float(g2["guidance"]), config=util.gen_config(d),
                         _exp_noise=dev(g2["exp_noise"].reshape(steps, B * N, V)), _uniform=dev(g2["uniform"].reshape(steps, B, N)))
    same = float((out.cpu() == torch.from_numpy(g2["result"])).float().mean())
    print(f"[parity] precision 2 tiny t2i trajectory under the reference's noise: {same:.4f} of the tokens identical")
    assert same == 1.0 and torch.equal(ids_c.cpu(), torch.from_numpy(g2["final_input_ids"]))
    for s in range(steps):
        _check(m(dev(g2["fwd_in"][s]), attention_mask=dev(g2["mask"])), torch.from_numpy(g2["fwd_logits"][s]), f"teacher-forced step {s}")
    # mmu_generate through the KV cache (the general decode layer with fp16 operands): the reference's tokens
    g3 = util.golden("showo_tiny_mmu.npz")
    toks = m.mmu_generate(dev(g3["ids"]), attention_mask=dev(g3["mask"]), max_new_tokens=len(g3["tokens"]), top_k=1)
    assert [int(t) for t in toks] == g3["tokens"].tolist()
    for tag, kw in (("topk5", dict(top_k=5, temperature=0.7)), ("full", dict(top_k=None, temperature=1.3))):
        toks = m.mmu_generate(dev(g3["ids"]), attention_mask=dev(g3["mask"]), max_new_tokens=8, _exp_noise=dev(g3[f"exp_noise_{tag}"]), **kw)
        assert [int(t) for t in toks] == g3[f"tokens_{tag}"].tolist(), tag
    # 2 -> 0: the bf16 path's own bits; -> 1: accuracy mode; -> 2: the same fp16 bits as before
    ref0 = util.build_showo(d, sd)(dev(g["t2i_ids"]), attention_mask=dev(g["t2i_mask"]))
    assert torch.equal(m.set_precision(0)(dev(g["t2i_ids"]), attention_mask=dev(g["t2i_mask"])), ref0)
    r1 = util.relerr(m.set_precision(1)(dev(g["t2i_ids"]), attention_mask=dev(g["t2i_mask"])), torch.from_numpy(g["t2i_logits"]))
    assert r1[0] <= 1e-4
    assert torch.equal(m.set_precision(2)(dev(g["t2i_ids"]), attention_mask=dev(g["t2i_mask"])), lg2)
    # a training step on a model in precision 2 runs on bf16 images (the trainer's engine is precision 0) and inference comes back in fp16
    m.train()
    tr = util.pkg().Trainer(m, lr=1e-3)
    tr.step(dev(g["train_ids"]), dev(g["train_mask"]), dev(g["train_labels"]), 2, 1, 2, d.max_text_len)
    m.eval()
    sd_now = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    want = O.showo_logits(sd_now, d, torch.from_numpy(g["t2i_ids"]), attention_mask=torch.from_numpy(g["t2i_mask"]))
    _check(m(dev(g["t2i_ids"]), attention_mask=dev(g["t2i_mask"])), want, "tiny t2i logits after one optimizer step vs the oracle on the updated weights")
    assert util.lib().load().showo_engine_get_precision(m.engine()) == 2


def test_tiny_fp16_mode_vs_rounding_point_oracle_with_fp16_points():
    """the rounding-point oracle with dtype = float16 (lm_head sites exempt: the head is a split-bf16 product) models the GPU's
    precision 2 the way Bf16Points models precision 0: what remains is accumulation order and the occasional flipped rounding"""
    g = util.golden("showo_tiny_forward.npz")
    d, sd = util.tiny_state()
    m = util.build_showo(d, sd).set_precision(2)
    sdt = O.to_torch(sd)
    sites = [s for s in O.Bf16Points.SITES if s not in ("w_lm", "hf")]
    for key in ("t2i", "mmu"):
        ids, mask = torch.from_numpy(g[f"{key}_ids"]), torch.from_numpy(g[f"{key}_mask"])
        want = O.showo_logits(sdt, d, ids, attention_mask=mask, pts=O.Bf16Points(qkv_round=True, dtype=torch.float16, sites=sites))
        got = m(dev(g[f"{key}_ids"]), attention_mask=dev(g[f"{key}_mask"]))
        rmax, rrms = util.relerr(got, want)
        ref = util.relerr(want, torch.from_numpy(g[f"{key}_logits"]))
        print(f"[parity] precision 2 tiny {key} logits vs the fp16 rounding-point oracle: rel_max={rmax:.3e} rel_rms={rrms:.3e} "
              f"(that oracle vs the fp32 reference: {ref[0]:.3e} / {ref[1]:.3e})")
        assert rmax <= 1e-3 and rrms <= 5e-4
