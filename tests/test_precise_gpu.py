"""GPU parity tests of ACCURACY MODE ON THE PRODUCTION KERNELS (round 5): the K-concatenated split-bf16 forms.

A split-precision product a.w = a_hi w_hi + a_lo w_hi + a_hi w_lo is ONE bf16 GEMM over K' = 3K between A' = [a_hi | a_lo | a_hi] and
W' rows [w_hi | w_hi | w_lo] (fp32 accumulation in the MFMA), so the production GEMM kernels run unchanged; the fused projection
epilogue and the attention have (hi, lo)-pair forms.  Everything here is checked against an fp64 evaluation of the UNROUNDED fp32
operands -- i.e. against the reference's fp32 arithmetic (inference_t2i.py:67, models/phi.py:657-722, 204-212) -- with the
tolerance north_star states divided by 30: 3e-5 of the output scale (measured: 2-8e-6)."""
import numpy as np
import pytest
import torch

import util
from util import O, dev, from_bf16_bits, to_bf16_bits, bf16_round

pytestmark = pytest.mark.gpu
TOL = 3e-5


def L():
    return util.lib()


def S():
    return util.lib().stream()


def split(t):
    """fp32 -> (hi, lo) fp32 tensors holding the bf16 values: hi = RNE(t), lo = RNE(t - hi)"""
    hi = bf16_round(t)
    return hi, bf16_round(t - hi)


def cat3_act(t):
    hi, lo = split(t)
    return torch.cat([hi, lo, hi], dim=-1)


def cat3_w(w):
    hi, lo = split(w)
    return torch.cat([hi, hi, lo], dim=-1)


def pair(hi_bits, lo_bits):
    return from_bf16_bits(hi_bits).double() + from_bf16_bits(lo_bits).double()


def _rope():
    return O.rope_tables(32, 2048, 10000.0)


@pytest.mark.parametrize("variant", [0, 256, 224, 1160, 2256, 2224, 3192, 4160])
@pytest.mark.parametrize("B,Lq,nH,F,pos0,tiled", [(2, 387, 4, 512, 0, 1), (1, 300, 4, 256, 0, 0), (2, 258, 4, 512, 129, 1), (1, 1, 4, 256, 40, 1)])
def test_split_projection_tracks_fp32_reference(variant, B, Lq, nH, F, pos0, tiled):
    """showo_gemm_qkv_fc1_split: Q / K / V^T / gelu(fc1) as (hi, lo) pairs vs the fp64 evaluation of the fp32 operands"""
    torch.manual_seed(B * 1000 + Lq + F)
    H = nH * 64
    h = torch.randn(B * Lq, H)
    W = torch.randn(3 * H + F, H) * 0.05
    bias = torch.randn(3 * H + F) * 0.1
    qw, qb, kw, kb = torch.randn(64) * .1 + 1, torch.randn(64) * .05, torch.randn(64) * .1 + 1, torch.randn(64) * .05
    cos, sin = _rope()
    Lcap = pos0 + Lq + 3
    Lp = ((pos0 + Lq + 63) // 64) * 64
    z = lambda *s: torch.zeros(s, dtype=torch.int16, device="cuda")
    Q, Ql, K, Kl, Vt, Vl = z(B, nH, Lq, 64), z(B, nH, Lq, 64), z(B, nH, Lcap, 64), z(B, nH, Lcap, 64), z(B, nH, 64, Lp), z(B, nH, 64, Lp)
    ldf = 2 * (H + F)  # the engine's act layout: [attn_hi | ffn_hi | attn_lo | ffn_lo]
    act = z(B * Lq, ldf)
    A3 = dev(to_bf16_bits(cat3_act(h)))
    W3 = dev(to_bf16_bits(cat3_w(W)))
    N = 3 * H + F
    if tiled:
        Wt = torch.zeros(int(L().load().showo_gemm_tiled_elems(N, 3 * H)), dtype=torch.int16, device="cuda")
        L().call("showo_gemm_tile_weight", L().ptr(W3), 3 * H, N, 3 * H, L().ptr(Wt), S())
        W3 = Wt
    if variant:
        L().call("showo_gemm_tune", 4, variant << 8, None)
    try:
        L().call("showo_gemm_qkv_fc1_split", L().ptr(A3), 3 * H, L().ptr(W3), 3 * H, 3 * H, L().ptr(dev(bias)), L().ptr(dev(qw)),
                 L().ptr(dev(qb)), L().ptr(dev(kw)), L().ptr(dev(kb)), L().ptr(dev(cos)), L().ptr(dev(sin)), L().ptr(Q), L().ptr(Ql),
                 L().ptr(K), L().ptr(Kl), L().ptr(Vt), L().ptr(Vl), act.data_ptr() + 2 * H, act.data_ptr() + 2 * (H + F + H), ldf, F, B, Lq,
                 nH, 32, 1e-5, pos0, Lcap, Lp, tiled, S())
        torch.cuda.synchronize()
    finally:
        L().call("showo_gemm_tune", 4, 0, None)
    y = h.double() @ W.double().T + bias.double()
    x = y[:, :3 * H].view(B, Lq, 3, nH, 64)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    cs, sn = cos[pos0:pos0 + Lq].double(), sin[pos0:pos0 + Lq].double()
    q = O.apply_partial_rope(O.layer_norm(q, qw.double(), qb.double(), 1e-5), cs, sn, 32) * 0.125
    k = O.apply_partial_rope(O.layer_norm(k, kw.double(), kb.double(), 1e-5), cs, sn, 32)
    f = y[:, 3 * H:]
    g = 0.5 * f * (1.0 + torch.tanh(0.7978845608028654 * (f + 0.044715 * f ** 3)))

    def close(got, want, what):
        err = float((got.cpu() - want).abs().max() / want.abs().max())
        assert err < TOL, (what, err)

    close(pair(Q, Ql), q, "Q")
    Kc = pair(K, Kl)
    close(Kc[:, :, pos0:pos0 + Lq], k, "K")
    assert (Kc[:, :, :pos0] == 0).all() and (Kc[:, :, pos0 + Lq:] == 0).all()  # only the new rows are written
    vt = pair(Vt, Vl)
    close(vt[..., pos0:pos0 + Lq], v.transpose(2, 3), "Vt")
    assert (vt[..., :pos0] == 0).all() and (vt[..., pos0 + Lq:] == 0).all()
    a = from_bf16_bits(act).double().cpu()
    close(a[:, H:H + F] + a[:, H + F + H:], g, "gelu(fc1)")
    assert (a[:, :H] == 0).all() and (a[:, H + F:H + F + H] == 0).all()  # the attention's columns are not touched
    # the halves are a proper split: |lo| <= half an ulp of hi (2^-8 |hi|)
    hi, lo = from_bf16_bits(Q).cpu(), from_bf16_bits(Ql).cpu()
    assert (lo.abs() <= 2.0 ** -8 * hi.abs()).all()


def _masks(d, Lq):
    rs = np.random.RandomState(1)
    T = min(129, Lq // 3)
    N = Lq - T - 2
    rows = []
    for k in (3, T, T // 2 + 1):
        rows.append([d.pad_id] * (T - k) + rs.randint(0, 100, size=k).tolist() + [d.soi_id] + [d.mask_token_id] * N + [d.eoi_id])
    yield "t2i", O.mask_t2i(torch.tensor(rows), d.pad_id, d.soi_id, d.eoi_id)
    ids_m = torch.tensor([[d.mmu_id, d.soi_id] + [7] * N + [d.eoi_id] + [5] * (Lq - N - 3)] * 2)
    yield "mmu", O.mask_mmu(ids_m, d.eoi_id)
    if Lq >= 100:
        vis = torch.rand(2, 1, Lq, Lq) < 0.5
        vis |= torch.eye(Lq, dtype=torch.bool)[None, None]
        m = torch.where(vis, torch.zeros(()), torch.full((), O.NEG_MASK))
        m[:, :, :, 3] = -1.5  # soft bias column: the dense fallback must add it
        yield "dense", m


def _split_attn(q, k, v, mask, Lcap=None):
    """q [B,nH,Lq,64] (already scaled), k, v [B,nH,Lk,64] fp32 -> hi + lo of the kernel's output [B,Lq,nH*64] (double)"""
    B, nH, Lq, _ = q.shape
    Lk = k.shape[2]
    Lcap = Lcap or Lk
    Lp = ((Lk + 63) // 64) * 64
    z = lambda *s: torch.zeros(s, dtype=torch.int16, device="cuda")

    def up(t, shape, sl):
        hi, lo = split(t)
        a, b = z(*shape), z(*shape)
        a[sl] = to_bf16_bits(hi).cuda()
        b[sl] = to_bf16_bits(lo).cuda()
        return a, b

    Q, Ql = up(q, (B, nH, Lq, 64), (slice(None),) * 4)
    K, Kl = up(k, (B, nH, Lcap, 64), (slice(None), slice(None), slice(0, Lk)))
    Vt, Vl = up(v.transpose(2, 3).contiguous(), (B, nH, 64, Lp), (slice(None), slice(None), slice(None), slice(0, Lk)))
    ldo = 2 * nH * 64 + 128
    Od = z(B * Lq, ldo)
    iv = torch.zeros((B, Lq, 4), dtype=torch.int32, device="cuda")
    flag = torch.zeros(4, dtype=torch.int32, device="cuda")
    md = None
    if mask is not None:
        md = dev(mask)
        L().call("showo_mask_compress", L().ptr(md), L().ptr(iv), L().ptr(flag), B, Lq, Lk, S())
    L().call("showo_attn_fwd_split", L().ptr(Q), L().ptr(Ql), L().ptr(K), L().ptr(Kl), L().ptr(Vt), L().ptr(Vl),
             L().ptr(iv) if mask is not None else None, L().ptr(flag) if mask is not None else None, L().ptr(md), L().ptr(Od),
             Od.data_ptr() + 2 * (nH * 64 + 64), B, nH, Lq, Lk, Lcap, Lp, ldo, S())
    torch.cuda.synchronize()
    o = from_bf16_bits(Od).double().cpu()
    H = nH * 64
    assert (o[:, H:H + 64] == 0).all() and (o[:, 2 * H + 64:] == 0).all()
    return (o[:, :H] + o[:, H + 64:2 * H + 64]).view(B, Lq, H), int(flag[0])


@pytest.mark.parametrize("Lq,nH", [(27, 2), (258, 2), (387, 2), (1155, 1)])
def test_split_attention_tracks_fp32_sdpa(Lq, nH):
    d = util.tiny_dims()
    torch.manual_seed(Lq)
    for name, mask in _masks(d, Lq):
        B = mask.shape[0]
        # the model's scale: q, k leave a LayerNorm (unit variance) and q carries the 1/8 -> scores of std ~1; the dropped lo*lo terms and
        # the roundings of the low halves perturb a score by ~4e-6 |q||k| and the soft-max turns that into a RELATIVE error of P, so
        # the bound scales with the score magnitude (std 2 here: measured 1e-5; std 8: 4e-5)
        q = torch.randn(B, nH, Lq, 64) * 0.25
        k = torch.randn(B, nH, Lq, 64)
        v = torch.randn(B, nH, Lq, 64)
        got, flag = _split_attn(q, k, v, mask)
        assert flag == (1 if name == "dense" else 0), name
        s = q.double() @ k.double().transpose(2, 3) + mask.double()
        want = (torch.softmax(s, dim=-1) @ v.double()).transpose(1, 2).reshape(B, Lq, nH * 64)
        err = float((got - want).abs().max() / want.abs().max())
        assert err < TOL, (name, err)


def test_split_attention_decode_row_and_rescale_branch():
    """Lq = 1 against a cache (the accuracy-mode decode step: Lk < Lcap, causal default), and a dominating late key (running-max
    rescale, CDNA playbook rule 26)"""
    torch.manual_seed(9)
    nH, Lk = 3, 650
    q = torch.randn(1, nH, 1, 64) * 0.5
    k = torch.randn(1, nH, Lk, 64) * 2
    v = torch.randn(1, nH, Lk, 64)
    got, _ = _split_attn(q, k, v, None, Lcap=704)
    want = (torch.softmax(q.double() @ k.double().transpose(2, 3), dim=-1) @ v.double()).transpose(1, 2).reshape(1, 1, nH * 64)
    assert float((got - want).abs().max() / want.abs().max()) < TOL
    Lq = 200
    q = torch.randn(1, 1, Lq, 64) * 0.5
    k = torch.randn(1, 1, Lq, 64)
    v = torch.randn(1, 1, Lq, 64)
    k[0, 0, 170] = q[0, 0, 180] * 20.0
    mask = torch.zeros(1, 1, Lq, Lq)
    got, _ = _split_attn(q, k, v, mask)
    want = (torch.softmax(q.double() @ k.double().transpose(2, 3), dim=-1) @ v.double()).transpose(1, 2).reshape(1, Lq, 64)
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max() / want.abs().max()) < TOL


@pytest.mark.parametrize("M,H,F", [(774, 256, 1024), (300, 128, 512), (1, 256, 1024)])
def test_split_residual_gemm_on_the_act_layout(M, H, F):
    """The engine's residual launch in accuracy mode: x += [attn | ffn] [Wd | W2]^T + b as ONE showo_gemm_kcat_bf16 over
    K' = 3 (H + F): A0 = act (all 2 (H + F) columns = [attn_hi | ffn_hi | attn_lo | ffn_lo]), A1 = act again (its first H + F columns),
    weight rows [Wd_hi | W2_hi | Wd_hi | W2_hi | Wd_lo | W2_lo] (tiled)."""
    torch.manual_seed(M + H)
    HF = H + F
    attn, ffn = torch.randn(M, H), torch.randn(M, F)
    Wd, W2 = torch.randn(H, H) * 0.05, torch.randn(H, F) * 0.03
    b, x = torch.randn(H) * 0.1, torch.randn(M, H)
    ah, al = split(attn)
    fh, fl = split(ffn)
    act = dev(to_bf16_bits(torch.cat([ah, fh, al, fl], dim=1)))
    dh, dl = split(Wd)
    wh, wl = split(W2)
    W3 = dev(to_bf16_bits(torch.cat([dh, wh, dh, wh, dl, wl], dim=1)))
    Wt = torch.zeros(int(L().load().showo_gemm_tiled_elems(H, 3 * HF)), dtype=torch.int16, device="cuda")
    L().call("showo_gemm_tile_weight", L().ptr(W3), 3 * HF, H, 3 * HF, L().ptr(Wt), S())
    xd = dev(x)
    L().call("showo_gemm_kcat_bf16", L().ptr(act), 2 * HF, 2 * HF, L().ptr(act), 2 * HF, HF, L().ptr(Wt), 3 * HF, L().ptr(dev(b)),
             L().ptr(xd), H, L().ptr(xd), H, M, H, 3, 1, S())
    torch.cuda.synchronize()
    want = x.double() + attn.double() @ Wd.double().T + ffn.double() @ W2.double().T + b.double()
    assert float((xd.cpu().double() - want).abs().max() / want.abs().max()) < TOL
