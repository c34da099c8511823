"""GPU parity tests of the training (backward) kernels against torch autograd of the CPU oracle's formulas."""
import numpy as np
import pytest
import torch

import util
from util import O, dev, from_bf16_bits, to_bf16_bits, bf16_round

pytestmark = pytest.mark.gpu


def L():
    return util.lib()


def S():
    return util.lib().stream()


def _bits(t):
    return dev(to_bf16_bits(t))


@pytest.mark.parametrize("B,nH,Lq,kind", [(2, 2, 50, "t2i"), (1, 3, 387, "t2i"), (2, 1, 130, "causal"), (2, 2, 200, "mmu")])
def test_attention_backward_vs_autograd(B, nH, Lq, kind):
    """dQ, dK, dV of the fused attention vs autograd of softmax(Q K^T + mask) V on the same bf16-rounded operands.
    Tolerance: P, dS and the outputs are rounded to bf16 once each -> 2^-7 of the gradient scale."""
    torch.manual_seed(Lq + B)
    d = util.tiny_dims()
    Lp = (Lq + 63) // 64 * 64
    H = nH * 64
    q = bf16_round(torch.randn(B, nH, Lq, 64) * 0.4)   # pre-scaled Q as the forward path stores it
    k = bf16_round(torch.randn(B, nH, Lq, 64))
    v = bf16_round(torch.randn(B, nH, Lq, 64))
    do = bf16_round(torch.randn(B, Lq, H))
    if kind == "causal":
        mask = torch.where(torch.tril(torch.ones(Lq, Lq, dtype=torch.bool)), 0.0, O.NEG_MASK)[None, None].repeat(B, 1, 1, 1)
    elif kind == "mmu":
        ids = torch.full((B, Lq), 5)
        ids[:, Lq // 3] = d.eoi_id
        mask = O.mask_mmu(ids, d.eoi_id)
    else:
        rows = []
        for i in range(B):
            npad = 3 + 4 * i
            ntext = Lq // 3 - npad
            rows.append([d.pad_id] * npad + [7] * ntext + [d.soi_id] + [d.mask_token_id] * (Lq - npad - ntext - 2) + [d.eoi_id])
        mask = O.mask_t2i(torch.tensor(rows), d.pad_id, d.soi_id, d.eoi_id)
    # reference: autograd in fp64
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    s = qr @ kr.transpose(-1, -2) + mask.double()
    p = torch.softmax(s, -1)
    o = (p @ vr).transpose(1, 2).reshape(B, Lq, H)
    o.backward(do.double())
    # device inputs
    Qd, Kd = _bits(q), _bits(k)
    Vt = torch.zeros((B, nH, 64, Lp), dtype=torch.int16, device="cuda")
    Vt[..., :Lq] = _bits(v.transpose(2, 3).contiguous())
    qkv_v = _bits(v.transpose(1, 2).reshape(B * Lq, H).contiguous())  # V rows token-major (stride H)
    md = dev(mask)
    iv = torch.zeros((B, Lq, 4), dtype=torch.int32, device="cuda")
    flag = torch.zeros(4, dtype=torch.int32, device="cuda")
    L().call("showo_mask_compress", L().ptr(md), L().ptr(iv), L().ptr(flag), B, Lq, Lq, S())
    Od = torch.zeros((B * Lq, H), dtype=torch.int16, device="cuda")
    lse = torch.zeros((B, nH, Lq), dtype=torch.float32, device="cuda")
    L().call("showo_attn_fwd_lse", L().ptr(Qd), L().ptr(Kd), L().ptr(Vt), L().ptr(iv), L().ptr(flag), None, L().ptr(Od), L().ptr(lse),
             B, nH, Lq, Lq, Lq, Lp, H, S())
    want_lse = torch.logsumexp(s.detach(), -1)
    assert (lse.cpu().double() - want_lse).abs().max() < 2e-2
    assert (from_bf16_bits(Od).cpu().double() - o.detach().reshape(B * Lq, H)).abs().max() < 2.5 * 2 ** -8 * float(o.abs().max()) + 1e-3
    QT = torch.zeros((B, nH, 64, Lp), dtype=torch.int16, device="cuda")
    KT = torch.zeros_like(QT)
    dOT = torch.zeros_like(QT)
    L().call("showo_head_transpose", L().ptr(Qd), L().ptr(QT), B, nH, Lq, Lp, nH * Lq * 64, Lq * 64, 64, S())
    L().call("showo_head_transpose", L().ptr(Kd), L().ptr(KT), B, nH, Lq, Lp, nH * Lq * 64, Lq * 64, 64, S())
    assert torch.equal(from_bf16_bits(QT).cpu()[..., :Lq], q.transpose(2, 3)) and (from_bf16_bits(QT).cpu()[..., Lq:] == 0).all()
    D = torch.zeros((B, nH, Lq), dtype=torch.float32, device="cuda")
    dQ = torch.zeros((B * Lq, H), dtype=torch.int16, device="cuda")
    dK = torch.zeros_like(dQ)
    dV = torch.zeros_like(dQ)
    L().call("showo_attn_bwd", L().ptr(Qd), L().ptr(Kd), L().ptr(QT), L().ptr(KT), L().ptr(qkv_v), H, L().ptr(Od), L().ptr(_bits(do.reshape(B * Lq, H))),
             H, L().ptr(dOT), L().ptr(lse), L().ptr(D), L().ptr(iv), L().ptr(flag), L().ptr(dQ), H, L().ptr(dK), H, L().ptr(dV), H,
             B, nH, Lq, Lp, S())
    torch.cuda.synchronize()

    def tok(g):  # [B,nH,L,64] -> token-major [B*L, H]
        return g.transpose(1, 2).reshape(B * Lq, H)

    for name, got, want in (("dQ", dQ, tok(qr.grad)), ("dK", dK, tok(kr.grad)), ("dV", dV, tok(vr.grad))):
        g = from_bf16_bits(got).cpu().double()
        err = float((g - want).abs().max())
        scale = float(want.abs().max())
        rms = float((g - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
        print(f"[parity] attention bwd {kind} L={Lq} {name}: max err {err:.3e} / scale {scale:.3e}, rel rms {rms:.3e}")
        assert err < 2 ** -6 * scale + 1e-3 and rms < 1e-2, name
