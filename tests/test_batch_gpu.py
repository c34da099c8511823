"""GPU: training-batch construction on the device -- mask_or_random_replace_tokens against the
reference's own outputs (tests/golden/prompting.npz, made by oracle/make_golden.py) and the oracle restatement, and the
mixed t2i + lm + mmu batch with on-device visibility intervals driving the training forward / backward."""
import random

import numpy as np
import pytest
import torch

import util
from util import O, dev

pytestmark = pytest.mark.gpu


class _Sec(dict):
    __getattr__ = dict.__getitem__


def _cfg(**tr):
    return type("Cfg", (), {"training": _Sec(tr)})


def _fn():
    return util.pkg().training_utils.mask_or_random_replace_tokens


def _sched():
    return util.pkg().cosine_schedule


@pytest.mark.parametrize("name,tr", [("mlm", dict(min_masking_rate=0.0)),
                                     ("mlm_minrate", dict(min_masking_rate=0.6)),
                                     ("mlm_all", dict(min_masking_rate=0.0, predict_all_tokens=True)),
                                     ("mlm_rr", dict(min_masking_rate=0.0, noise_type="random_replace"))])
def test_mask_tokens_equals_reference_on_its_own_draws(name, tr):
    g = util.golden("prompting.npz")
    tokens, mask_id = dev(g["mlm_tokens"]), int(g["mlm_mask_id"])
    noise = (dev(g[f"{name}_draw0"]), dev(g[f"{name}_draw1"]))
    inp, lab, lw, mp = _fn()(tokens, mask_id, _cfg(**tr), _sched(), noise=noise)
    assert inp.dtype == torch.int64 and np.array_equal(inp.cpu().numpy(), g[f"{name}_in"])   # bit-exact (index work)
    assert np.array_equal(lab.cpu().numpy(), g[f"{name}_lab"])
    assert np.allclose(mp.cpu().numpy(), g[f"{name}_prob"], atol=1e-6)
    if f"{name}_lw" in g.files:
        assert np.allclose(lw.cpu().numpy(), g[f"{name}_lw"], atol=1e-6)
    else:
        assert lw is None


def test_mask_tokens_contiguous_region_and_eval_ratios_follow_the_host_rng_like_the_reference():
    g = util.golden("prompting.npz")
    tokens, mask_id = dev(g["mlm_tokens"]), int(g["mlm_mask_id"])
    random.seed(32)
    inp, lab, lw, mp = _fn()(tokens, mask_id, _cfg(min_masking_rate=0.1, mask_contiguous_region_prob=1.0), _sched(),
                             noise=(dev(g["mlm_rect_draw0"]), None))
    assert np.array_equal(inp.cpu().numpy(), g["mlm_rect_in"]) and np.array_equal(lab.cpu().numpy(), g["mlm_rect_lab"])
    m = (inp == mask_id).view(-1, 8, 8).cpu()
    for b in range(m.shape[0]):  # every row masks one full rectangle
        ys, xs = torch.nonzero(m[b], as_tuple=True)
        assert int(m[b].sum()) == (int(ys.max()) - int(ys.min()) + 1) * (int(xs.max()) - int(xs.min()) + 1)
    random.seed(32)
    torch.manual_seed(31)
    inp, lab, lw, mp = _fn()(tokens, mask_id, _cfg(min_masking_rate=0.0, eval_mask_ratios=[0.25, 0.5, 0.9]), _sched(), is_train=False)
    assert np.allclose(mp.cpu().numpy(), g["mlm_eval_prob"])
    assert ((inp == mask_id).sum(1).cpu().numpy() == np.rint(64 * g["mlm_eval_prob"])).all()


def test_mask_tokens_default_rng_order_large_rows_and_ties():
    """without injected noise the device generator is consumed like the reference does (rand(B), then rand(B, N)); 1024-token
    rows (512x512 images) and exactly-equal noise values (stable order) against the oracle"""
    B, N, mask_id = 12, 1024, 58497
    tokens = torch.randint(50305, 58497, (B, N), device="cuda")
    torch.manual_seed(77)
    inp, lab, lw, mp = _fn()(tokens, mask_id, _cfg(min_masking_rate=0.0), _sched())
    torch.manual_seed(77)
    t = torch.rand(B, device="cuda")
    u = torch.rand(B, N, device="cuda")
    p = util.pkg().cosine_schedule(t).clip(0.0)
    assert torch.equal(mp, p)
    i2, l2, m2 = O.mask_tokens_np(tokens.cpu().numpy(), u.cpu().numpy(), p.cpu().numpy(), mask_id)
    assert np.array_equal(inp.cpu().numpy(), i2) and np.array_equal(lab.cpu().numpy(), l2)
    # ties: quantised noise has many equal values; both sides break them by index
    uq = (u * 16).floor() / 16
    inp, lab, _, mp = _fn()(tokens, mask_id, _cfg(min_masking_rate=0.0), _sched(), noise=(t, uq))
    i2, l2, _ = O.mask_tokens_np(tokens.cpu().numpy(), uq.cpu().numpy(), mp.cpu().numpy(), mask_id)
    assert np.array_equal(inp.cpu().numpy(), i2) and np.array_equal(lab.cpu().numpy(), l2)
    # every row masks exactly round(N p) >= 1 positions, the rest keep their token and get label -100
    k = (inp == mask_id).sum(1)
    assert torch.equal(k, (N * mp).round().clamp(min=1).long())
    assert ((lab == -100) == (inp != mask_id)).all()
    with pytest.raises(RuntimeError):
        _fn()(tokens.cpu(), mask_id, _cfg(min_masking_rate=0.0), _sched())  # no CPU path


def test_build_training_batch_intervals_drive_the_training_step_like_dense_reference_masks():
    from stub_tokenizer import StubTokenizer
    P = util.pkg()
    d, sd = util.tiny_state()
    up = P.UniversalPrompting(StubTokenizer(vocab=d.llm_vocab), max_text_len=d.max_text_len, cond_dropout_prob=0.3)
    sp = {k: int(v) for k, v in up.sptids_dict.items()}
    assert sp["<|pad|>"] == d.pad_id and sp["<|soi|>"] == d.soi_id and sp["<|eoi|>"] == d.eoi_id
    N = d.num_vq_tokens
    torch.manual_seed(3)
    img_t2i = torch.randint(0, d.codebook, (3, N), device="cuda") + d.image_offset
    img_mmu = torch.randint(0, d.codebook, (2, N), device="cuda") + d.image_offset
    texts_t2i = ["a red cube", "", "one two three four five six seven eight nine ten eleven"]
    texts_lm = ["plain text only " * 3, "short"]
    texts_mmu = ["what is this ? a cube", "x"]
    cfg = _cfg(min_masking_rate=0.0)
    ids, labels, mask, mask_prob, (b1, b2, b3) = P.training_utils.build_training_batch(
        up, cfg, d.mask_token_id, P.cosine_schedule, img_t2i, texts_t2i, texts_lm, img_mmu, texts_mmu)
    L = d.max_text_len + 1 + 1 + N + 1
    assert (b1, b2, b3) == (3, 2, 2) and tuple(ids.shape) == (7, L) == tuple(labels.shape) and mask.shape == (7, 1, L, L)
    # t2i rows: masked positions carry the mask id in the inputs and the original token in the labels
    img_in, img_lab = ids[:3, -(N + 1):-1], labels[:3, -(N + 1):-1]
    assert ((img_in == d.mask_token_id) == (img_lab != -100)).all() and (img_lab[img_lab != -100] == img_t2i[img_lab != -100]).all()
    assert (labels[ids == d.pad_id] == -100).all() and (labels[3:5][ids[3:5] != d.pad_id] == ids[3:5][ids[3:5] != d.pad_id]).all()
    assert (labels[5:, :N + 3] == -100).all()
    # the intervals are the reference's three dense masks, row block by row block (train.py:522-577)
    c = ids.cpu()
    dense = torch.cat([O.mask_t2i(c[:3], d.pad_id, d.soi_id, d.eoi_id, rm_pad_in_image=True),
                       O.mask_t2i(c[3:5], d.pad_id, d.soi_id, d.eoi_id, rm_pad_in_image=False),
                       O.mask_mmu(c[5:], d.eoi_id)], dim=0)
    iv = mask.check().iv.cpu()
    col = torch.arange(L)[None, None, :]
    rec = ((col >= iv[..., 0:1]) & (col < iv[..., 1:2])) | ((col >= iv[..., 2:3]) & (col < iv[..., 3:4]))
    assert torch.equal(rec, dense[:, 0] == 0)
    # and a training step through them == the same step through the dense masks (forward bits, losses, gradients)
    m = util.build_showo(d, sd).train()
    kw = dict(labels=labels, batch_size_t2i=b1, batch_size_lm=b2, batch_size_mmu=b3, max_seq_length=d.max_text_len)
    outs = []
    for am in (mask, dense.cuda()):
        m.zero_grad()
        logits, l1, l2, l3 = m(ids, attention_mask=am, **kw)
        (1.0 * l1 + 0.1 * l2 + 1.0 * l3).backward()
        outs.append((logits.clone(), torch.stack([l1, l2, l3]).clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[0][1]).all() and len(outs[0][2]) > 10
    for n, gr in outs[0][2].items():
        assert torch.equal(gr, outs[1][2][n]), n
    # native trainer takes the same object
    tr = P.Trainer(m)
    losses = tr.step(ids, mask, labels, b1, b2, b3, d.max_text_len)
    assert torch.isfinite(losses).all()


def test_trainer_poisons_losses_when_intervals_cannot_represent_the_mask():
    """pads scattered inside the caption need three visibility runs per row: the builder raises its flag, and a training
    forward driven by those intervals returns NaN losses (device-side check, no host sync) instead of silently wrong ones"""
    P = util.pkg()
    d, sd = util.tiny_state()
    m = util.build_showo(d, sd).train()
    L = d.max_text_len + 1 + 1 + d.num_vq_tokens + 1
    ids = torch.randint(0, 200, (2, L), device="cuda")
    ids[:, d.max_text_len + 1] = d.soi_id
    ids[:, -1] = d.eoi_id
    ids[0, 1] = d.pad_id
    ids[0, 3] = d.pad_id
    labels = ids.clone()
    mask = P.prompting_utils.intervals_predict_next(ids, pad_id=d.pad_id, soi_id=d.soi_id, eoi_id=d.eoi_id, rm_pad_in_image=True)
    assert int(mask.flag[0]) != 0
    with pytest.raises(ValueError):
        mask.check()
    _, l1, l2, l3 = m(ids, attention_mask=mask, labels=labels, batch_size_t2i=2, batch_size_lm=0, batch_size_mmu=0,
                      max_seq_length=d.max_text_len)
    assert torch.isnan(l1)
    dense = P.prompting_utils.create_attention_mask_predict_next(ids, pad_id=d.pad_id, soi_id=d.soi_id, eoi_id=d.eoi_id,
                                                                 rm_pad_in_image=True)
    _, l1, l2, l3 = m(ids, attention_mask=dense, labels=labels, batch_size_t2i=2, batch_size_lm=0, batch_size_mmu=0,
                      max_seq_length=d.max_text_len)
    assert torch.isnan(l1)  # same for the dense form of such a mask: the interval backward could not honour it
    ok = ids.clone()
    ok[0, 1], ok[0, 3] = 7, 8
    dense = P.prompting_utils.create_attention_mask_predict_next(ok, pad_id=d.pad_id, soi_id=d.soi_id, eoi_id=d.eoi_id,
                                                                 rm_pad_in_image=True)
    _, l1, l2, l3 = m(ok, attention_mask=dense, labels=labels, batch_size_t2i=2, batch_size_lm=0, batch_size_mmu=0,
                      max_seq_length=d.max_text_len)
    assert torch.isfinite(l1)
