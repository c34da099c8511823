"""CPU: host-side logic and the C-ABI surface (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import util
from util import O, Wt


def test_library_exports_every_declared_symbol():
    L = util.lib()
    lib = L.load()
    hdr = open(os.path.join(util.ROOT, "include", "showo_hip.h")).read()
    declared = set(re.findall(r"\b(showo_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"showo_hip"}
    assert len(declared) > 30
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/showo_hip.h but not exported"
    assert set(L.EXPORTED_SYMBOLS) == declared, (set(L.EXPORTED_SYMBOLS) ^ declared)
    assert lib.showo_abi_version() == 1


def test_no_cpu_fallback():
    """the product path must fail loudly without a GPU"""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    d, sd_np = util.tiny_state()
    S = util.pkg().Showo
    m = S(False, d.vocab, d.llm_vocab, codebook_size=d.codebook, num_vq_tokens=d.num_vq_tokens, hidden_size=d.hidden,
          intermediate_size=d.ffn, num_hidden_layers=d.layers, num_attention_heads=d.heads)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 27, dtype=torch.long))
    with pytest.raises(RuntimeError):
        util.pkg().MAGVITv2().decode_code(torch.zeros(1, 16, dtype=torch.long))


def test_product_never_imports_oracle():
    for root, _, files in os.walk(os.path.join(util.ROOT, "show-o_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "showo_oracle" not in src and "ref_loader" not in src and "/root/reference" not in src, f


def test_state_dict_keys_match_reference_layout():
    d, sd_np = util.tiny_state()
    S = util.pkg().Showo
    m = S(False, d.vocab, d.llm_vocab, codebook_size=d.codebook, num_vq_tokens=d.num_vq_tokens, hidden_size=d.hidden,
          intermediate_size=d.ffn, num_hidden_layers=d.layers, num_attention_heads=d.heads)
    assert list(m.state_dict().keys()) == list(sd_np.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == sd_np[k].shape, k
    m.load_state_dict(O.to_torch(sd_np), strict=True)
    assert m.mask_token_id == d.vocab - 1 == m.config.mask_token_id
    # full-size key list = reference's: 24*18 + 5 tensors
    full = Wt.ShowoDims()
    assert full.vocab == 58498 and full.image_offset == 50305 and full.mask_token_id == 58497
    v = util.pkg().MAGVITv2()
    ref_keys = Wt.make_magvit_state  # generator mirrors the reference's names (validated in make_golden.py strict load)
    sd_v = ref_keys(seed=1)
    assert set(v.state_dict().keys()) == set(sd_v.keys())
    for k, t in v.state_dict().items():
        assert tuple(t.shape) == sd_v[k].shape, k


def test_schedule_constants_match_reference_golden():
    g = util.golden("showo_tiny_t2i.npz")
    S = util.pkg().sampling
    ml, tp = S.t2i_step_constants(int(g["steps"]), 16, 1.0, S.get_mask_chedule("cosine"))
    assert np.array_equal(np.array(ml), g["mask_len"])
    assert np.array_equal(np.array(tp, dtype=np.float32), g["temps"].astype(np.float32))
    # last step: cos(pi/2) < 0 in fp32 -> floor -> -1 (reference quirk)
    ml18, tp18 = S.t2i_step_constants(18, 256)
    assert ml18[-1] == -1.0 and tp18[-1] == 0.0
    with pytest.raises(ValueError):
        S.get_mask_chedule("nope")
    for name in ("linear", "pow2", "sigmoid"):
        f = S.get_mask_chedule(name)
        assert 0.0 <= float(f(torch.tensor(0.3))) <= 1.0
