"""EXECUTES the body of the reference's own, unmodified `inference_t2i.py` (mode t2i) against this package -- the second half of the
acceptance check that `tests/test_reference_callsites.py` starts (which binds every call site statically).

`runpy` runs the script where it lies (/root/reference; never copied) with the bindings INTEGRATION.md section 1 describes:
`models` -> showo_amd (Showo, MAGVITv2, get_mask_chedule), `training.prompting_utils` -> showo_amd.prompting_utils,
`training.utils.image_transform` -> showo_amd.image_utils; and with the control plane stubbed: `get_config` returns a dict namespace
with the keys of configs/showo_demo.yaml the script reads (OmegaConf is not installed), `wandb` records what is logged,
`AutoTokenizer.from_pretrained` returns the deterministic stub tokenizer (no tokenizer files offline), `from_pretrained` of the two
models builds tiny random-weight instances (no checkpoints offline).

Two forms of the same run:
  * CPU (runs in the build container, where the reference is): the three GPU entry points the script reaches -- the mask builder,
    `Showo.t2i_generate`, `MAGVITv2.decode_code` -- are replaced by shape-checking stand-ins (the product has no CPU path by design);
    everything else is the real package: class surface, attributes, `UniversalPrompting`, schedules, argument plumbing.  The stand-in of
    `t2i_generate` binds the script's call against the REAL signature first and checks every tensor's shape / dtype.
  * GPU (`-m gpu`): the same run on the real kernels; it needs a GPU AND the reference tree, which this setup never has on one machine
    (the GPU box has no /root/reference), so it is skipped there and kept for a host that has both."""
import inspect
import math
import os
import runpy
import sys
import types

import numpy as np
import pytest
import torch

import util
from util import O

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")


class NS(dict):
    """dict with attribute access and .get(): what the scripts use of an OmegaConf node"""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _ns(d):
    return NS({k: _ns(v) if isinstance(v, dict) else v for k, v in d.items()})


def _run_inference_t2i(tmp_path, monkeypatch, on_gpu):
    P = util.pkg()
    sys.path.insert(0, os.path.join(util.ROOT, "oracle"))
    from stub_tokenizer import StubTokenizer
    d, sd = util.tiny_state()
    prompts = ["a red cube on a table", "two dogs", "the sea at night", "a", "green hills far away"]
    pf = tmp_path / "prompts.txt"
    pf.write_text("\n".join(prompts))
    batch = 2
    config = _ns({
        "wandb": {"resume": False}, "experiment": {"name": "acceptance"}, "mode": "t2i",
        "model": {"showo": {"llm_model_path": "stub", "pretrained_model_path": "stub", "num_vq_tokens": d.num_vq_tokens,
                            "codebook_size": d.codebook, "llm_vocab_size": d.llm_vocab, "num_new_special_tokens": d.num_new_special_tokens,
                            "w_clip_vit": False},
                  "vq_model": {"type": "magvitv2", "vq_model_name": "stub"}},
        "dataset": {"preprocessing": {"max_seq_length": d.max_text_len}, "params": {"validation_prompts_file": str(pf), "resolution": 64}},
        "training": {"cond_dropout_prob": 0.1, "generation_temperature": 1.0},
        "validation_prompts_file": str(pf), "batch_size": batch, "guidance_scale": 1.75, "generation_timesteps": 3,
    })
    record = {"wandb_init": [], "wandb_log": [], "t2i": [], "decode": [], "mask": []}

    # ---- control plane stubs
    import importlib.machinery

    def module(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)  # importlib.util.find_spec(name) is asked by third-party integrations
        return m

    wandb = module("wandb")
    wandb.util = types.SimpleNamespace(generate_id=lambda: "run0")
    wandb.init = lambda **kw: record["wandb_init"].append(kw)
    wandb.log = lambda data, step=None: record["wandb_log"].append((step, data))
    wandb.Image = lambda image, caption=None: (np.asarray(image), caption)
    monkeypatch.setitem(sys.modules, "wandb", wandb)
    import transformers
    monkeypatch.setattr(transformers.AutoTokenizer, "from_pretrained", staticmethod(lambda *a, **k: StubTokenizer()))

    def flatten(cfg, resolve=True, prefix=""):
        for k, v in cfg.items():
            if isinstance(v, dict):
                yield from flatten(v, resolve, prefix + k + ".")
            else:
                yield prefix + k, v

    # ---- the bindings of INTEGRATION.md section 1
    models = module("models")
    models.Showo, models.MAGVITv2, models.get_mask_chedule = P.Showo, P.MAGVITv2, P.get_mask_chedule
    training = module("training")
    training.__path__ = []
    tpu = module("training.prompting_utils")
    tpu.UniversalPrompting = P.UniversalPrompting
    tpu.create_attention_mask_predict_next = P.prompting_utils.create_attention_mask_predict_next
    tu = module("training.utils")
    tu.get_config = lambda: config
    tu.flatten_omega_conf = flatten
    tu.image_transform = P.image_utils.image_transform
    for name, mod in (("models", models), ("training", training), ("training.prompting_utils", tpu), ("training.utils", tu)):
        monkeypatch.setitem(sys.modules, name, mod)

    # ---- no checkpoints offline: from_pretrained builds tiny random-weight instances
    def showo_from_pretrained(cls, path, *a, **k):
        m = P.Showo(False, d.vocab, d.llm_vocab, codebook_size=d.codebook, num_vq_tokens=d.num_vq_tokens, hidden_size=d.hidden,
                    intermediate_size=d.ffn, num_hidden_layers=d.layers, num_attention_heads=d.heads, max_batch=2 * batch, max_seq=64)
        m.load_state_dict(O.to_torch(sd), strict=True)
        return m

    monkeypatch.setattr(P.Showo, "from_pretrained", classmethod(showo_from_pretrained))
    monkeypatch.setattr(P.MAGVITv2, "from_pretrained", classmethod(lambda cls, name, *a, **k: P.MAGVITv2(ch=32, max_batch=batch, max_res=64)))

    if not on_gpu:
        monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
        real_sig = inspect.signature(P.Showo.t2i_generate)
        L = d.max_text_len + 1 + 1 + d.num_vq_tokens + 1

        def fake_mask(sequence, pad_id=128256, soi_id=128257, eoi_id=128258, rm_pad_in_image=False, return_inverse_mask=True):
            record["mask"].append(tuple(sequence.shape))
            assert sequence.dtype == torch.int64 and sequence.shape[1] == L
            return O.mask_t2i(sequence, pad_id, soi_id, eoi_id, rm_pad_in_image=rm_pad_in_image)

        def fake_t2i(self, *args, **kw):
            b = real_sig.bind(self, *args, **kw)  # the script's call must bind against the REAL signature
            b.apply_defaults()
            a = b.arguments
            ids, unc, am = a["input_ids"], a["uncond_input_ids"], a["attention_mask"]
            B = ids.shape[0]
            assert ids.dtype == torch.int64 and tuple(ids.shape) == (B, L) and tuple(unc.shape) == (B, L)
            assert tuple(am.shape) == (2 * B, 1, L, L) and am.dtype == torch.float32
            assert int((ids == self.mask_token_id).sum()) == B * d.num_vq_tokens  # every image slot starts masked
            assert a["timesteps"] == 3 and a["guidance_scale"] == 1.75 and callable(a["noise_schedule"]) and a["config"] is config
            assert float(a["noise_schedule"](torch.tensor(0.0))) == pytest.approx(1.0)
            record["t2i"].append(B)
            return torch.randint(0, d.codebook, (B, d.num_vq_tokens))

        def fake_decode(self, code, shape=None):
            assert code.dtype == torch.int64 and code.shape[1] == d.num_vq_tokens and int(code.max()) < d.codebook
            record["decode"].append(code.shape[0])
            side = int(math.isqrt(d.num_vq_tokens)) * 16
            return torch.rand(code.shape[0], 3, side, side) * 2 - 1

        tpu.create_attention_mask_predict_next = fake_mask
        monkeypatch.setattr(P.Showo, "t2i_generate", fake_t2i)
        monkeypatch.setattr(P.MAGVITv2, "decode_code", fake_decode)

    monkeypatch.setattr(sys, "argv", ["inference_t2i.py"])
    runpy.run_path(os.path.join(REF, "inference_t2i.py"), run_name="__main__")
    return record, prompts, batch, d


def _check(record, prompts, batch, d):
    nb = math.ceil(len(prompts) / batch)
    assert len(record["wandb_init"]) == 1 and record["wandb_init"][0]["name"] == "acceptance_t2i_t2i"
    assert [s for s, _ in record["wandb_log"]] == list(range(0, len(prompts), batch))
    side = int(math.isqrt(d.num_vq_tokens)) * 16
    n_img = 0
    for step, data in record["wandb_log"]:
        for i, (img, caption) in enumerate(data["generated_images"]):
            assert img.shape == (side, side, 3) and img.dtype == np.uint8 and caption == prompts[step + i]
            n_img += 1
    assert n_img == len(prompts)
    return nb


def test_inference_t2i_script_runs_unchanged_against_this_package_cpu(tmp_path, monkeypatch):
    if torch.cuda.is_available():
        pytest.skip("the GPU form of this test runs the real kernels")
    record, prompts, batch, d = _run_inference_t2i(tmp_path, monkeypatch, on_gpu=False)
    nb = _check(record, prompts, batch, d)
    assert record["t2i"] == [2, 2, 1] and record["decode"] == [2, 2, 1] and len(record["mask"]) == nb  # ragged last batch included
    assert all(s[0] == 2 * b for s, b in zip(record["mask"], record["t2i"]))  # CFG: cond + uncond rows in one mask


@pytest.mark.gpu
def test_inference_t2i_script_runs_unchanged_against_this_package_gpu(tmp_path, monkeypatch):
    """the same run on the real kernels (needs a GPU and the reference tree on one machine)"""
    record, prompts, batch, d = _run_inference_t2i(tmp_path, monkeypatch, on_gpu=True)
    _check(record, prompts, batch, d)
