"""TEST INFRASTRUCTURE ONLY — loads the *real* reference (showlab/Show-o) from /root/reference.

This file is used only in the build container (where /root/reference is mounted) to
  (1) validate the in-repo CPU restatement `oracle/showo_oracle.py` against the reference itself, and
  (2) generate the committed golden fixtures under tests/golden/ (see oracle/make_golden.py).
Nothing under `show-o_amd/` imports it; nothing here is copied from the reference: it only *imports*
the reference modules after installing import stubs for third-party packages that are not installed
 and a PhiConfig shim for transformers 5.x.
"""
import os
import sys
import types
import importlib

REF_ROOT = os.environ.get("SHOWO_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Sub:
    """Subscriptable placeholder for jaxtyping annotations."""
    def __class_getitem__(cls, item):
        return cls


def _install_stubs():
    if "diffusers" in sys.modules and getattr(sys.modules["diffusers"], "_showo_stub", False):
        return
    import torch
    _mod("omegaconf", OmegaConf=object, DictConfig=dict, ListConfig=list)
    _mod("jaxtyping", **{n: type(n, (_Sub,), {}) for n in
                         ["Bool", "Complex", "Float", "Inexact", "Int", "Integer", "Num", "Shaped", "UInt"]})
    _mod("typeguard", typechecked=lambda f=None, **k: f if f is not None else (lambda g: g))

    class _Logger:
        def __getattr__(self, k):
            return lambda *a, **kw: None

    class _Logging:
        @staticmethod
        def get_logger(name=None):
            return _Logger()

    d = _mod("diffusers", __version__="0.30.1", _showo_stub=True)
    d.__path__ = []
    du = _mod("diffusers.utils",
              FLAX_WEIGHTS_NAME="flax_model.msgpack", SAFE_WEIGHTS_INDEX_NAME="x.safetensors.index.json",
              WEIGHTS_INDEX_NAME="x.bin.index.json", SAFETENSORS_WEIGHTS_NAME="m.safetensors",
              WEIGHTS_NAME="m.bin", CONFIG_NAME="config.json", MIN_PEFT_VERSION="0.0",
              _add_variant=None, _get_checkpoint_shard_files=None, _get_model_file=None,
              deprecate=lambda *a, **k: None, is_accelerate_available=lambda: False,
              is_torch_version=lambda *a, **k: True, logging=_Logging,
              is_bitsandbytes_available=lambda: False, is_bitsandbytes_version=lambda *a, **k: False,
              check_peft_version=lambda *a, **k: None)
    du.__path__ = []
    du.__getattr__ = lambda name: None  # any other name imports as None
    _mod("diffusers.utils.hub_utils", PushToHubMixin=type("PushToHubMixin", (), {}),
         load_or_create_model_card=None, populate_model_card=None)
    dm = _mod("diffusers.models")
    dm.__path__ = []
    _mod("diffusers.models.model_loading_utils", _determine_device_map=None, _fetch_index_file=None,
         _load_state_dict_into_model=None, load_model_dict_into_meta=None, load_state_dict=None)

    class _Cfg(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

    class ConfigMixin:
        def register_to_config(self, **kw):
            if "_internal_dict" not in self.__dict__:
                object.__setattr__(self, "_internal_dict", _Cfg())
            self.__dict__["_internal_dict"].update(kw)

        @property
        def config(self):
            return self.__dict__["_internal_dict"]

    def register_to_config(init):
        import functools
        import inspect

        @functools.wraps(init)
        def wrapper(self, *args, **kwargs):
            sig = inspect.signature(init)
            names = [p for p in list(sig.parameters)[1:] if sig.parameters[p].kind is not inspect.Parameter.VAR_KEYWORD]
            cfg = {n: sig.parameters[n].default for n in names if sig.parameters[n].default is not inspect.Parameter.empty}
            cfg.update(dict(zip(names, args)))
            cfg.update(kwargs)
            if "_internal_dict" not in self.__dict__:
                object.__setattr__(self, "_internal_dict", _Cfg())
            self.__dict__["_internal_dict"].update(cfg)
            init(self, *args, **kwargs)
        return wrapper

    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)


def shimmed_phi_config(**overrides):
    """PhiConfig whose defaults equal microsoft/phi-1_5, with the 4.41-era attributes models/phi.py reads."""
    from transformers import PhiConfig
    cfg = PhiConfig(**overrides)
    cfg.rope_theta = 10000.0
    cfg.rope_scaling = None
    cfg.partial_rotary_factor = 0.5
    cfg._attn_implementation = "sdpa"
    return cfg


_REF = {}


def load_reference():
    """Returns a namespace with the reference's hot-path modules imported from REF_ROOT."""
    if _REF:
        return types.SimpleNamespace(**_REF)
    if not reference_available():
        raise RuntimeError(f"reference not found at {REF_ROOT}")
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # The repo root also has packages; make sure `models` resolves to the reference's.
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[k]
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        phi = importlib.import_module("models.phi")
        showo = importlib.import_module("models.modeling_showo")
        magvit = importlib.import_module("models.modeling_magvitv2")
        sampling = importlib.import_module("models.sampling")
        common = importlib.import_module("models.common_modules")
    import importlib.util as _ilu
    spec = _ilu.spec_from_file_location("ref_prompting_utils",
                                                  os.path.join(REF_ROOT, "training", "prompting_utils.py"))
    pu = _ilu.module_from_spec(spec)
    spec.loader.exec_module(pu)
    _REF.update(phi=phi, showo=showo, magvit=magvit, sampling=sampling, common=common, prompting=pu)
    return types.SimpleNamespace(**_REF)


def load_reference_training_utils():
    """the reference's training/utils.py (mask_or_random_replace_tokens, get_loss_weight); torchvision is absent offline and
    only used by its image transforms, so it is stubbed"""
    load_reference()
    if "train_utils" in _REF:
        return _REF["train_utils"]
    for name in ("torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            sys.modules[name] = _mod(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    import importlib.util as _ilu
    spec = _ilu.spec_from_file_location("ref_training_utils", os.path.join(REF_ROOT, "training", "utils.py"))
    tu = _ilu.module_from_spec(spec)
    spec.loader.exec_module(tu)
    _REF["train_utils"] = tu
    return tu


def build_reference_showo(phi_overrides=None, vocab_size=58498, llm_vocab_size=50295, codebook_size=8192,
                          num_vq_tokens=256, w_clip_vit=False, seed=0):
    """Seeded random-init reference Showo (no pretrained weights are available offline)."""
    import contextlib, io
    import torch
    import transformers
    ref = load_reference()
    cfg = shimmed_phi_config(**(phi_overrides or {}))
    orig = transformers.AutoConfig.from_pretrained
    ref.showo.AutoConfig.from_pretrained = staticmethod(lambda *a, **k: cfg)
    try:
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            m = ref.showo.Showo(w_clip_vit=w_clip_vit, vocab_size=vocab_size, llm_vocab_size=llm_vocab_size,
                                llm_model_path="microsoft/phi-1_5", codebook_size=codebook_size,
                                num_vq_tokens=num_vq_tokens)
    finally:
        ref.showo.AutoConfig.from_pretrained = orig
    return m.eval()


def build_reference_magvit(seed=0):
    import contextlib, io
    import torch
    ref = load_reference()
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref.magvit.MAGVITv2()
    return m.eval()
