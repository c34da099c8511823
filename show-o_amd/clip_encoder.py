"""CLIP ViT vision tower of the w_clip_vit understanding path -- drop-in for the reference's `models/clip_encoder.py`
(`CLIPVisionTower`, :6-88; used by inference_mmu.py:60-62,133 and training/train_w_clip_vit.py:530-580), SURVEY.md §8f row 2.

Same constructor argument, attributes (`vision_tower_name`, `select_layer = -2`, `select_feature = 'patch'`, `is_loaded`,
`image_processor`), properties (`dtype`, `device`, `config`, `hidden_size`, `num_patches`, `num_patches_per_side`,
`dummy_feature`) and `forward(images)` -> `hidden_states[-2][:, 1:]`.  The reference delegates the arithmetic to
transformers' `CLIPVisionModel`; here it runs on the gfx950 engine of `csrc/clip_engine.hip` (only the 23 layers the selected
feature needs).  `self.vision_tower` holds the frozen parameters under transformers' state-dict names
(`vision_model.embeddings.patch_embedding.weight`, ...), so checkpoints of `openai/clip-vit-large-patch14-336` load
unchanged -- from a LOCAL directory (config.json + model.safetensors / pytorch_model.bin); there is no hub access here.
"""
import ctypes as C
import json
import os

import torch
import torch.nn as nn

from . import _lib
from .persistence import _load_file

# openai/clip-vit-large-patch14-336 vision_config (the tower the reference's configs name: configs/showo_demo_w_clip_vit*.yaml)
CLIP_VIT_L_14_336 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336,
                         patch_size=14, layer_norm_eps=1e-5, hidden_act="quick_gelu", num_channels=3)


class _Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


class _Node(nn.Module):
    """anonymous container: lets a flat {dotted key: tensor} spec become a module tree with exactly those state-dict keys"""


def _set_param(root, key, tensor):
    parts = key.split(".")
    node = root
    for p in parts[:-1]:
        if not hasattr(node, p):
            node.add_module(p, _Node())
        node = getattr(node, p)
    node.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


def vision_state_spec(cfg):
    """transformers' CLIPVisionModel state-dict keys and shapes (4.41 naming, the checkpoint format)"""
    H, F, S, ps = cfg["hidden_size"], cfg["intermediate_size"], cfg["image_size"], cfg["patch_size"]
    spec = {"vision_model.embeddings.class_embedding": (H,),
            "vision_model.embeddings.patch_embedding.weight": (H, cfg.get("num_channels", 3), ps, ps),
            "vision_model.embeddings.position_embedding.weight": ((S // ps) ** 2 + 1, H),
            "vision_model.pre_layrnorm.weight": (H,), "vision_model.pre_layrnorm.bias": (H,)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"vision_model.encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            spec[p + f"self_attn.{nm}.weight"], spec[p + f"self_attn.{nm}.bias"] = (H, H), (H,)
        spec[p + "layer_norm1.weight"], spec[p + "layer_norm1.bias"] = (H,), (H,)
        spec[p + "mlp.fc1.weight"], spec[p + "mlp.fc1.bias"] = (F, H), (F,)
        spec[p + "mlp.fc2.weight"], spec[p + "mlp.fc2.bias"] = (H, F), (H,)
        spec[p + "layer_norm2.weight"], spec[p + "layer_norm2.bias"] = (H,), (H,)
    spec["vision_model.post_layernorm.weight"], spec["vision_model.post_layernorm.bias"] = (H,), (H,)
    return spec


def _canonical(key):
    """checkpoint / transformers>=5 key -> 4.41 vision key, or None for tensors that are not part of the vision tower"""
    if key.startswith("vision_tower."):
        key = key[len("vision_tower."):]
    if key.split(".")[0] in ("embeddings", "encoder", "pre_layrnorm", "post_layernorm"):
        key = "vision_model." + key
    if not key.startswith("vision_model.") or key.endswith("position_ids"):
        return None
    return key


class CLIPVisionTower(nn.Module):
    def __init__(self, vision_tower, config=None, state_dict=None, max_batch=4):
        """vision_tower: local checkpoint directory (what the reference passes to `from_pretrained`).  `config` (dict of
        CLIPVisionConfig fields) and `state_dict` may be given instead of files (tests, synthetic benchmarks)."""
        super().__init__()
        self.is_loaded = False
        self.vision_tower_name = vision_tower
        self.select_layer = -2
        self.select_feature = "patch"
        self.max_batch = int(max_batch)
        self._given = (config, state_dict)
        self._clip, self._versions = None, None
        self.image_processor = None
        self.load_model()
        self.cfg_only = self._cfg

    # ---- loading ----------------------------------------------------------------------------------------------------
    def _read_config(self):
        cfg, _ = self._given
        if cfg is None:
            path = os.path.join(str(self.vision_tower_name), "config.json")
            if not os.path.isfile(path):
                raise EnvironmentError(f"{path} not found: pass a local checkpoint directory (no hub access) or config=...")
            raw = json.load(open(path, encoding="utf-8"))
            cfg = raw.get("vision_config", raw)  # a full CLIPConfig nests the vision part
        out = dict(CLIP_VIT_L_14_336)
        out.update({k: cfg[k] for k in out if k in cfg})
        if out["hidden_act"] != "quick_gelu":
            raise NotImplementedError(f"hidden_act={out['hidden_act']}: the engine implements CLIP's quick_gelu")
        if out["hidden_size"] != 64 * out["num_attention_heads"]:
            raise NotImplementedError("the gfx950 attention kernel is built for head_dim 64 (CLIP ViT-L/14: 1024 / 16)")
        return _Cfg(out)

    def load_model(self, device_map=None):
        if self.is_loaded:
            print('{} is already loaded, `load_model` called again, skipping.'.format(self.vision_tower_name))
            return
        self._cfg = self._read_config()
        spec = vision_state_spec(self._cfg)
        _, sd = self._given
        if sd is None:
            d = str(self.vision_tower_name)
            names = ("model.safetensors", "pytorch_model.safetensors", "pytorch_model.bin")
            path = next((os.path.join(d, n) for n in names if os.path.isfile(os.path.join(d, n))), None)
            if path is None:
                raise EnvironmentError(f"no weight file ({', '.join(names)}) in {d}")
            sd = _load_file(path, "cpu")
        got = {}
        for k, v in sd.items():
            ck = _canonical(k)
            if ck in spec:
                if tuple(v.shape) != tuple(spec[ck]):
                    raise ValueError(f"{k}: shape {tuple(v.shape)} != {tuple(spec[ck])}")
                got[ck] = v
        missing = [k for k in spec if k not in got]
        if missing:
            raise KeyError(f"CLIP vision checkpoint lacks {len(missing)} tensors, e.g. {missing[:4]}")
        self.vision_tower = _Node()
        for k in spec:  # spec order = state-dict order
            _set_param(self.vision_tower, k, torch.as_tensor(got[k]).detach().clone().float())
        self.vision_tower.requires_grad_(False)
        try:  # host-side preprocessing (resize / crop / normalise) stays transformers' CLIPImageProcessor, as in the reference
            if os.path.isfile(os.path.join(str(self.vision_tower_name), "preprocessor_config.json")):
                from transformers import CLIPImageProcessor
                self.image_processor = CLIPImageProcessor.from_pretrained(str(self.vision_tower_name))
        except Exception:  # pragma: no cover - optional dependency pieces missing
            self.image_processor = None
        self.is_loaded = True

    # ---- engine -------------------------------------------------------------------------------------------------------
    def _drop(self):
        if getattr(self, "_clip", None) is not None:
            _lib.load().showo_clip_destroy(self._clip)
        self._clip, self._versions = None, None

    def __del__(self):
        try:
            self._drop()
        except Exception:
            pass

    def mark_weights_dirty(self):
        """re-upload every parameter at the next call (needed after updates through `.data`, which do not bump tensor versions)"""
        if self._clip is not None:
            self._versions = {}

    def set_precision(self, precision):
        """0 (default): bf16 GEMM / attention operands with fp32 accumulation.  1: accuracy mode -- split-bf16 (hi + lo) MFMA GEMMs,
        fp32 LayerNorm / attention / quick_gelu (csrc/clip_engine.hip, csrc/precise.hip): the fp32 tower the reference runs
        (models/clip_encoder.py:29-49) to ~1e-4.  Inference only; returns self."""
        if int(precision) not in (0, 1):
            raise ValueError("precision must be 0 (bf16 operands) or 1 (split-bf16, fp32-class)")
        self._precision = int(precision)
        if self._clip is not None:
            _lib.call("showo_clip_set_precision", self._clip, self._precision)
            if self._precision == 1 and not _lib.load().showo_clip_precise_ready(self._clip):
                self._versions = {}  # the low halves are made by the loader: upload everything again at the next call
        return self

    def engine(self, batch=1):
        _lib.require_gpu()
        lib = _lib.load()
        if self.device.type != "cuda":
            raise RuntimeError("CLIPVisionTower parameters must live on the GPU (tower.to('cuda')); no CPU path exists")
        if self._clip is not None and batch > self.max_batch:
            self.max_batch = int(batch)
            self._drop()
        if self._clip is None:
            c = self._cfg
            cfg = _lib.ClipConfig()
            cfg.image_size, cfg.patch_size, cfg.hidden, cfg.heads = c.image_size, c.patch_size, c.hidden_size, c.num_attention_heads
            cfg.ffn, cfg.layers = c.intermediate_size, c.num_hidden_layers
            cfg.run_layers = c.num_hidden_layers + 1 + self.select_layer  # hidden_states[select_layer] = output of that many layers
            cfg.max_batch, cfg.ln_eps = max(self.max_batch, batch), c.layer_norm_eps
            self.max_batch = cfg.max_batch
            h = C.c_void_p()
            _lib.check(lib.showo_clip_create(C.byref(cfg), C.byref(h)), "showo_clip_create")
            self._clip, self._versions = h, {}
            _lib.call("showo_clip_set_precision", self._clip, int(getattr(self, "_precision", 0)))
        for k, v in self.vision_tower.state_dict().items():
            ver = (v.data_ptr(), v._version)
            if self._versions.get(k) != ver:
                src = v.detach()
                if src.dtype != torch.float32 or not src.is_contiguous():
                    src = src.float().contiguous()
                _lib.call("showo_clip_load", self._clip, k.encode(), _lib.ptr(src), src.numel(), _lib.stream())
                self._versions[k] = ver
                if src is not v:
                    torch.cuda.current_stream().synchronize()
        missing = lib.showo_clip_missing(self._clip)
        if missing:
            raise RuntimeError(f"CLIP engine is missing {missing} tensors")
        return self._clip

    # ---- reference API ------------------------------------------------------------------------------------------------
    def _features(self, images):
        if images.dim() != 4 or images.shape[1] != 3 or images.shape[2] != self._cfg.image_size or images.shape[3] != self._cfg.image_size:
            raise ValueError(f"images must be [B,3,{self._cfg.image_size},{self._cfg.image_size}], got {tuple(images.shape)}")
        if self.select_feature not in ("patch",):
            raise ValueError(f'Unexpected select feature: {self.select_feature}')
        x = images.to(device=self.device).detach().float().contiguous()
        B = x.shape[0]
        eng = self.engine(B)
        out = torch.empty((B, self.num_patches, self.hidden_size), dtype=torch.float32, device=x.device)
        _lib.call("showo_clip_features", eng, _lib.ptr(x), B, _lib.ptr(out), _lib.stream())
        return out

    @torch.no_grad()
    def forward(self, images):
        if type(images) is list:
            return [self._features(im.unsqueeze(0)).to(im.dtype) for im in images]
        return self._features(images).to(images.dtype)

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        return self.vision_tower.vision_model.embeddings.class_embedding.dtype

    @property
    def device(self):
        return self.vision_tower.vision_model.embeddings.class_embedding.device

    @property
    def config(self):
        return self._cfg if self.is_loaded else self.cfg_only

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches_per_side(self):
        return self.config.image_size // self.config.patch_size

    @property
    def num_patches(self):
        return (self.config.image_size // self.config.patch_size) ** 2
