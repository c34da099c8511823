"""Checkpoint format I/O: the on-disk layout of the reference's `ModelMixin` / `ConfigMixin`
(models/modeling_utils.py:47-49, 270-414, 416-865; training/train.py:851-889 `checkpoint-N/unwrapped_model/`):

    <dir>/config.json                      {"_class_name": ..., "_diffusers_version": ..., <constructor arguments>}
    <dir>/pytorch_model.safetensors        (safe_serialization=True, the default)   or   <dir>/pytorch_model.bin
    <dir>/pytorch_model-0000i-of-0000n.*   + pytorch_model.*.index.json  when a state dict exceeds max_shard_size

Same method names and arguments (`save_pretrained`, `from_pretrained`, `save_config`, `load_config`).  Loading is
direct-to-device: the module is built on the meta device (no 5.8 GB host initialisation of values that are about to be
overwritten), materialised on the target device, and each tensor file is read straight there (safetensors `device=`).
The state-dict keys and shapes are the reference's, so files written by either side load on the other."""
import json
import os
import re
import warnings

import torch

CONFIG_NAME = "config.json"
WEIGHTS_NAME = "pytorch_model.bin"
SAFETENSORS_WEIGHTS_NAME = "pytorch_model.safetensors"
_FALLBACK_NAMES = ("diffusion_pytorch_model.safetensors", "model.safetensors", "diffusion_pytorch_model.bin")
_DIFFUSERS_VERSION = "0.30.1"  # the reference's pin (requirements.txt:32); written for file compatibility only


def _add_variant(name, variant):
    if variant is None:
        return name
    parts = name.split(".")
    return ".".join(parts[:-1] + [variant] + parts[-1:])


def _parse_size(s):
    if isinstance(s, int):
        return s
    m = re.fullmatch(r"\s*(\d+(?:\.\d+)?)\s*([KMGT]i?B)\s*", str(s), flags=re.I)
    if not m:
        raise ValueError(f"max_shard_size: cannot parse {s!r}")
    unit = m.group(2).upper()
    mult = {"K": 10 ** 3, "M": 10 ** 6, "G": 10 ** 9, "T": 10 ** 12}[unit[0]] if "I" not in unit else \
        {"K": 2 ** 10, "M": 2 ** 20, "G": 2 ** 30, "T": 2 ** 40}[unit[0]]
    return int(float(m.group(1)) * mult)


def _save_file(sd, path, safe, save_function):
    if safe:
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in sd.items()}, path, metadata={"format": "pt"})
    else:
        (save_function or torch.save)(sd, path)


def _load_file(path, device):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device=str(device))
    return torch.load(path, map_location=device, weights_only=True)


class PretrainedMixin:
    """mix into an nn.Module whose constructor arguments are listed in `_config_keys`"""
    config_name = CONFIG_NAME
    _config_keys = ()

    # ---- config ---------------------------------------------------------------------------------------------------
    def _config_dict(self):
        cfg = getattr(self, "config", {})
        out = {"_class_name": type(self).__name__, "_diffusers_version": _DIFFUSERS_VERSION}
        out.update({k: cfg[k] for k in self._config_keys if k in cfg})
        extra = self._extra_config()
        if extra:
            out["_extra"] = extra  # underscore keys are private to the writer: the reference's ConfigMixin skips them
        return out

    def _extra_config(self):
        """constructor arguments outside the reference's config (e.g. a non-default Phi geometry of a test model)"""
        return {}

    def save_config(self, save_directory, push_to_hub=False, **kwargs):
        if os.path.isfile(save_directory):
            raise AssertionError(f"Provided path ({save_directory}) should be a directory, not a file")
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, self.config_name), "w", encoding="utf-8") as f:
            f.write(json.dumps(self._config_dict(), indent=2, sort_keys=True) + "\n")

    @classmethod
    def load_config(cls, pretrained_model_name_or_path, subfolder=None, **kwargs):
        d = pretrained_model_name_or_path if subfolder is None else os.path.join(pretrained_model_name_or_path, subfolder)
        path = d if os.path.isfile(d) else os.path.join(d, cls.config_name)
        if not os.path.isfile(path):
            raise EnvironmentError(f"{path} not found (hub downloads are not supported: pass a local directory)")
        with open(path, encoding="utf-8") as f:
            return json.load(f)

    # ---- weights --------------------------------------------------------------------------------------------------
    def save_pretrained(self, save_directory, is_main_process=True, save_function=None, safe_serialization=True, variant=None,
                        max_shard_size="10GB", push_to_hub=False, state_dict=None, **kwargs):
        if push_to_hub:
            raise NotImplementedError("no network: push_to_hub is not supported")
        if os.path.isfile(save_directory):
            warnings.warn(f"Provided path ({save_directory}) should be a directory, not a file")
            return
        os.makedirs(save_directory, exist_ok=True)
        if is_main_process:
            self.save_config(save_directory)
        sd = state_dict if state_dict is not None else self.state_dict()
        sd = {k: v.detach() for k, v in sd.items()}
        name = _add_variant(SAFETENSORS_WEIGHTS_NAME if safe_serialization else WEIGHTS_NAME, variant)
        stem, ext = name.split(".", 1)
        limit = _parse_size(max_shard_size)
        shards, cur, cur_bytes = [], {}, 0
        for k, v in sd.items():
            nb = v.numel() * v.element_size()
            if cur and cur_bytes + nb > limit:
                shards.append(cur)
                cur, cur_bytes = {}, 0
            cur[k] = v
            cur_bytes += nb
        shards.append(cur)
        if not is_main_process:
            return
        # stale files of a previous save with the same stem are removed like the reference does (:362-378)
        for fn in os.listdir(save_directory):
            if re.fullmatch(re.escape(stem) + r"(-\d{5}-of-\d{5})?\." + re.escape(ext) + r"(\.index\.json)?", fn):
                os.remove(os.path.join(save_directory, fn))
        if len(shards) == 1:
            _save_file(shards[0], os.path.join(save_directory, name), safe_serialization, save_function)
            return
        weight_map, total = {}, 0
        for i, sh in enumerate(shards):
            fn = f"{stem}-{i + 1:05d}-of-{len(shards):05d}.{ext}"
            _save_file(sh, os.path.join(save_directory, fn), safe_serialization, save_function)
            for k, v in sh.items():
                weight_map[k] = fn
                total += v.numel() * v.element_size()
        with open(os.path.join(save_directory, name + ".index.json"), "w", encoding="utf-8") as f:
            f.write(json.dumps({"metadata": {"total_size": total}, "weight_map": weight_map}, indent=2, sort_keys=True) + "\n")

    @classmethod
    def _weight_files(cls, d, variant, use_safetensors):
        order = []
        if use_safetensors is not False:
            order.append(_add_variant(SAFETENSORS_WEIGHTS_NAME, variant))
        if use_safetensors is not True:
            order.append(_add_variant(WEIGHTS_NAME, variant))
        order += [n for n in _FALLBACK_NAMES if (use_safetensors is not False or not n.endswith(".safetensors"))]
        for name in order:
            if os.path.isfile(os.path.join(d, name)):
                return [os.path.join(d, name)]
            idx = os.path.join(d, name + ".index.json")
            if os.path.isfile(idx):
                with open(idx, encoding="utf-8") as f:
                    files = sorted(set(json.load(f)["weight_map"].values()))
                return [os.path.join(d, fn) for fn in files]
        raise EnvironmentError(f"no weight file ({', '.join(order)}) found in {d}")

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, variant=None, use_safetensors=None, device=None,
                        torch_dtype=None, **kwargs):
        """kwargs override constructor arguments of config.json (and carry engine sizing such as max_batch / max_seq)."""
        d = pretrained_model_name_or_path if subfolder is None else os.path.join(pretrained_model_name_or_path, subfolder)
        if not os.path.isdir(d):
            raise EnvironmentError(f"{d} is not a local directory (hub downloads are not supported offline)")
        raw = cls.load_config(d)
        cfg = {k: v for k, v in raw.items() if not k.startswith("_")}
        cfg.update(raw.get("_extra", {}))
        cfg.update(kwargs)
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        with torch.device("meta"):
            model = cls(**cfg)
        model = model.to_empty(device=device)
        if hasattr(model, "_reset_buffers"):
            model._reset_buffers()  # non-persistent buffers are not in checkpoints: recompute them after to_empty
        want = model.state_dict()
        seen = set()
        for path in cls._weight_files(d, variant, use_safetensors):
            part = _load_file(path, device)
            for k, v in part.items():
                if k not in want:
                    continue
                if tuple(v.shape) != tuple(want[k].shape):
                    raise ValueError(f"{k}: checkpoint shape {tuple(v.shape)} != model shape {tuple(want[k].shape)}")
                with torch.no_grad():
                    want[k].copy_(v)
                seen.add(k)
            unexpected = [k for k in part if k not in want]
            if unexpected:
                warnings.warn(f"{os.path.basename(path)}: {len(unexpected)} unexpected keys ignored, e.g. {unexpected[:3]}")
            del part
        missing = [k for k in want if k not in seen]
        if missing:
            raise KeyError(f"checkpoint {d} lacks {len(missing)} keys, e.g. {missing[:5]}")
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        if hasattr(model, "_weights_changed"):
            model._weights_changed = True
        return model.eval()
