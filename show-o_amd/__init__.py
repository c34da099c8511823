"""show-o_amd — MI355X (gfx950) native hot path of showlab/Show-o.

Public names mirror the reference's `models/__init__.py:1-4`:
    Showo, MAGVITv2, get_mask_chedule
All arithmetic runs in hand-written HIP kernels behind the C ABI of `libshowo_hip.so`
(include/showo_hip.h); see DESIGN.md / INTEGRATION.md.
"""
from .sampling import get_mask_chedule, cosine_schedule  # noqa: F401
from .modeling_showo import Showo, gen_config  # noqa: F401
from .modeling_magvitv2 import MAGVITv2  # noqa: F401
from . import _lib  # noqa: F401
from .training import Trainer  # noqa: F401
from . import prompting_utils  # noqa: F401
from . import training_utils  # noqa: F401
from .prompting_utils import UniversalPrompting  # noqa: F401
from .training_utils import mask_or_random_replace_tokens  # noqa: F401
from .clip_encoder import CLIPVisionTower  # noqa: F401
from . import image_utils  # noqa: F401
from .image_utils import image_transform  # noqa: F401
from . import synthetic  # noqa: F401
from . import checkpointing  # noqa: F401
