"""MI355X-native UNETR training hot path (gfx950 HIP kernels behind the reference's nn.Module interface).

The directory name starts with a digit, so import it with
``importlib.import_module("3dmedicalimagesegmentation_amd")`` -- or put this directory itself on
``sys.path`` and keep the reference scripts' own ``from unetr import UNETR`` line unchanged.
"""
from .unetr import UNETR, UNETRLogits, default_precision  # noqa: F401
from .losses import DiceCELoss, ranking_loss  # noqa: F401
from .optim import AdamW  # noqa: F401
from .inference import DiceMetric, sliding_window_inference  # noqa: F401
from .train_step import TrainStep  # noqa: F401
from .functional import invalidate_weight_shadows  # noqa: F401
from . import _capi, ddp, functional, inference, train_step  # noqa: F401

__all__ = ["UNETR", "UNETRLogits", "DiceCELoss", "ranking_loss", "AdamW", "default_precision", "sliding_window_inference",
           "DiceMetric", "TrainStep", "invalidate_weight_shadows"]
