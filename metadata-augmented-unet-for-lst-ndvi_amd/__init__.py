"""MI355X-native (gfx950) implementation of the Metadata-Augmented U-Net hot path.

Importable as ``mau_amd`` through the alias module at the repository root (the directory name
contains hyphens).  Importing loads ``libmau_hip.so`` and fails loudly when it is missing.
"""
from . import _lib                      # noqa: F401  (raises if the HIP library is absent)
from .model import (MetadataEncoder, TemporalEncoder, UrbanPredictor, UrbanPredictor_unet,  # noqa: F401
                    UrbanPredictor_unetpp, VGGBlock)
from .losses import (compute_loss_l1_grad_ssim, compute_loss_mse, compute_loss_mse_gradient,   # noqa: F401
                     gradient_loss)
from .inference import GraphedInference  # noqa: F401
from .train_graph import GraphedTrainStep  # noqa: F401
from .optim import AdamW  # noqa: F401
from .functional import mark_params_updated  # noqa: F401
from . import data                      # noqa: F401  (input pipeline: compact tiles, device-side one-hot + RandomFlip)

__all__ = ["UrbanPredictor", "UrbanPredictor_unet", "UrbanPredictor_unetpp", "VGGBlock", "MetadataEncoder",
           "TemporalEncoder", "compute_loss_mse", "compute_loss_mse_gradient", "compute_loss_l1_grad_ssim", "gradient_loss",
           "GraphedInference", "GraphedTrainStep", "AdamW", "mark_params_updated"]
