"""MI355X-native (gfx950) implementation of the analysis/synthesis + hyperprior + likelihood
+ rate-distortion hot path of achraf-15/neural_image_compression, behind the reference's own
Python module surface.  The arithmetic lives in liblic_hip.so (hand-written HIP, C ABI in
include/lic.h); there is no CPU fallback."""
import os as _os

# The model overlaps its decoder and latent-side branches on two HIP streams; with an RCCL communicator in
# the process the runtime serialised them unless the hardware-queue count is set explicitly.
# Only effective if this import precedes the first HIP call of the process.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .models import JointAutoregressiveHierarchical, HierarchicalMixtureResidual, ScalableImageCoding  # noqa: F401
from .loss import rd_loss, vision_rd_loss  # noqa: F401
from .optim import FusedAdam  # noqa: F401
from . import _lib  # noqa: F401

__all__ = ["JointAutoregressiveHierarchical", "HierarchicalMixtureResidual", "ScalableImageCoding", "rd_loss",
           "vision_rd_loss", "FusedAdam"]
