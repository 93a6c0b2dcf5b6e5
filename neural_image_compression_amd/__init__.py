"""MI355X-native (gfx950) implementation of the analysis/synthesis + hyperprior + likelihood
+ rate-distortion hot path of achraf-15/neural_image_compression, behind the reference's own
Python module surface.  The arithmetic lives in liblic_hip.so (hand-written HIP, C ABI in
include/lic.h); there is no CPU fallback."""
from .models import JointAutoregressiveHierarchical, HierarchicalMixtureResidual  # noqa: F401
from .loss import rd_loss  # noqa: F401
from . import _lib  # noqa: F401

__all__ = ["JointAutoregressiveHierarchical", "HierarchicalMixtureResidual", "rd_loss"]
