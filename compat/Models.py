from neural_image_compression_amd.models import (HierarchicalMixtureResidual, JointAutoregressiveHierarchical,  # noqa: F401
                                                 ScalableImageCoding)
