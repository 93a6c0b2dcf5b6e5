from neural_image_compression_amd.entropy import (EntropyModel, FactorizedEntropyBottleneck,  # noqa: F401
                                                  GaussianConditional, GaussianMixtureConditional)
