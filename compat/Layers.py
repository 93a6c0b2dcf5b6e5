from neural_image_compression_amd.layers import (ResidualBlock, ResidualBlockUpsample,  # noqa: F401
                                                 ResidualBlockWithStride, TransposedDeconv3x3)
