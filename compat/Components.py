from neural_image_compression_amd.components import (Decoder3x3, Decoder5x5, Encoder3x3, Encoder5x5,  # noqa: F401
                                                     HyperDecoder3x3, HyperDecoder5x5, HyperEncoder3x3,
                                                     HyperEncoder5x5, LatentSpaceTransform)
