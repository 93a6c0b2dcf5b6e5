from neural_image_compression_amd.entropy import ContextModel, MaskedConv2d  # noqa: F401
