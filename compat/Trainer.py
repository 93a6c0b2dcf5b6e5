from neural_image_compression_amd.trainer import Trainer  # noqa: F401
