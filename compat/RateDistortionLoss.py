from neural_image_compression_amd.loss import rd_loss, vision_rd_loss  # noqa: F401
