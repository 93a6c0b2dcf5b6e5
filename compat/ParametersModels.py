from neural_image_compression_amd.entropy import EntropyParameters  # noqa: F401
