from neural_image_compression_amd.evaluator import CompressionEvaluator  # noqa: F401
