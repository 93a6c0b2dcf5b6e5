"""Import shim for drivers written against the reference's Dataloader.py: the preprocessed crops come from
uint8 NHWC shards decoded once offline (see neural_image_compression_amd/data.py) instead of per-item PIL."""
from neural_image_compression_amd.data import (ShardDataset, ShardLoader, shard_from_image_files,  # noqa: F401
                                               write_shard)
