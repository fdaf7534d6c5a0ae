from .data_loader import (KJJ0DataLoader, SyntheticTokenLoader, download_fineweb10B_files,
                          make_synthetic_fineweb_files)
from .distributed_data_loader import DistributedKJJ0DataLoader
from .shards import open_shard, read_header, write_shard

__all__ = ["KJJ0DataLoader", "SyntheticTokenLoader", "DistributedKJJ0DataLoader",
           "download_fineweb10B_files", "make_synthetic_fineweb_files", "open_shard", "read_header",
           "write_shard"]
