"""Host-side placement logic for block-range shards (mirrors shard_worker in csrc/dxtlt_api.cpp).

The transformed buffer is SoA over the WHOLE block array, so a shard's result is not one contiguous span: for each
stream s (offset multiplier `off`, `w` bytes per block) the shard [first, first+count) owns bytes
[off*N + w*first, off*N + w*(first+count)).  Because blocks are independent, a shard transformed as a stand-alone
buffer yields exactly those slices, packed: slice s sits at [off*count, off*count + w*count) of the shard output.
No collective is needed; the "host concat" is one copy per stream per shard."""
from __future__ import annotations

import numpy as np


def scatter_shard_streams(whole_soa: np.ndarray, shard_soa: np.ndarray, total_blocks: int, first: int, count: int,
                          table) -> None:
    """Forward direction: place a stand-alone shard result into the whole transformed buffer."""
    for off, w in table:
        whole_soa[off * total_blocks + w * first: off * total_blocks + w * (first + count)] = \
            shard_soa[off * count: off * count + w * count]


def gather_shard_streams(whole_soa: np.ndarray, total_blocks: int, first: int, count: int, table) -> np.ndarray:
    """Inverse direction: pack a shard's slice of every stream into a stand-alone transformed buffer."""
    block = sum(w for _, w in table)
    out = np.empty(count * block, dtype=np.uint8)
    for off, w in table:
        out[off * count: off * count + w * count] = \
            whole_soa[off * total_blocks + w * first: off * total_blocks + w * (first + count)]
    return out
