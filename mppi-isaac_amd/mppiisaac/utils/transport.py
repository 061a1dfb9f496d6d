"""Wire format of the reference's planner RPC: torch.save blobs
(reference mppiisaac/utils/transport.py:5-14); pinned by tests/golden/transport.json."""
import io

import torch


def torch_to_bytes(t: torch.Tensor) -> bytes:
    buf = io.BytesIO()
    torch.save(t, buf)
    return buf.getvalue()


def bytes_to_torch(b: bytes, map_location=None) -> torch.Tensor:
    """map_location=None is the reference's behaviour (a blob saved from a CUDA tensor comes back on that device).  The
    planner passes "cpu" for the world state it receives: it only needs the numbers, and restoring a [1, 2n] tensor onto
    the GPU just to read it back costs two extra device round trips per control iteration."""
    return torch.load(io.BytesIO(b), map_location=map_location)
