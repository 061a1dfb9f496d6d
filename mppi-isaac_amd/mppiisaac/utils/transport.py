"""Wire format of the reference's planner RPC: torch.save blobs
(reference mppiisaac/utils/transport.py:5-14); pinned by tests/golden/transport.json."""
import io

import torch


def torch_to_bytes(t: torch.Tensor) -> bytes:
    buf = io.BytesIO()
    torch.save(t, buf)
    return buf.getvalue()


def bytes_to_torch(b: bytes) -> torch.Tensor:
    return torch.load(io.BytesIO(b))
