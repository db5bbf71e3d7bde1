"""Multi-GPU pieces of the path (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The only part of the path that shards is the T-frame ViT encode: frames are independent through the tower
(T is the batch dim, modeling_image.py:641-643; add_time_attn is False for the image tower).  Rank r encodes a
contiguous block of frames and ONE all-gather of the 1024-wide visual tokens (before the 4096-wide projector: 4x
fewer bytes) rebuilds the full [T, 256, Dv] tensor in chronological order on every rank.  The LLM of one
conversation does not shard (no TP/SP in the reference); conversations are data-parallel replicas with no collective.
"""
import torch
import torch.distributed as dist


def frame_partition(T, world_size):
    """Contiguous, order-preserving split of T frames: returns [(start, count)] per rank (counts differ by <= 1)."""
    base, extra = divmod(T, world_size)
    out, s = [], 0
    for r in range(world_size):
        c = base + (1 if r < extra else 0)
        out.append((s, c))
        s += c
    return out


def sharded_frame_features(encode_fn, pixels, group=None):
    """pixels [T,3,H,W] (same on every rank) -> features [T, NV, Dv] on every rank.

    encode_fn(frames[c,3,H,W]) -> [c, NV, Dv] is the local ViT encode (TeoEngine.vit_features on the GPU path).
    Ragged T is handled by padding every rank's block to the largest block (the pad rows are dropped after the gather).
    """
    if group is None and not (dist.is_available() and dist.is_initialized()):
        return encode_fn(pixels)
    ws = dist.get_world_size(group)
    rank = dist.get_rank(group)
    T = pixels.shape[0]
    parts = frame_partition(T, ws)
    cmax = max(c for _, c in parts)
    s, c = parts[rank]
    local = encode_fn(pixels[s:s + c]) if c > 0 else None
    if local is None:
        probe = encode_fn(pixels[:1])               # shape/dtype only (T < world_size)
        local = probe[:0]
    NV, Dv = local.shape[1], local.shape[2]
    send = torch.zeros(cmax, NV, Dv, dtype=local.dtype, device=local.device)
    send[:c] = local
    recv = torch.empty(ws * cmax, NV, Dv, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(ws, cmax, NV, Dv)
    return torch.cat([recv[r, :parts[r][1]] for r in range(ws)], dim=0)


def shard_conversations(n_items, rank, world_size):
    """Conversation-level data parallelism: item indices owned by `rank` (round-robin, no collective on the path)."""
    return list(range(rank, n_items, world_size))
