"""Multi-GPU pieces of the path (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The only part of the path that shards is the T-frame ViT encode: frames are independent through the tower
(T is the batch dim, modeling_image.py:641-643; add_time_attn is False for the image tower).  Rank r encodes a
contiguous block of frames and ONE all-gather of the 1024-wide visual tokens (before the 4096-wide projector: 4x
fewer bytes) rebuilds the full [T, 256, Dv] tensor in chronological order on every rank.  The LLM of one
conversation does not shard (no TP/SP in the reference); conversations are data-parallel replicas with no collective.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L


class TeoComm:
    """The RCCL communicator of this rank behind the C ABI (teo_ctx_create / teo_allgather_visual, include/teo_hip.h).

    The 128-byte unique id is generated on rank 0 by the library and handed to the other ranks by an object broadcast of
    the process group, or through a caller-owned key-value store (control plane, host only); the data path is one ncclAllGather enqueued by the library on the
    caller's HIP stream -- no torch collective."""

    _serial = 0          # communicators created so far by this process (every rank creates them in the same order)

    def __init__(self, rank, world_size, device_index, store=None, group=None, tag=None):
        self.lib = L.load()
        self.rank, self.world = int(rank), int(world_size)
        TeoComm._serial += 1
        uid = (C.c_char * L.COMM_ID_BYTES)()
        if self.world > 1:
            if self.rank == 0:
                L.check(self.lib.teo_comm_unique_id(uid), "teo_comm_unique_id")
            if store is not None:
                # an explicit key-value store (e.g. a TCPStore the caller owns): one key per communicator, so a second
                # communicator never reads the first one's id
                key = f"teo_comm_id/{tag if tag is not None else TeoComm._serial}"
                if self.rank == 0:
                    store.set(key, bytes(uid.raw))
                blob = bytes(store.get(key))
            else:
                # public API only: the 128 bytes travel as an object broadcast of the process group (control plane, host side)
                box = [bytes(uid.raw) if self.rank == 0 else None]
                dist.broadcast_object_list(box, src=0, group=group)
                blob = box[0]
            uid = (C.c_char * L.COMM_ID_BYTES).from_buffer_copy(bytes(blob)[:L.COMM_ID_BYTES])
            id_ptr = C.cast(uid, C.c_void_p)
        else:
            id_ptr = C.c_void_p(0)
        h = C.c_void_p()
        L.check(self.lib.teo_ctx_create(self.rank, self.world, id_ptr, int(device_index), C.byref(h)), "teo_ctx_create")
        self.handle = h

    def info(self):
        """(rank, world_size, cu_count, hbm_bytes) as the library's context holds them (teo_ctx_info)."""
        r, w, cu, hbm = C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
        L.check(self.lib.teo_ctx_info(self.handle, C.byref(r), C.byref(w), C.byref(cu), C.byref(hbm)), "teo_ctx_info")
        return r.value, w.value, cu.value, hbm.value

    def all_gather_rows(self, send, recv):
        """send [rows, dim], recv [world * rows, dim], same dtype, contiguous, on this rank's GPU; runs on the current stream."""
        if not (send.is_contiguous() and recv.is_contiguous() and send.dtype == recv.dtype):
            raise ValueError("all_gather_rows: contiguous tensors of one dtype are required")
        rows, dim = send.shape[0], send[0].numel() if send.shape[0] else recv[0].numel()
        dt = {torch.float32: L.TEO_F32, torch.bfloat16: L.TEO_BF16, torch.float16: L.TEO_F16}[send.dtype]
        st = C.c_void_p(torch.cuda.current_stream(send.device).cuda_stream)
        L.check(self.lib.teo_allgather_visual(self.handle, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), rows, dim,
                                              dt, st), "teo_allgather_visual")
        return recv

    def close(self):
        if getattr(self, "handle", None):
            self.lib.teo_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def frame_partition(T, world_size):
    """Contiguous, order-preserving split of T frames: returns [(start, count)] per rank (counts differ by <= 1)."""
    base, extra = divmod(T, world_size)
    out, s = [], 0
    for r in range(world_size):
        c = base + (1 if r < extra else 0)
        out.append((s, c))
        s += c
    return out


def sharded_frame_features(encode_fn, pixels, group=None, comm=None):
    """pixels [T,3,H,W] (same on every rank) -> features [T, NV, Dv] on every rank.

    encode_fn(frames[c,3,H,W]) -> [c, NV, Dv] is the local ViT encode (TeoEngine.vit_features on the GPU path).
    Ragged T is handled by padding every rank's block to the largest block (the pad rows are dropped after the gather).
    comm: a TeoComm -> the gather is the library's RCCL all-gather (the multi-GPU product path).  Without it the gather goes
    through torch.distributed (gloo: the CPU tests and the two-ranks-on-one-GPU plumbing test).
    """
    if comm is not None:
        ws, rank = comm.world, comm.rank
    elif group is None and not (dist.is_available() and dist.is_initialized()):
        return encode_fn(pixels)
    else:
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
    if comm is not None:
        comm.all_gather_rows(send.view(cmax * NV, Dv), recv.view(ws * cmax * NV, Dv))
    else:
        dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(ws, cmax, NV, Dv)
    return torch.cat([recv[r, :parts[r][1]] for r in range(ws)], dim=0)


def shard_conversations(n_items, rank, world_size):
    """Conversation-level data parallelism: item indices owned by `rank` (round-robin, no collective on the path)."""
    return list(range(rank, n_items, world_size))
