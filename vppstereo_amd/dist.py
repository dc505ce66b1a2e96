"""Frame-batch sharding over the GPUs of one node.

Frames are independent (test.py:307-311 loops over them; the only carried state is libc's
rand(), which becomes a per-frame stream seeded with seed + global frame index, so results do
not depend on the sharding).  Ranks own contiguous blocks of frames; there is no data-path
collective.  The only exchange is the final gather of the float32 disparity maps to rank 0
(torch.distributed: RCCL over xGMI on the GPU box, gloo on CPU for the tests).
"""


def shard_range(n_frames, rank, world_size):
    """Contiguous block [lo, hi) of frames owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_frames, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def frame_seed(base_seed, global_frame_index):
    """srand() seed of a frame: independent of how the batch is sharded."""
    return (int(base_seed) + int(global_frame_index)) & 0xFFFFFFFF


def gather_disparities(local_disp, n_frames, dst=0, group=None):
    """Gather per-rank [b_local,H,W] float32 disparity tensors to rank `dst` as [n_frames,H,W].

    Uses all_gather on equal-sized padded shards (one collective; on 8x MI355X xGMI is fully
    connected so every rank's shard travels its own link).  Returns the full tensor on `dst`,
    None elsewhere.  With world_size == 1 it is the identity."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_disp
    ws = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bmax = (n_frames + ws - 1) // ws
    H, W = local_disp.shape[-2:]
    buf = torch.zeros((bmax, H, W), dtype=local_disp.dtype, device=local_disp.device)
    buf[: local_disp.shape[0]] = local_disp
    parts = [torch.empty_like(buf) for _ in range(ws)]
    if dist.get_backend(group) == "gloo" and buf.is_cuda:
        # dry runs without RCCL: stage through the host
        hparts = [p.cpu() for p in parts]
        dist.all_gather(hparts, buf.cpu(), group=group)
        parts = [p.to(buf.device) for p in hparts]
    else:
        dist.all_gather(parts, buf, group=group)
    if rank != dst:
        return None
    out = []
    for r in range(ws):
        lo, hi = shard_range(n_frames, r, ws)
        out.append(parts[r][: hi - lo])
    return torch.cat(out, 0)
