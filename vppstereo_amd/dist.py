"""Frame-batch sharding over the GPUs of one node.

Frames are independent (test.py:307-311 loops over them; the only carried state is libc's
rand(), which becomes a per-frame stream seeded with seed + global frame index, so results do
not depend on the sharding).  Ranks own contiguous blocks of frames; there is no data-path
collective.  The only exchange is the final gather of the float32 disparity maps to rank 0
(torch.distributed: RCCL over xGMI on the GPU box, gloo on CPU for the tests), issued
asynchronously so that it overlaps the next batch.
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


class GatherHandle:
    """A gather of one step's disparity shards in flight.  `wait()` makes the caller's stream (CUDA) or
    thread (CPU) wait for it; `result()` waits and returns [n_frames,H,W] on the root, None elsewhere."""

    def __init__(self, work, parts, n_frames, ws, is_root, staged=None):
        self._work, self._parts, self._n, self._ws, self._root, self._staged = work, parts, n_frames, ws, is_root, staged

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None

    def result(self):
        import torch
        self.wait()
        if not self._root:
            return None
        out = []
        for r in range(self._ws):
            lo, hi = shard_range(self._n, r, self._ws)
            out.append(self._parts[r][: hi - lo])
        full = torch.cat(out, 0)
        return full.to(self._staged) if self._staged is not None else full


def gather_disparities_async(local_disp, n_frames, dst=0, group=None, force_collective=False):
    """Start gathering the per-rank [b_local,H,W] float32 disparity shards to rank `dst`.

    A gather TO THE ROOT (grouped send/recv under RCCL): only `dst` receives, one shard per xGMI link
    in parallel, nothing travels between the other ranks -- unlike a ring all_gather, whose per-link
    traffic grows with the world size.  Asynchronous: the collective runs on the communicator's own
    stream, so the next batch's kernels overlap it; `local_disp` must stay untouched until `wait()`
    (double-buffer the output).  Shards are padded to equal size (frame counts differ by at most one).

    A world of one rank needs no exchange and returns at once -- unless `force_collective` is set, which sends even that
    case through `torch.distributed.gather` (the communicator is created, the collective is queued on its stream, the
    caller waits on the work handle): how a one-GPU box exercises the RCCL branch (tests/test_gpu_nccl.py).

    The shards are whatever `Engine.vpp_rsgm` left in `local_disp`; a fused aggregation launch that lost its lock step
    has turned them into NaN by then (void_if_lost_kernel), so a void shard cannot pass for disparities on the root."""
    import torch
    import torch.distributed as dist
    have_pg = dist.is_available() and dist.is_initialized()
    if not have_pg or (dist.get_world_size(group) == 1 and not force_collective):
        return GatherHandle(None, [local_disp], local_disp.shape[0], 1, True)
    ws = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bmax = (n_frames + ws - 1) // ws
    H, W = local_disp.shape[-2:]
    if local_disp.shape[0] == bmax:
        buf = local_disp
    else:
        buf = torch.zeros((bmax, H, W), dtype=local_disp.dtype, device=local_disp.device)
        buf[: local_disp.shape[0]] = local_disp
    staged = None
    if dist.get_backend(group) == "gloo" and buf.is_cuda:
        staged = buf.device                      # dry runs without RCCL: stage through the host
        buf = buf.cpu()
    parts = [torch.empty_like(buf) for _ in range(ws)] if rank == dst else None
    dst_global = dist.get_global_rank(group, dst) if group is not None else dst
    work = dist.gather(buf, gather_list=parts, dst=dst_global, group=group, async_op=True)
    return GatherHandle(work, parts, n_frames, ws, rank == dst, staged)


def gather_disparities(local_disp, n_frames, dst=0, group=None):
    """Blocking form of `gather_disparities_async`: the full [n_frames,H,W] tensor on `dst`, None
    elsewhere; the identity when there is no process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_disp
    return gather_disparities_async(local_disp, n_frames, dst, group).result()
