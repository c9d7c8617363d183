"""Multi-GPU sharding of one render: one process per GPU, pixels x samples partition trivially.

The image is cut into 8x8 pixel tiles (column-major tile order, like the image); rank r of N
renders the tiles t with ``t % N == r`` (include/rtw_hip.h ``shard_index/shard_count``) into a
zero-initialised full-size framebuffer, and the N framebuffers are summed onto rank 0 with ONE
collective (``torch.distributed.reduce``; backend "nccl" is RCCL over xGMI on ROCm).  Adding
zeros is exact and every (pixel, chunk) RNG stream is keyed by the pixel, so the result is
bit-identical to the single-GPU image for any N.  The reference has no multi-process path
(only ``Threads.@threads`` over rows, /root/reference/src/render.jl:23).
"""
import numpy as np

from .structs import image_height


def owned_pixel_mask(image_width, shard_index, shard_count):
    """Boolean ``[H, W]`` mask of the pixels shard ``shard_index`` of ``shard_count`` renders."""
    W, H = int(image_width), image_height(image_width)
    tiles_i = (H + 7) // 8
    i = np.arange(H)[:, None] // 8
    j = np.arange(W)[None, :] // 8
    t = j * tiles_i + i
    return (t % int(shard_count)) == int(shard_index)


def render_sharded(render_shard, image_width, *, group=None, dst=0):
    """Run ``render_shard(shard_index, shard_count) -> torch.Tensor`` (this rank's zero-padded
    framebuffer, any shape, same on every rank) and reduce the shards onto rank ``dst``.

    Returns the full framebuffer on rank ``dst`` and the local (partial) one elsewhere.
    ``render_shard`` is the HIP path in production (DeviceRenderer.render_into on this rank's
    GPU); the CPU tests pass a stand-in to exercise the partition + collective under gloo.
    """
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return render_shard(0, 1)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    fb = render_shard(rank, world)
    if world > 1:
        dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return fb
