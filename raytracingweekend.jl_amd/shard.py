"""Multi-GPU sharding of one render: one process per GPU, pixels x samples partition trivially.

The image is cut into 8x8 pixel tiles (column-major tile order, like the image); rank r of N
renders the tiles t with ``t % N == r`` (include/rtw_hip.h ``shard_index/shard_count``).  Every
(pixel, chunk) RNG stream is keyed by the pixel and the pixel sums are exact, so the result is
bit-identical to the single-GPU image for any N.  Two ways to put the shards together, both ONE
collective (backend "nccl" is RCCL over xGMI on ROCm):

``reduce``  each rank renders into a zero-initialised full-size framebuffer and the N framebuffers
            are summed onto rank 0 (``torch.distributed.reduce``; adding zeros is exact) -- the
            "RCCL reduce of per-tile framebuffers" of BASELINE.json; moves a full frame per rank.
``gather``  each rank renders only its tiles, compact and tile-major (``RTW_FLAG_COMPACT_TILES``),
            rank 0 gathers them (``torch.distributed.gather``, 1/N of a frame per rank) and
            scatters the tiles into the frame with one indexed copy.

The reference has no multi-process path (only ``Threads.@threads`` over rows,
/root/reference/src/render.jl:23).  Inside ONE process the library itself can use several devices
(``render(..., devices=...)``, ``rtw_params.n_devices``).
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


def local_tile_count(image_width, shard_index, shard_count):
    """Number of 8x8 tiles shard ``shard_index`` of ``shard_count`` owns."""
    W, H = int(image_width), image_height(image_width)
    n_tiles = ((H + 7) // 8) * ((W + 7) // 8)
    return (n_tiles - int(shard_index) + int(shard_count) - 1) // int(shard_count) if n_tiles > shard_index else 0


def compact_elems(image_width, shard_index, shard_count):
    """Elements of the compact tile-major shard buffer (``RTW_FLAG_COMPACT_TILES``)."""
    return local_tile_count(image_width, shard_index, shard_count) * 64 * 3


def compact_to_frame_index(image_width, shard_index, shard_count, pad_tiles=None):
    """For every pixel slot of a compact shard buffer (tile k, slot ``(i mod 8) + 8 (j mod 8)``) the
    pixel's position ``j * H + i`` in the column-major frame, or -1 for slots outside the image (edge
    tiles) and for padding tiles.  ``pad_tiles``: length of the buffer in tiles (>= the shard's own)."""
    W, H = int(image_width), image_height(image_width)
    tiles_i = (H + 7) // 8
    n_local = local_tile_count(W, shard_index, shard_count)
    n = n_local if pad_tiles is None else int(pad_tiles)
    k = np.arange(n)[:, None]
    slot = np.arange(64)[None, :]
    t = k * int(shard_count) + int(shard_index)
    tj, ti = t // tiles_i, t % tiles_i
    i0 = ti * 8 + (slot & 7)
    j0 = tj * 8 + (slot >> 3)
    ok = (k < n_local) & (i0 < H) & (j0 < W)
    return np.where(ok, j0 * H + i0, -1).astype(np.int64).reshape(-1)


_index_cache = {}


def render_sharded(render_shard, image_width, *, group=None, dst=0, mode="reduce", after_render=None):
    """Run this rank's shard and put the shards together on rank ``dst``.

    ``mode="reduce"``: ``render_shard(shard_index, shard_count) -> torch.Tensor`` returns this rank's
    zero-padded full framebuffer (any shape, same on every rank); the framebuffers are summed.
    ``mode="gather"``: ``render_shard(shard_index, shard_count) -> torch.Tensor`` returns the compact
    tile-major shard (``compact_elems`` elements, or longer: only that prefix is used); rank ``dst``
    gets the assembled frame as a flat ``H*W*3`` tensor in ``Matrix{RGB{T}}`` layout.

    ``dst`` is a GLOBAL rank (what ``torch.distributed.reduce/gather`` take); with a sub-``group`` it must be a member.

    ``after_render`` (optional callable): called right after this rank's render has been enqueued and before the collective --
    bench.py records a HIP event there to time the render and the collective separately.

    Returns the full framebuffer on rank ``dst`` and the local (partial) one elsewhere.
    ``render_shard`` is the HIP path in production (DeviceRenderer.render_into on this rank's GPU);
    the CPU tests pass a stand-in to exercise the partition + collective under gloo.
    """
    import torch
    import torch.distributed as dist

    if mode not in ("reduce", "gather"):
        raise ValueError("mode must be 'reduce' or 'gather'")
    W, H = int(image_width), image_height(image_width)
    on = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if on else (0, 1)      # group-relative: the shard index
    # the collectives take a GLOBAL destination rank; "am I the destination" is decided on global ranks too
    is_dst = (dist.get_rank() == int(dst)) if on else True
    if on and group is not None and dist.get_group_rank(group, int(dst)) < 0:
        raise ValueError(f"dst={dst} is not a member of the group")
    fb = render_shard(rank, world)
    if after_render is not None:
        after_render()
    if mode == "reduce":
        if world > 1:
            dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM, group=group)
        return fb
    # gather: equal-sized pieces (the shards differ by at most one tile), then one indexed copy on dst
    pad_tiles = local_tile_count(W, 0, world)
    mine = fb.reshape(-1)[:compact_elems(W, rank, world)]
    piece = mine
    if mine.numel() != pad_tiles * 192:
        piece = torch.zeros(pad_tiles * 192, dtype=fb.dtype, device=fb.device)
        piece[:mine.numel()] = mine
    pieces = None
    if world > 1:
        if is_dst:
            pieces = [torch.empty_like(piece) for _ in range(world)]
        dist.gather(piece, pieces, dst=dst, group=group)
    else:
        pieces = [piece]
    if not is_dst:
        return fb
    key = (W, world, str(fb.device))
    if key not in _index_cache:
        idx = np.concatenate([compact_to_frame_index(W, r, world, pad_tiles) for r in range(world)])
        src = np.flatnonzero(idx >= 0)
        _index_cache[key] = (torch.from_numpy(src).to(fb.device), torch.from_numpy(idx[src]).to(fb.device))
    src, dest = _index_cache[key]
    frame = torch.empty(W * H, 3, dtype=fb.dtype, device=fb.device)
    frame.index_copy_(0, dest, torch.cat(pieces).reshape(-1, 3).index_select(0, src))
    return frame.reshape(-1)
