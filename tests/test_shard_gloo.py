"""The N>1 path on CPU: two processes, gloo backend, shard partition + ONE collective onto rank 0
(both forms: reduce of zero-padded frames, gather of compact tile-major shards).
The render backend here is a stand-in built from the oracle (the HIP path needs a GPU); what is
exercised is exactly the host logic bench.py runs over RCCL: owned_pixel_mask / render_sharded."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import rtw_amd as R
    import rtw_oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = load_golden("cfg1_2spheres_96x54_16spp_d4_f32")

        def render_shard(idx, cnt):
            # stand-in for DeviceRenderer.render_into(shard_index=idx, shard_count=cnt): full oracle
            # image with the pixels this shard does not own zeroed
            img, _ = O.render(g["flat"], g["cam"], g["width"], g["height"], g["spp"], T=np.float32,
                              max_depth=g["depth"], seed=g["seed"], n_chunks=g["n_chunks"], omp_threads=1)
            mask = R.owned_pixel_mask(g["width"], idx, cnt)
            return torch.from_numpy(np.where(mask[..., None], img, np.float32(0)).astype(np.float32))

        order = []

        def render_shard_logged(idx, cnt):
            order.append("render")
            return render_shard(idx, cnt)

        # after_render: bench.py records a HIP event there to time the render and the collective separately (collective_ms)
        fb = R.render_sharded(render_shard_logged, g["width"], after_render=lambda: order.append("hook"))
        assert order == ["render", "hook"], order

        def render_compact(idx, cnt):
            # stand-in for render_into(..., compact=True): this shard's tiles only, tile-major
            img, _ = O.render(g["flat"], g["cam"], g["width"], g["height"], g["spp"], T=np.float32,
                              max_depth=g["depth"], seed=g["seed"], n_chunks=g["n_chunks"], omp_threads=1)
            col_major = np.ascontiguousarray(img.transpose(1, 0, 2)).reshape(-1, 3)       # pixel j*H + i
            index = R.compact_to_frame_index(g["width"], idx, cnt)
            comp = np.where(index[:, None] >= 0, col_major[np.maximum(index, 0)], np.float32(-5)).astype(np.float32)
            return torch.from_numpy(comp.reshape(-1))

        fg = R.render_sharded(render_compact, g["width"], mode="gather")
        if rank == 0:
            frame = fg.numpy().reshape(g["width"], g["height"], 3).transpose(1, 0, 2)
            q.put(("ok", bool(np.array_equal(fb.numpy(), g["image"]) and np.array_equal(frame, g["image"]))))
        else:
            q.put(("partial", float(fb.abs().sum())))
    finally:
        dist.destroy_process_group()


def test_two_rank_reduce_reassembles_image():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ("ok", True) in res


def test_render_sharded_single_process(rtw):
    import torch
    fb = rtw.render_sharded(lambda i, n: torch.full((2, 2), float(n)), 96)
    assert fb.tolist() == [[1.0, 1.0], [1.0, 1.0]]
    calls = []
    rtw.render_sharded(lambda i, n: torch.zeros(4), 96, after_render=lambda: calls.append(1))
    assert calls == [1]
    with pytest.raises(ValueError):
        rtw.render_sharded(lambda i, n: torch.zeros(1), 96, mode="scatter")


@pytest.mark.parametrize("width,count", [(96, 1), (100, 3), (33, 8), (320, 5)])
def test_compact_layout_index_is_a_partition(rtw, width, count):
    """every pixel of the frame appears exactly once over the shards' compact buffers"""
    H = rtw.image_height(width)
    seen = np.zeros(width * H, np.int64)
    for r in range(count):
        idx = rtw.compact_to_frame_index(width, r, count)
        assert idx.size == rtw.compact_elems(width, r, count) // 3 == rtw.local_tile_count(width, r, count) * 64
        np.add.at(seen, idx[idx >= 0], 1)
        mask = rtw.owned_pixel_mask(width, r, count)               # [H, W]; frame position of (i, j) is j*H + i
        own = np.zeros(width * H, bool); own[idx[idx >= 0]] = True
        assert np.array_equal(own.reshape(width, H).T, mask)
    assert np.all(seen == 1)
    padded = rtw.compact_to_frame_index(width, count - 1, count, pad_tiles=rtw.local_tile_count(width, 0, count))
    assert padded.size == rtw.local_tile_count(width, 0, count) * 64 and (padded >= 0).sum() == (idx >= 0).sum()
