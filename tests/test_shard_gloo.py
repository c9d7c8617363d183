"""The N>1 path on CPU: two processes, gloo backend, shard partition + ONE reduce onto rank 0.
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

        fb = R.render_sharded(render_shard, g["width"])
        if rank == 0:
            q.put(("ok", np.array_equal(fb.numpy(), g["image"])))
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
