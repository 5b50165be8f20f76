"""world_size-2 (and 3) gloo runs of the multi-GPU exchange step on CPU tensors:
every rank holds its packed stripes of a known frame, the root must receive the
exact frame.  (The stripes come from the oracle's frame here; on the GPU box
tests/test_gpu_dist.py renders them with the HIP path.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import vermilion_amd as va
from vermilion_amd import dist as vdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, R, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(1234)
        frame = torch.rand((H, W, 5), generator=g)  # same on every rank
        rows = torch.as_tensor(va.local_row_indices(H, R, rank, world))
        local = frame[rows].contiguous()
        assert local.shape[0] == va.local_rows(H, R, rank, world)
        times = vdist.ExchangeTimes()
        out = vdist.gather_frame(local, W, H, R, rank, world, dst=0, times=times)
        # the exchange is timed apart from the rendering on every rank; only the root de-interleaves
        assert len(times.pairs["gather"]) == 1 and times.ms("gather") >= 0.0
        assert len(times.pairs["assemble"]) == (1 if rank == 0 else 0)
        if rank == 0:
            q.put(bool(torch.equal(out, frame)))
        else:
            assert out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H,R", [(2, 19, 37, 4), (3, 8, 50, 16), (2, 5, 3, 16)])
def test_gather_frame_gloo(world, W, H, R):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, R, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
