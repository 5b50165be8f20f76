"""Two ranks on the one GPU of the test box: each renders its stripes with the
HIP path, the frames travel over gloo (two ranks cannot share one device under
RCCL), the root assembles and compares with the unsharded frame."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    import vermilion_amd as va
    from vermilion_amd import dist as vdist
    from vermilion_amd import scenes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        W, H, R = 96, 70, 16
        c = scenes.lattice_camera()
        cam = va.make_camera(c["position"], c["rotation_deg"], W, H, 16)
        with va.Scene(*scenes.lattice(), device=0) as sc:
            img, _ = sc.render(cam, va.make_opts(seed=8, rank=rank, world=world, stripe_rows=R))
            out = vdist.gather_frame(torch.from_numpy(img), W, H, R, rank, world, dst=0)
            if rank == 0:
                full, _ = sc.render(cam, va.make_opts(seed=8))
                q.put(bool(np.array_equal(out.numpy().view(np.uint32), full.view(np.uint32))))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_render_and_gather():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
