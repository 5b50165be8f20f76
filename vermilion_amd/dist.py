"""Multi-GPU sharding of one frame: one process per GPU, interleaved row stripes,
one gather of the packed per-rank framebuffers to the root (RCCL over xGMI when
the tensors live on GPUs; `torch.distributed` backend "nccl" is RCCL on ROCm),
then a de-interleave on the root.

The reference has no analogue (single process, OpenMP over pixels,
core/integrators/pathtracer.cpp:226); the pixel loop has no cross-pixel state,
so stripes are independent and the only exchange is the final gather.  Sample
streams are keyed by the *global* pixel index, so the assembled frame is
bit-identical to the single-GPU frame for every world size.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .scene import local_row_indices


def max_local_rows(height, stripe_rows, world):
    return max(len(local_row_indices(height, stripe_rows, r, world)) for r in range(max(world, 1)))


def assemble_host(parts, width, height, stripe_rows, world):
    """parts[r]: array-like [>=local_rows_r, W, 5] -> full frame [H, W, 5] (numpy or torch, CPU)"""
    first = parts[0]
    if hasattr(first, "new_empty"):  # torch tensor
        frame = first.new_empty((height, width, 5))
        import torch
        for r in range(world):
            rows = torch.as_tensor(local_row_indices(height, stripe_rows, r, world))
            frame[rows] = parts[r][: len(rows)]
        return frame
    frame = np.empty((height, width, 5), np.float32)
    for r in range(world):
        rows = local_row_indices(height, stripe_rows, r, world)
        frame[rows] = np.asarray(parts[r])[: len(rows)]
    return frame


class ExchangeTimes:
    """The exchange step of gather_frame timed apart from the rendering ("gather time separately"): event
    pairs on the current HIP stream around the gather and around the de-interleave kernel (wall-clock pairs for CPU
    tensors).  Read with ms() once the stream has been synchronised."""

    def __init__(self):
        self.pairs = {"gather": [], "assemble": []}

    def mark(self, cuda_device):
        import time
        if cuda_device is None:
            return time.perf_counter()
        import torch
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(cuda_device))
        return e

    def ms(self, key):
        """summed over the calls since construction"""
        tot = 0.0
        for a, b in self.pairs[key]:
            tot += (b - a) * 1e3 if isinstance(a, float) else a.elapsed_time(b)
        return tot


def gather_frame(local, width, height, stripe_rows, rank, world, dst=0, group=None, times=None):
    """Gather the packed per-rank stripes to `dst` and de-interleave them there.

    local: torch tensor [local_rows, W, 5] (cuda -> RCCL gather + HIP assemble
    kernel; cpu -> gloo gather + index assembly).  Returns the [H, W, 5] frame
    on dst, None elsewhere.  world == 1 returns `local` unchanged.
    times: an ExchangeTimes that collects the gather's and the assembly's durations."""
    import torch
    import torch.distributed as dist

    if world <= 1:
        return local
    dev = local.device if local.is_cuda else None
    mrows = max_local_rows(height, stripe_rows, world)
    stride = mrows * width * 5
    t0 = times.mark(dev) if times else None
    padded = local.new_zeros((stride,))
    padded[: local.numel()] = local.reshape(-1)
    if rank == dst:
        big = local.new_empty((world * stride,))
        views = [big[r * stride:(r + 1) * stride] for r in range(world)]
        dist.gather(padded, gather_list=views, dst=dst, group=group)
    else:
        dist.gather(padded, gather_list=None, dst=dst, group=group)
        if times:
            times.pairs["gather"].append((t0, times.mark(dev)))
        return None
    t1 = times.mark(dev) if times else None
    if big.is_cuda:
        frame = torch.empty((height, width, 5), dtype=torch.float32, device=big.device)
        stream = torch.cuda.current_stream(big.device).cuda_stream
        L.check(L.lib().vmx_assemble_device(C.c_void_p(big.data_ptr()), stride, width, height, stripe_rows, world,
                                            C.c_void_p(frame.data_ptr()), big.device.index or 0,
                                            C.c_void_p(stream)))
    else:
        frame = assemble_host([v.view(mrows, width, 5) for v in views], width, height, stripe_rows, world)
    if times:
        times.pairs["gather"].append((t0, t1))
        times.pairs["assemble"].append((t1, times.mark(dev)))
    return frame
