#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json metric).

One "step" = one full frame of the Sponza stand-in (256,152 triangles) at
1920x1080, 256 spp, rendered by the HIP path tracer (PathTracer::Render
equivalent), inputs resident in HBM before the timed region.  The timed frame
is the one in which EVERY counted ray is a full MeshEngine::RayCast — BVH query,
the whole sphere table, normal, uv ("a ray = one nearest-hit
query (BVH + sphere table)"; vmx_opts.reserved[0] bit 8).  The library's
default form of the same frame (the traversal kernels settle the rays whose
Radiance step ends by the path's own draws) and the VMX_SAMPLING_ELIDE_DEAD form
are bit-identical frames that take less time: they are reported beside it
(`frame_ms`), never as `value`.  With N > 1 the
frame is sharded into interleaved 16-row (4-row beyond 4 ranks) stripes (one process per GPU), the
packed stripes are gathered to rank 0 over RCCL and de-interleaved there; the
gather and assembly are inside the timed step.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (first: one HIP runtime per process)
import torch.distributed as dist  # noqa: E402

import vermilion_amd as va  # noqa: E402
from vermilion_amd import dist as vdist  # noqa: E402
from vermilion_amd import scenes  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# vmx_opts.reserved[0] bit 8: every Radiance step shaded in full by k_shade (one phase over all paths, a hit record per
# ray) — every counted ray gets the whole of MeshEngine::RayCast (meshEngine.cpp:239-509), which is SURVEY 8(d)'s ray
HEADLINE_FORM = 0x100
# VALU issue roof: 256 CUs x 4 SIMD-32, a wave64 VALU instruction occupies its SIMD for 2 cycles, 2.4 GHz max clock
# (MI355X_MICROARCH.md "Wave scheduling", "Per-instruction cycle constants") -> wave-level instructions per second
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 2
# Work-based roofline: arithmetic + comparison instructions ONE lane needs for one
# inner-node visit (both child boxes, bbox.cpp:70-83 + the decisions of bvh.cpp:103-114) and for one triangle test
# (triangle.cpp:4-54) in each traversal kernel's data layout; min3/max3 and the division count as one each.
#   camera rays (origin folded into the per-frame tables): 12 mul + 4 min3/max3 + 4 cmp;  cross 9 + det 5 + div 1 +
#     u 6 + v 6 + t 1 + 8 cmp
#   bounce rays: 12 sub + 12 mul + 12 min/max + 4 min3/max3 + 4 cmp;  cross 9 + det 5 + div 1 + tvec 3 + u 6 +
#     cross 9 + v 6 + t 6 + 8 cmp
A_INNER = {"trace_camera": 20, "trace_bounce": 44}
A_TRI = {"trace_camera": 36, "trace_bounce": 53}
KERNEL_TEXT = {
    "trace_camera": "k_trace_w<0> (persistent BVH traversal of the camera rays, one hit record per ray)",
    "trace_bounce": "k_trace_w<1> (persistent BVH traversal of a bounce generation, quad-cooperative record fetch)",
    "shade_camera": "k_shade<0> over every camera path (rest of RayCast: sphere table, normal, uv + one Radiance step)",
    "shade_bounce": "k_shade<1> over every path of a bounce generation", "tail": "k_paths<2> (fused tail of the last bounce generations)",
    "raygen": "k_raygen", "resolve": "k_resolve", "fused": "k_paths<0>",
}


def alg_bytes(stage):
    """SURVEY.md §8(d), whole path: B_ray = 64*N_inner + 48*N_tri + 48 + 64*[tri hit] + 96*[path continues]"""
    return (64 * stage["inner_visits"] + 48 * stage["tri_tests"] + 48 * stage["rays"] + 64 * stage["tri_hits"]
            + 96 * stage["continued"])


def trace_alg_bytes(stage):
    """the traversal kernel's share of it: node records + triangle records + ray read (32 B) + hit write (8 B)"""
    return 64 * stage["inner_visits"] + 48 * stage["tri_tests"] + 40 * stage["rays"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--scene", default="sponza260k")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-spp", type=int, default=32, help="spp of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline frame (no early-stop frame, no SAH-tree frame): the profiling runs of tools/pmc.sh")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N>1 path on a box with fewer GPUs than ranks (frames travel through host memory)")
    ap.add_argument("--device", type=int, default=-1, help="force this HIP device for every rank (rehearsal)")
    ap.add_argument("--multi", default="", help="one process, several devices behind the C ABI (vmx_multi_*): comma-separated "
                    "device list, e.g. 0,1,2,3 (a device may repeat: rehearsal on one GPU); the gather and the assembly on the "
                    "first device are inside the timed step.  Not combined with torch.distributed.run")
    ap.add_argument("--corrected-spp", type=int, default=64, help="spp of the corrected-sampling (r2 = U) frame")
    ap.add_argument("--n1-ms", type=float, default=0.0,
                    help="ms_per_step of the N=1 line of the same command: with it an N>1 line also carries parallel_efficiency "
                         "= n1_ms / (N x ms_per_step)")
    ap.add_argument("--reorder", type=lambda v: int(v, 0), default=0,
                    help="experiment (A/B library only, VMX_LIB=build/libvermilion_hip_ab.so): bounce reordering key, tools/sort_probe.py")
    args = ap.parse_args()

    if args.multi:
        return main_multi(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (there is no CPU path)")
    dev_index = args.device if args.device >= 0 else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    gen, camf = scenes.SCENES[args.scene]
    pos, nrm, uv = gen()
    sc = va.Scene(pos, nrm, uv, device=dev_index)
    desc = sc.describe()
    c = camf()
    W, H, spp = args.width, args.height, args.spp
    cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
    # interleaved stripes: 16 rows up to 4 ranks, 4 rows beyond (measured per rank on one GPU, tools/shard_probe.py:
    # the slowest of 8 ranks takes 17.5 ms with 16-row stripes, 16.7 ms with 4-row stripes; at 4 ranks 29.7 / 30.0 ms)
    stripe = 16 if world <= 4 else 4
    rows = va.local_rows(H, stripe, rank, world)
    local = torch.empty((rows, W, 5), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step(early_stop=False, counters=False, sampling=va.VMX_SAMPLING_PARITY, pipeline=HEADLINE_FORM, times=None):
        opts = va.make_opts(seed=args.seed, early_stop=early_stop, sampling=sampling, rank=rank,
                            world=world, stripe_rows=stripe, collect_counters=counters, reorder=args.reorder, pipeline=pipeline)
        st = sc.render_device(cam, opts, local.data_ptr(), stream)
        st["kernels"] = sc.timings()  # per-kernel hipEvent durations of this frame (on the render stream)
        src = local if args.backend == "nccl" or world == 1 else local.cpu()
        frame = vdist.gather_frame(src, W, H, stripe, rank, world, dst=0, times=times)
        return st, frame

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    exchange = {}  # N > 1: the exchange step of the timed headline steps, apart from the rendering

    def timed(k, exchange_out=None, **kw):
        # (a generation-2 Python garbage collection inside a timed region showed up as +36 ms on the three
        # early-stop frames whenever --steps happened to place it there: collect first, keep it out)
        gc.collect()
        gc.disable()
        times = vdist.ExchangeTimes() if (world > 1 and exchange_out is not None) else None
        sync()
        t0 = time.perf_counter()
        stats = [step(times=times, **kw)[0] for _ in range(k)]
        sync()
        dt = time.perf_counter() - t0
        gc.enable()
        rays = float(sum(s["rays_primary"] + s["rays_secondary"] for s in stats))
        if world > 1:
            rdev = dev if args.backend == "nccl" else torch.device("cpu")
            t = torch.tensor([dt], dtype=torch.float64, device=rdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            r = torch.tensor([rays], dtype=torch.float64, device=rdev)
            dist.all_reduce(r, op=dist.ReduceOp.SUM)
            dt, rays = float(t.item()), float(r.item())
            if times is not None:
                # per rank: device time of its stripes' render, of its part in the gather (hipEvents on the stream the
                # collective is ordered on; wall clock for the gloo rehearsal) and, on the root, of k_assemble
                mine = torch.tensor([sum(s["ms_device"] for s in stats) / k, times.ms("gather") / k, times.ms("assemble") / k],
                                    dtype=torch.float64, device=rdev)
                allr = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(allr, mine)
                allr = torch.stack(allr).cpu().numpy()
                exchange_out.update({
                    "slowest_rank_render_ms": round(float(allr[:, 0].max()), 3),
                    "fastest_rank_render_ms": round(float(allr[:, 0].min()), 3),
                    "rank_render_ms": [round(float(x), 3) for x in allr[:, 0]],
                    # a rank that finishes early waits inside the collective for the slowest one: the root's figure is
                    # the one that lies on the frame's critical path
                    "gather_ms": round(float(allr[0, 1]), 3),
                    "gather_ms_by_rank": [round(float(x), 3) for x in allr[:, 1]],
                    "assemble_ms": round(float(allr[0, 2]), 3),
                    "what": "per timed step: hipEvent pairs around each rank's render, around dist.gather (RCCL over xGMI) "
                            "and around k_assemble on the root; a rank's gather time includes its wait for the slowest rank",
                })
        return dt, rays, stats

    # counters pass (instrumented kernels, untimed): algorithmic bytes of this exact frame
    cst, _ = step(counters=True)
    for _ in range(args.warmup):
        step()
    dt, rays, stats = timed(args.steps, exchange_out=exchange)
    ms_per_step = dt / args.steps * 1e3
    value = rays / dt / 1e6

    # reference-faithful frame (early stop on, pathtracer.cpp:290-311), same frame otherwise
    es_k = 0
    if not args.no_extras:
        step(early_stop=True)
        es_k = max(1, min(args.steps, 3))
        es_dt, es_rays, es_stats = timed(es_k, early_stop=True)
        step(early_stop=True, pipeline=0)
        esd_dt, esd_rays, esd_stats = timed(es_k, early_stop=True, pipeline=0)

    # The library's DEFAULT form of the same frame: whoever creates a ray also notes whether the Radiance step that traces
    # it is the path's last one by the path's own draws (DESIGN_HISTORY.md 5.1); the traversal kernels settle such rays where they
    # finish — BVH query, the light spheres' reach test, counted — and hand only the others to k_shade.  Bit-identical
    # frame in less time, but its settled rays are not full RayCasts, so it is a frame time, not the headline's ray rate.
    sorted_info = None
    if not args.no_extras:
        _, g0 = step()
        g0 = g0.clone() if g0 is not None else None
        _, g1 = step(pipeline=0)
        fs_k = max(1, min(args.steps, 3))
        fs_dt, fs_rays, fs_stats = timed(fs_k, pipeline=0)
        sorted_info = {
            "what": "the library's default form of the headline frame (vmx_opts.reserved[0] = 0): the traversal kernels settle "
                    "the rays whose Radiance step ends by the path's own draws (BVH query + light-sphere reach test, counted) "
                    "and hand the others to k_shade as records; for the settled rays the wall spheres, normal and uv are "
                    "never evaluated, so its ray rate is a BVH-query rate, not SURVEY 8(d)'s",
            "frame_bit_identical_to_headline": bool(torch.equal(g0.view(torch.int32), g1.view(torch.int32))) if rank == 0 else None,
            "ms_per_frame": round(fs_dt / fs_k * 1e3, 3), "bvh_queries_Mrays_per_s": round(fs_rays / fs_dt / 1e6, 2),
            "rays_per_frame": int(fs_rays / fs_k),
            "kernel_ms": {k: round(sum(x["kernels"][k]["ms"] for x in fs_stats) / fs_k, 3)
                          for k in fs_stats[0]["kernels"] if fs_stats[0]["kernels"][k]["launches"]},
        }

    # VMX_SAMPLING_ELIDE_DEAD: the same frames, bit for bit, without the rays whose step cannot change the path's colour
    # (78 % of them under the reference's r2 = 10 U).  Wall-clock per frame is the second half of BASELINE.json's metric;
    # the headline value above stays the default build's, which traces every ray the reference traces.
    el_info = None
    if not args.no_extras:
        ELIDE = va.VMX_SAMPLING_PARITY | va.VMX_SAMPLING_ELIDE_DEAD
        same = []
        for es in (False, True):
            _, f0 = step(early_stop=es)
            f0 = f0.clone() if f0 is not None else None  # (one GPU: the frame is the render target itself)
            _, f1 = step(early_stop=es, sampling=ELIDE, pipeline=0)
            if rank == 0:
                same.append(bool(torch.equal(f0.view(torch.int32), f1.view(torch.int32))))
        el_k = max(1, min(args.steps, 3))
        el_dt, el_rays, el_stats = timed(el_k, sampling=ELIDE, pipeline=0)
        ele_dt, ele_rays, ele_stats = timed(el_k, early_stop=True, sampling=ELIDE, pipeline=0)

        def kms(sts):
            return {k: round(sum(x["kernels"][k]["ms"] for x in sts) / len(sts), 3)
                    for k in sts[0]["kernels"] if sts[0]["kernels"][k]["launches"]}
        el_info = {
            "what": "the same two frames with VMX_SAMPLING_ELIDE_DEAD (opt-in): rays whose Radiance step provably cannot change "
                    "the path's colour are not traced (vmx_kernels.hip: step_is_dead); rays_per_frame counts traced rays only",
            "frames_bit_identical_to_headline": all(same) if rank == 0 else None,
            "fixed_count": {"ms_per_frame": round(el_dt / el_k * 1e3, 3), "rays_per_frame": int(el_rays / el_k),
                            "speedup_vs_headline": round(ms_per_step / (el_dt / el_k * 1e3), 2), "kernel_ms": kms(el_stats)},
            "early_stop": {"ms_per_frame": round(ele_dt / el_k * 1e3, 3), "rays_per_frame": int(ele_rays / el_k),
                           "speedup_vs_headline_form": round((es_dt / es_k) / (ele_dt / el_k), 2) if es_k else None,
                           "kernel_ms": kms(ele_stats)},
        }

    # `corrected` sampling (r2 = U: a real cosine-weighted lobe, pathtracer.cpp:156,170 with the factor 10 removed):
    # SURVEY 8(d) / BASELINE.md §3 ask for it beside `parity`.  Bounces dominate here (~25 rays per sample).
    corr_info = None
    if world == 1 and not args.no_extras and args.corrected_spp >= 4:
        ccam = va.make_camera(c["position"], c["rotation_deg"], W, H, args.corrected_spp, back_size=(3.6, 3.6 * H / W))
        copts = va.make_opts(seed=args.seed, early_stop=False, sampling=va.VMX_SAMPLING_CORRECTED)
        sc.render_device(ccam, copts, local.data_ptr(), stream)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        cs_ = sc.render_device(ccam, copts, local.data_ptr(), stream)
        torch.cuda.synchronize(dev)
        cdt = time.perf_counter() - t0
        ck = sc.timings()
        crays = cs_["rays_primary"] + cs_["rays_secondary"]
        corr_info = {
            "what": "same scene and camera with r2 = U instead of the reference's r2 = 10 U (VMX_SAMPLING_CORRECTED), fixed "
                    f"{args.corrected_spp} spp: no image-parity claim against the reference, oracle parity only "
                    "(tests/test_gpu_parity.py)",
            "spp": args.corrected_spp, "ms_per_frame": round(cdt * 1e3, 3), "Mrays_per_s": round(crays / cdt / 1e6, 2),
            "rays_per_frame": int(crays), "rays_per_sample": round(crays / max(cs_["samples"], 1), 2),
            "kernel_ms": {k: round(v["ms"], 3) for k, v in ck.items() if v["launches"]},
            "kernel_launches": {k: int(v["launches"]) for k, v in ck.items() if v["launches"]},
        }

    # §8 f-4: the engine's default integrator on the same scene and camera (BruteForceTracer, most pixels stop after
    # 3 samples): device time of one frame
    bf_info = None
    if world == 1 and not args.no_extras:
        bopts = va.make_opts(seed=args.seed)
        sc.render_bruteforce(cam, bopts)
        _, bst = sc.render_bruteforce(cam, bopts)
        bf_info = {"what": "BruteForceTracer::Render (integrators.cpp:9-186) of the same scene and camera, device time",
                   "ms_per_frame": round(bst["ms_device"], 3), "samples": int(bst["samples"]),
                   "Mrays_per_s": round((bst["rays_primary"] + bst["rays_secondary"]) / bst["ms_device"] / 1e3, 2)}

    # §8 f-1 quality builder (binned SAH, not the reference's topology): same frame, extra figure only
    q_info = None
    cq_info = None
    if world == 1 and not args.no_extras:
        # ... and the same quality from the GPU builder (PLOC): scene creation is then a per-frame operation
        va.Scene(pos, nrm, uv, device=dev_index, builder=va._lib.VMX_BVH_PLOC).close()  # first hipcub launches
        t0 = time.perf_counter()
        psc = va.Scene(pos, nrm, uv, device=dev_index, builder=va._lib.VMX_BVH_PLOC)
        p_build = time.perf_counter() - t0
        popts = va.make_opts(seed=args.seed, early_stop=False, collect_counters=True)
        pc = psc.render_device(cam, popts, local.data_ptr(), stream)
        popts = va.make_opts(seed=args.seed, early_stop=False)
        psc.render_device(cam, popts, local.data_ptr(), stream)  # (allocates what the uncounted form needs: not in the timed frames)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        pk = max(1, min(args.steps, 3))
        pst = [psc.render_device(cam, popts, local.data_ptr(), stream) for _ in range(pk)]
        torch.cuda.synchronize(dev)
        pdt = time.perf_counter() - t0
        p_info = {
            "what": "same frame over the tree the GPU builds by parallel locally-ordered clustering (VMX_BVH_PLOC); "
                    "scene_create_ms includes the host-to-device copy of the triangles",
            "scene_create_ms": round(p_build * 1e3, 2),
            "Mrays_per_s": round(sum(s["rays_primary"] + s["rays_secondary"] for s in pst) / pdt / 1e6, 2),
            "ms_per_frame": round(pdt / pk * 1e3, 3),
            "inner_visits_per_ray": round(pc["primary"]["inner_visits"] / max(pc["primary"]["rays"], 1), 2),
            "bvh": {k: psc.describe()[k] for k in ("n_nodes", "max_depth")},
        }
        # `corrected` sampling over the same GPU-built quality tree: the configuration a user of a real cosine-lobe tracer
        # would pick (bounce rays go from ~95 to ~40 node visits)
        cq_info = None
        if args.corrected_spp >= 4:
            ccam = va.make_camera(c["position"], c["rotation_deg"], W, H, args.corrected_spp, back_size=(3.6, 3.6 * H / W))
            copts = va.make_opts(seed=args.seed, early_stop=False, sampling=va.VMX_SAMPLING_CORRECTED)
            psc.render_device(ccam, copts, local.data_ptr(), stream)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            cq = psc.render_device(ccam, copts, local.data_ptr(), stream)
            torch.cuda.synchronize(dev)
            cqdt = time.perf_counter() - t0
            cqk = psc.timings()
            cqrays = cq["rays_primary"] + cq["rays_secondary"]
            cq_info = {
                "what": "corrected sampling (r2 = U) over the tree the GPU builds by parallel locally-ordered clustering "
                        f"(VMX_BVH_PLOC), fixed {args.corrected_spp} spp: oracle parity over the exported tree "
                        "(tests/test_gpu_parity.py::test_quality_bvh_builder_f1), no image-parity claim against the reference",
                "spp": args.corrected_spp, "ms_per_frame": round(cqdt * 1e3, 3), "Mrays_per_s": round(cqrays / cqdt / 1e6, 2),
                "rays_per_frame": int(cqrays),
                "kernel_ms": {k: round(v["ms"], 3) for k, v in cqk.items() if v["launches"]},
                "kernel_launches": {k: int(v["launches"]) for k, v in cqk.items() if v["launches"]},
            }
        psc.close()
        qsc = va.Scene(pos, nrm, uv, device=dev_index, builder=va._lib.VMX_BVH_SAH)
        qopts = va.make_opts(seed=args.seed, early_stop=False, collect_counters=True)
        qc = qsc.render_device(cam, qopts, local.data_ptr(), stream)
        qopts = va.make_opts(seed=args.seed, early_stop=False)
        qsc.render_device(cam, qopts, local.data_ptr(), stream)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        qk = max(1, min(args.steps, 3))
        qst = [qsc.render_device(cam, qopts, local.data_ptr(), stream) for _ in range(qk)]
        torch.cuda.synchronize(dev)
        qdt = time.perf_counter() - t0
        qrays = sum(s["rays_primary"] + s["rays_secondary"] for s in qst)
        q_info = {
            "what": "same frame over the binned-SAH tree (vmx_scene_create_ex, VMX_BVH_SAH): ties / pruning order "
                    "follow that tree, so triangle IDs can differ from the reference on exact-distance ties",
            "Mrays_per_s": round(qrays / qdt / 1e6, 2),
            "ms_per_frame": round(qdt / qk * 1e3, 3),
            "inner_visits_per_ray": round(qc["primary"]["inner_visits"] / max(qc["primary"]["rays"], 1), 2),
            "bvh": {k: qsc.describe()[k] for k in ("n_nodes", "max_depth")},
        }
        qsc.close()

    if rank == 0:
        prim_ms = sum(s["primary"]["ms"] for s in stats)
        # per-kernel device time of the timed steps, measured live (hipEvent pairs on the render stream)
        kms = {k: sum(s["kernels"][k]["ms"] for s in stats) / args.steps for k in stats[0]["kernels"]}
        klaunch = {k: sum(s["kernels"][k]["launches"] for s in stats) for k in stats[0]["kernels"]}
        dominant = max(kms, key=kms.get)
        prof = None
        ppath = os.path.join(ROOT, "profiles", "counters.json")
        if os.path.exists(ppath) and world == 1 and (W, H, spp, args.scene) == (1920, 1080, 256, "sponza260k"):
            try:
                pj = json.load(open(ppath))
                # (counters are a property of binary + workload + pipeline form: only those collected on this headline form)
                prof = pj["kernels"] if pj.get("form") == HEADLINE_FORM else None
                # one timed interval of the library (vmx_timings) covers both shading phases: add their counters up
                # (instructions, cache accesses and HBM bytes are additive; lane utilisation weighted by instructions)
                for stage in ("camera", "bounce") if prof else ():
                    a, b = prof.get("shade_ends_" + stage, {}).get("derived"), prof.get("shade_" + stage, {}).get("derived")
                    if a and b:
                        va_, vb_ = a.get("valu_wave_insts_per_launch", 0.0), b.get("valu_wave_insts_per_launch", 0.0)
                        if va_ + vb_ > 0:
                            b["valu_lane_utilization"] = (a.get("valu_lane_utilization", 0.0) * va_
                                                          + b.get("valu_lane_utilization", 0.0) * vb_) / (va_ + vb_)
                        for key in ("valu_wave_insts_per_launch", "salu_wave_insts_per_launch", "l1_accesses_per_launch",
                                    "hbm_bytes_per_launch"):
                            if key in a and key in b:
                                b[key] = b[key] + a[key]
            except Exception:
                prof = None

        # what the two traversal kernels had to do in one step (counters pass of this exact frame): the camera-ray
        # kernel traces every depth-0 ray; the first bounce generation (the paths that continue after their first
        # hit) goes through k_trace_w<1>, later generations through the fused tail — the instrumented kernels count
        # per stage, so the bounce kernel's share is the stage's per-ray averages x its rays
        work = {}
        if world == 1:
            pr, bo = cst["primary"], cst["bounce"]
            work["trace_camera"] = {"rays": pr["rays"], "inner_visits": pr["inner_visits"], "tri_tests": pr["tri_tests"],
                                    "alg_bytes": trace_alg_bytes(pr), "from": "counters pass, depth-0 stage"}
            if bo["rays"] and klaunch.get("trace_bounce"):
                g1 = min(pr["continued"], bo["rays"])
                f = g1 / bo["rays"]
                work["trace_bounce"] = {"rays": g1, "inner_visits": bo["inner_visits"] * f, "tri_tests": bo["tri_tests"] * f,
                                        "alg_bytes": trace_alg_bytes(bo) * f,
                                        "from": "counters pass, bounce stage averages x rays of the first generation"}

        def kernel_roof(name):
            """Roofline of one kernel against the VALU issue peak (the roof that binds: DESIGN.md 6).
            frac / achieved = the WORK-based figure: the lane-operations the reference's tests need (counters pass of this
            exact frame x A_INNER / A_TRI), as full 64-lane instructions, over the launch duration measured here — it cannot
            exceed 1 and does not grow with wasted instructions.  issue_frac = wave-level VALU instructions actually issued
            (rocprofv3 PMC of this exact frame, profiles/counters.json) over the same duration: how busy the issue slots are."""
            n = max(klaunch[name], 1)
            avg_ms = kms[name] * args.steps / n
            if n > args.steps:  # several launches per step: the counters are those of the longest one, so is the duration
                avg_ms = sum(s["kernels"][name]["longest_ms"] for s in stats) / args.steps
            r = {"kernel": KERNEL_TEXT.get(name, name), "avg_launch_ms": round(avg_ms, 4),
                 "launches_per_step": n // max(args.steps, 1), "ms_per_step": round(kms[name], 3),
                 "bound": "valu_issue", "achieved": None, "peak": round(VALU_ISSUE_PEAK / 1e9, 1), "unit": "Gwave-inst/s",
                 "frac": None, "traffic": None}
            w = work.get(name)
            if w and avg_ms > 0:
                need = w["inner_visits"] * A_INNER[name] + w["tri_tests"] * A_TRI[name]
                ach = need / 64 / (avg_ms * 1e-3)
                r.update({"achieved": round(ach / 1e9, 2), "frac": round(ach / VALU_ISSUE_PEAK, 4),
                          "useful_valu_frac": round(ach / VALU_ISSUE_PEAK, 4),
                          "useful_lane_ops_per_ray": round(need / max(w["rays"], 1), 1),
                          "work": {"rays": int(w["rays"]), "inner_visits": int(w["inner_visits"]), "tri_tests": int(w["tri_tests"]),
                                   "A_inner": A_INNER[name], "A_tri": A_TRI[name], "from": w["from"]}})
            d = (prof or {}).get(name, {}).get("derived")
            if d and avg_ms > 0:
                # (a kernel launched more than once per step: counters are those of its longest launch; the bench
                # frame launches each traversal kernel once)
                iss = d["valu_wave_insts_per_launch"] / (avg_ms * 1e-3)
                r.update({"issue_achieved": round(iss / 1e9, 2), "issue_frac": round(iss / VALU_ISSUE_PEAK, 4),
                          "traffic": d.get("hbm_bytes_per_launch"),
                          "valu_wave_insts_per_launch": int(d["valu_wave_insts_per_launch"]),
                          # scalar ALU instructions share the issue slots (tools/sload_probe.hip): the same rate with them
                          "salu_wave_insts_per_launch": int(d["salu_wave_insts_per_launch"]) if "salu_wave_insts_per_launch" in d else None,
                          "issue_frac_valu_plus_salu": round((d["valu_wave_insts_per_launch"] + d["salu_wave_insts_per_launch"])
                                                             / (avg_ms * 1e-3) / VALU_ISSUE_PEAK, 4)
                          if "salu_wave_insts_per_launch" in d else None,
                          "valu_lane_utilization": round(d.get("valu_lane_utilization", 0.0), 3),
                          "effective_clock_GHz_under_pmc": round(d.get("effective_clock_GHz", 0.0), 3),
                          "l1_accesses_per_clk_per_cu": round(d["l1_accesses_per_launch"] / (avg_ms * 1e-3) / 2.4e9 / 256, 3)
                          if "l1_accesses_per_launch" in d else None,
                          "hbm_frac_measured": round(d["hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                          if "hbm_bytes_per_launch" in d else None})
                if "frac" in r and r["frac"] is None:  # a kernel without a work model (shading, fused tail): issue occupancy only
                    r["note"] = "no work model for this kernel: issue_frac is the occupancy of the issue slots, not a work-based fraction"
                if w:
                    # lane-operations the VALU executed per ray, and per-launch bytes through the vector L1 (64 B per
                    # tag lookup) and the scalar cache (64 B per s_load_dwordx16) next to the algorithmic bytes
                    r["valu_lane_ops_per_ray"] = round(d["valu_wave_insts_per_launch"] * 64 * d.get("valu_lane_utilization", 1.0)
                                                       / max(w["rays"], 1), 1)
                    c = (prof or {}).get(name, {}).get("counters", {})
                    r["cache_bytes_per_launch"] = {
                        "vector_l1": int(d["l1_accesses_per_launch"] * 64) if "l1_accesses_per_launch" in d else None,
                        "scalar_cache": int(c["SQ_INSTS_SMEM"] * 64) if "SQ_INSTS_SMEM" in c else None,
                        "algorithmic": int(w["alg_bytes"])}
            return r

        roof = kernel_roof(dominant)
        if not prof:  # no PMC profile for this configuration: say so instead of inventing an issue fraction or a traffic figure
            roof["note"] = ("profiles/counters.json holds no counters for this workload / pipeline form: traffic and issue_frac "
                            "are null, frac is the work-based figure from the live counters pass")
        roof["frac_what"] = ("(inner_visits x A_inner + tri_tests x A_tri) / 64 / launch duration / VALU issue peak: the share of the "
                             "issue roof spent on arithmetic the reference's tests need (bbox.cpp:70-83, bvh.cpp:103-114, "
                             "triangle.cpp:4-54); issue_frac counts every VALU instruction issued")
        roof["source"] = ("work: the counters pass of this run (vmx_opts.collect_counters, equal to the oracle's visit counts); "
                          "issued instructions and HBM bytes: profiles/counters.json (rocprofv3 --pmc, tools/pmc.sh + "
                          "tools/make_counters.py); durations: live hipEvent pairs on the render stream")
        # HBM bytes of the whole frame (sum over its kernels of the PMC bytes per launch x launches per step) against the
        # 8 TB/s peak over the step: north_star's "share of the HBM roofline" — not the binding roof of this path
        if prof:
            hb, missing = 0.0, []
            for k in kms:
                if kms[k] <= 0:
                    continue
                d_ = prof.get(k, {}).get("derived", {})
                if "hbm_bytes_per_launch" in d_:
                    hb += d_["hbm_bytes_per_launch"] * max(klaunch[k] // max(args.steps, 1), 1)
                else:
                    missing.append(k)
            roof["hbm_frac_whole_frame"] = round(hb / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            roof["hbm_bytes_whole_frame"] = int(hb)
            if missing:
                roof["hbm_frac_whole_frame_missing_kernels"] = missing
        else:
            roof["hbm_frac_whole_frame"] = None
        roof["other_kernels"] = {k: kernel_roof(k) for k in ("trace_camera", "trace_bounce", "shade_camera", "tail")
                                 if k != dominant and kms.get(k, 0) > 0}
        # SURVEY §8(d)'s algorithmic bytes, kept as the secondary view: the scene (34 MB) is cache-resident, so
        # HBM is not the roof of this path (measured HBM traffic is `hbm_frac_measured` of peak)
        cam_bytes = trace_alg_bytes(cst["primary"]) / max(cst["primary"]["launches"], 1)
        cam_ms = kms["trace_camera"] * args.steps / max(klaunch["trace_camera"], 1) if klaunch.get("trace_camera") else 0
        roof["algorithmic"] = {
            "what": "SURVEY 8(d) bytes per ray x rays of the camera-ray launch / its duration, against the 8 TB/s HBM "
                    "peak: > 1 because the records come from L2 / scalar cache / Infinity Cache, not from HBM",
            "alg_bytes_per_launch": int(cam_bytes),
            "alg_GBs": round(cam_bytes / (cam_ms * 1e-3) / 1e9, 1) if cam_ms else None,
            "alg_frac_of_hbm_peak": round(cam_bytes / (cam_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3) if cam_ms else None,
            "alg_bytes_per_ray": round(trace_alg_bytes(cst["primary"]) / max(cst["primary"]["rays"], 1), 1),
            "alg_bytes_per_ray_whole_path": round(alg_bytes(cst["primary"]) / max(cst["primary"]["rays"], 1), 1),
            "inner_visits_per_ray": round(cst["primary"]["inner_visits"] / max(cst["primary"]["rays"], 1), 2),
            "tri_tests_per_ray": round(cst["primary"]["tri_tests"] / max(cst["primary"]["rays"], 1), 2),
            "bounce_inner_visits_per_ray": round(cst["bounce"]["inner_visits"] / max(cst["bounce"]["rays"], 1), 2),
            "bounce_tri_tests_per_ray": round(cst["bounce"]["tri_tests"] / max(cst["bounce"]["rays"], 1), 2),
        }
        out = {
            "metric": "Mrays/sec (primary+secondary), 1920x1080 Sponza",
            "value": round(value, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.scene} ({desc['ntris']} tris, procedural Sponza stand-in) {W}x{H} {spp}spp, "
                            "reference sampling (r2=10U), fixed spp (early stop off), reference sphere table",
                "rays_per_frame": int(rays / args.steps),
                "ray": "one full MeshEngine::RayCast (meshEngine.cpp:239-509): BVH::getIntersection, the whole sphere table, "
                       "normal and uv — for EVERY counted ray (SURVEY 8(d)); vmx_opts.reserved[0] bit 8.  The library's "
                       "default form of this frame and the VMX_SAMPLING_ELIDE_DEAD form render the same bits in less time: "
                       "frame_ms, sorted_frame, elided_frame",
                "parallelism": f"stripes{stripe}x{world}" if world > 1 else "single",
                "bvh": {"nodes": desc["n_nodes"], "max_depth": desc["max_depth"], "leaf_size": desc["leaf_size"]},
            },
            "roofline": roof,
            "kernel_ms_per_step": {k: round(v, 3) for k, v in kms.items() if v > 0},
            "stage_ms_per_step": {
                "primary_trace": round(prim_ms / args.steps, 3),
                "bounce_trace_and_tail": round(sum(s["bounce"]["ms"] for s in stats) / args.steps, 3),
                "shade": round(sum(s["shade"]["ms"] for s in stats) / args.steps, 3),
                "device_total": round(sum(s["ms_device"] for s in stats) / args.steps, 3),
            },
            "whole_frame_alg_GBs": round((alg_bytes(cst["primary"]) + alg_bytes(cst["bounce"])) / (ms_per_step * 1e-3) / 1e9, 1)
            if world == 1 else None,
        }
        if sorted_info and el_info:
            # wall-clock per frame is the second half of BASELINE.json's metric: three bit-identical frames
            out["frame_ms"] = {
                "every_ray_a_full_RayCast (headline)": round(ms_per_step, 3),
                "rays_settled_where_they_finish (library default)": sorted_info["ms_per_frame"],
                "VMX_SAMPLING_ELIDE_DEAD": el_info["fixed_count"]["ms_per_frame"],
                "bit_identical": bool(sorted_info["frame_bit_identical_to_headline"] and el_info["frames_bit_identical_to_headline"]),
            }
        if es_k:
            out["reference_frame"] = {
                "what": "same frame with the reference's early-stop rule on (pathtracer.cpp:290-311), every ray a full RayCast "
                        "like the headline; library_default_ms: the library's default form of it",
                "library_default_ms": round(esd_dt / es_k * 1e3, 3),
                "ms_per_frame": round(es_dt / es_k * 1e3, 3),
                "Mrays_per_s": round(es_rays / es_dt / 1e6, 2),
                "rays_per_frame": int(es_rays / es_k),
                "passes": es_stats[0]["passes"],
                "kernel_ms": {k: round(sum(s["kernels"][k]["ms"] for s in es_stats) / es_k, 3)
                              for k in es_stats[0]["kernels"] if es_stats[0]["kernels"][k]["launches"]},
                "kernel_launches": {k: int(es_stats[0]["kernels"][k]["launches"])
                                    for k in es_stats[0]["kernels"] if es_stats[0]["kernels"][k]["launches"]},
                "device_ms": round(sum(s["ms_device"] for s in es_stats) / es_k, 3),
            }
        if q_info:
            out["quality_bvh"] = q_info
            out["quality_bvh_gpu_built"] = p_info
        if cq_info:
            out["corrected_frame_quality_bvh"] = cq_info
        if sorted_info:
            out["sorted_frame"] = sorted_info
        if exchange:
            out["exchange"] = exchange
        if world > 1 and args.n1_ms > 0:
            out["parallel_efficiency"] = round(args.n1_ms / (world * ms_per_step), 4)
        if el_info:
            out["elided_frame"] = el_info
        if corr_info:
            out["corrected_frame"] = corr_info
        if bf_info:
            out["bruteforce_frame"] = bf_info
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pos, nrm, uv, c, W, H, args.cpu_spp, args.seed)
    # everything this process holds is released before the line goes out, so that nothing of it outlives the line
    sc.close()
    del local
    gc.collect()
    torch.cuda.empty_cache()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def main_multi(args):
    """One process, several devices behind the C ABI (vmx_multi_*, what a Vermilion main.cpp would call): every device
    renders its interleaved stripes, pushes them device-to-device into the gather buffer on the first device, which
    de-interleaves; all of it inside the timed step.  Same JSON line as the torch.distributed path."""
    if int(os.environ.get("WORLD_SIZE", "1")) != 1:
        sys.exit("bench.py --multi is a single process: do not launch it through torch.distributed.run")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (there is no CPU path)")
    devices = [int(d) for d in args.multi.split(",") if d != ""]
    world = len(devices)
    gen, camf = scenes.SCENES[args.scene]
    pos, nrm, uv = gen()
    c = camf()
    W, H, spp = args.width, args.height, args.spp
    cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
    stripe = 16 if world <= 4 else 4
    root = torch.device("cuda", devices[0])
    frame = torch.empty((H, W, 5), dtype=torch.float32, device=root)
    ms = va.MultiScene(pos, nrm, uv, devices=devices)
    opts = va.make_opts(seed=args.seed, early_stop=False, sampling=va.VMX_SAMPLING_PARITY, stripe_rows=stripe, pipeline=HEADLINE_FORM)

    def sync():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(torch.device("cuda", d))

    for _ in range(args.warmup):
        ms.render_device(cam, opts, frame.data_ptr())
    gc.collect()
    gc.disable()
    sync()
    t0 = time.perf_counter()
    stats, xch = [], []
    for _ in range(args.steps):
        stats.append(ms.render_device(cam, opts, frame.data_ptr()))
        xch.append(ms.timings())  # the exchange step of this frame, timed apart from the rendering (vmx_multi_timings)
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    rays = float(sum(s["rays_primary"] + s["rays_secondary"] for s in stats))
    es_opts = va.make_opts(seed=args.seed, early_stop=True, stripe_rows=stripe, pipeline=HEADLINE_FORM)
    ms.render_device(cam, es_opts, frame.data_ptr())
    sync()
    t0 = time.perf_counter()
    es = ms.render_device(cam, es_opts, frame.data_ptr())
    sync()
    es_dt = time.perf_counter() - t0
    out = {
        "metric": "Mrays/sec (primary+secondary), 1920x1080 Sponza", "value": round(rays / dt / 1e6, 2), "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{args.scene} ({ms.ntris} tris, procedural Sponza stand-in) {W}x{H} {spp}spp, reference sampling "
                        "(r2=10U), fixed spp (early stop off), reference sphere table",
            "rays_per_frame": int(rays / args.steps),
            "ray": "one full MeshEngine::RayCast for every counted ray (vmx_opts.reserved[0] bit 8), as in the N=1 line",
            "parallelism": f"one process, vmx_multi over devices {devices}, stripes{stripe}x{world}, device-to-device gather "
                           "on the first device",
            "distinct_devices": len(set(devices)),
        },
        "roofline": {"bound": "valu_issue", "achieved": None, "peak": round(VALU_ISSUE_PEAK / 1e9, 1), "unit": "Gwave-inst/s",
                     "frac": None, "traffic": None,
                     "note": "per-kernel counters are a single-device figure: see the N=1 line (python bench.py)"},
        **({"parallel_efficiency": round(args.n1_ms / (world * (dt / args.steps * 1e3)), 4)} if args.n1_ms > 0 else {}),
        "slowest_device_ms_per_step": round(sum(s["ms_device"] for s in stats) / args.steps, 3),
        "exchange": {
            "slowest_rank_render_ms": round(sum(x["slowest_render_ms"] for x in xch) / len(xch), 3),
            "rank_render_ms": [round(sum(x["render_ms"][r] for x in xch) / len(xch), 3) for r in range(world)],
            "gather_ms": round(sum(x["gather_ms"] for x in xch) / len(xch), 3),
            "gather_ms_by_rank": [round(sum(x["copy_ms"][r] for x in xch) / len(xch), 3) for r in range(world)],
            "assemble_ms": round(sum(x["assemble_ms"] for x in xch) / len(xch), 3),
            "wall_ms": round(sum(x["wall_ms"] for x in xch) / len(xch), 3),
            "routes": [{"device": d, "route": {2: "root's own device", 1: "direct peer copy (xGMI)", 0: "staged through the host"}[r]}
                       for d, r in ms.routes()],
            "what": "per timed step: hipEvent pairs on each replica's stream around its render and around its stripes' "
                    "device-to-device copy into the root's gather buffer, and on the root's stream around k_assemble; wall_ms is "
                    "the host clock from handing the jobs out to the assembled frame",
        },
        "reference_frame": {"what": "same frame with the reference's early-stop rule on (pathtracer.cpp:290-311)",
                            "ms_per_frame": round(es_dt * 1e3, 3),
                            "Mrays_per_s": round((es["rays_primary"] + es["rays_secondary"]) / es_dt / 1e6, 2)},
    }
    ms.close()
    del frame
    gc.collect()
    torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)


def cpu_baseline(pos, nrm, uv, c, W, H, spp, seed):
    """The CPU oracle (the reference's algorithm restated, built with the reference's
    -Ofast -fopenmp flags) timed on this box's host cores on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    osc = O.OracleScene(pos, nrm, uv, fast=True)
    cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
    opts = va.make_opts(seed=seed, early_stop=False)
    warm = va.make_camera(c["position"], c["rotation_deg"], W // 8, H // 8, 4, back_size=(3.6, 3.6 * H / W))
    osc.render(warm, opts)
    t0 = time.perf_counter()
    _, st = osc.render(cam, opts)
    dt = time.perf_counter() - t0
    rays = st["rays_primary"] + st["rays_secondary"]
    return {
        "value": round(rays / dt / 1e6, 3),
        "unit": "Mrays/s",
        "cores": O.max_threads(),
        "kind": "port",
        "sample": f"same scene/camera/seed at {W}x{H}, {spp} spp fixed ({rays} rays, {dt:.1f} s), "
                  "OpenMP schedule(dynamic,1) over pixels, g++ -Ofast, per-pixel stderr progress suppressed",
    }


if __name__ == "__main__":
    main()
