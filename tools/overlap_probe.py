"""Do two frames rendered concurrently (two scene replicas = two workspaces, two streams, two host
threads) overlap on the GPU?  The second render starts `delay` ms after the first, so that its
camera-ray traversal (VALU-issue bound) runs beside the first one's bounce stage (vector-L1 bound).
Prints the back-to-back time of the same two renders for comparison.  """
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vermilion_amd as va
from vermilion_amd import scenes


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    pos, nrm, uv = scenes.sponza260k()
    c = scenes.sponza_camera()
    W, H = 1920, 1080
    cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
    scs = [va.Scene(pos, nrm, uv), va.Scene(pos, nrm, uv)]
    outs = [torch.empty((H, W, 5), device="cuda") for _ in range(2)]
    opts = va.make_opts(seed=1, early_stop=False)
    for i in range(2):
        scs[i].render_device(cam, opts, outs[i].data_ptr())
    torch.cuda.synchronize()
    seq = []
    for i in (0, 1, 0, 1):
        t0 = time.perf_counter()
        st = scs[i].render_device(cam, opts, outs[i].data_ptr())
        torch.cuda.synchronize()
        seq.append(((time.perf_counter() - t0) * 1e3, st["ms_device"]))
    print(f"{spp} spp frames back to back: wall " + ", ".join(f"{a:.1f}" for a, _ in seq) + " ms; device " +
          ", ".join(f"{b:.1f}" for _, b in seq) + f" ms -> two frames {seq[2][0] + seq[3][0]:.1f} ms")
    res = {}

    def work(i, d):
        time.sleep(d / 1e3)
        res[i] = scs[i].render_device(cam, opts, outs[i].data_ptr())

    for delay in (0.0, 15.0, 30.0, 45.0):
        th = [threading.Thread(target=work, args=(i, delay * i)) for i in range(2)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        par = (time.perf_counter() - t0) * 1e3
        a, b = res[0], res[1]
        print(f"  concurrent, second starts {delay:4.0f} ms later: both done after {par:6.1f} ms | "
              f"A device {a['ms_device']:.1f} (camera trace {a['primary']['ms']:.1f}, bounce {a['bounce']['ms']:.1f}, shade {a['shade']['ms']:.1f}) "
              f"B device {b['ms_device']:.1f} (camera trace {b['primary']['ms']:.1f}, bounce {b['bounce']['ms']:.1f}, shade {b['shade']['ms']:.1f})")


main()
