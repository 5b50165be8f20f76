"""EXPERIMENT: how much do two concurrent half-frames overlap?  Two scenes (same geometry, own
workspaces and streams), two host threads, each rendering 1920x1080 at spp/2; the second starts
`delay` ms later so that its camera-ray traversal runs beside the first one's bounce stage."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vermilion_amd as va
from vermilion_amd import scenes

def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    delay = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
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
    t0 = time.perf_counter()
    for i in range(2):
        scs[i].render_device(cam, opts, outs[i].data_ptr())
    torch.cuda.synchronize()
    seq = (time.perf_counter() - t0) * 1e3
    res = {}
    def work(i, d):
        time.sleep(d / 1e3)
        st = scs[i].render_device(cam, opts, outs[i].data_ptr())
        res[i] = st
    for rep in range(3):
        th = [threading.Thread(target=work, args=(i, delay * i)) for i in range(2)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        par = (time.perf_counter() - t0) * 1e3
        print(f"spp {spp} x2, delay {delay} ms, VMX_EXP_BLOCKS={os.environ.get('VMX_EXP_BLOCKS')}: sequential {seq:.1f} ms, concurrent {par:.1f} ms | "
              f"A dev {res[0]['ms_device']:.1f} (p {res[0]['primary']['ms']:.1f} b {res[0]['bounce']['ms']:.1f} s {res[0]['shade']['ms']:.1f}) "
              f"B dev {res[1]['ms_device']:.1f} (p {res[1]['primary']['ms']:.1f} b {res[1]['bounce']['ms']:.1f} s {res[1]['shade']['ms']:.1f})")

main()
