"""Where a traversal wave's time goes: wave-cycles and wave-steps of k_trace_w by the state in which the wave enters a
step (diagnostic build, -DVMX_STEP_PROFILE).
  make -C vermilion_amd/csrc OUT=../../build/libvmx_prof.so EXTRA=-DVMX_STEP_PROFILE
  VMX_LIB=build/libvmx_prof.so python tools/step_profile.py [spp]
Cycles are s_memtime differences around a step, as one wave sees them (time other waves of the SIMD were issuing included),
so the shares are shares of the kernel's wave-time; the instrumented kernel itself runs ~20 % slower than the product."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vermilion_amd as va
from vermilion_amd import scenes, _lib
lib = _lib.load()
fn = lib.vmx_debug_step_profile
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H = 1920, 1080
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
sc = va.Scene(pos, nrm, uv)
out = torch.empty((H, W, 5), device="cuda")
o = va.make_opts(seed=1, early_stop=False, reorder=int(os.environ.get("REORDER", "0"), 0), lds_entries=int(os.environ.get("LDSE", "0")))
sc.render_device(cam, o, out.data_ptr())
buf = np.zeros((2, 10, 2), dtype=np.uint64)
ws = lib.vmx_debug_wave_span
ws.argtypes = [ctypes.c_void_p, ctypes.c_int]
span = np.zeros((2, 4), dtype=np.uint64)
fn(None, 1); ws(None, 1)
sc.render_device(cam, o, out.data_ptr()); torch.cuda.synchronize()
fn(buf.ctypes.data, 0); ws(span.ctypes.data, 0)
for src, kn in enumerate(("k_trace_w<0>", "k_trace_w<1>")):
    if span[src, 3]:
        first, last, tot, n = (float(x) for x in span[src])
        print(f"{kn}: {int(n)} waves (all launches of the frame), last wave ends {(last - first) / 100:.0f} us after the first one starts, the average wave {(tot / n - first) / 100:.0f} us")
print({k: round(v["ms"], 2) for k, v in sc.timings().items() if v["launches"]})
names = ["same inner node", "inner nodes, not all the same", "leaves only", "inner nodes and leaves", "no traversing lane",
         "refill section", "NaN-exact batch / same node, mixed octants", "uniform_descent (asm loop)"]
for src, kn in enumerate(("k_trace_w<0> camera rays", "k_trace_w<1> bounce rays")):
    cyc, n = buf[src, :, 0].astype(float), buf[src, :, 1].astype(float)
    print(f"---- {kn}: {cyc.sum() / 1e9:.2f} G wave-cycles, {(n[:5].sum() + n[6] + n[7]) / 1e6:.1f} M wave-steps")
    if n[8]:
        print(f"uniform_descent: {n[8] / 1e6:.1f} M entries, {n[7] / n[8]:.2f} nodes per entry, {cyc[8] / n[8]:.1f} traversing lanes at entry")
    cyc[8] = 0
    for i, nm in enumerate(names):
        if n[i] and i < 8:
            print(f"{nm:32s} {n[i] / 1e6:9.1f} M  {100 * n[i] / max(n[:5].sum() + n[6] + n[7], 1):5.1f} % of steps   {cyc[i] / n[i]:8.0f} cycles each   {100 * cyc[i] / cyc.sum():5.1f} % of wave-time")
