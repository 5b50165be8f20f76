"""The phase-pure bounce-traversal probe (vmx_trace_pool.inc, A/B library) on the bench frame: kernel time, wave-steps
and lane-steps per phase, lane utilisation, against k_trace_w<1> on the same rays (profiles/r04_state_pool.txt).
  VMX_LIB=build/libvermilion_hip_ab.so [VMX_AB_POOL_SLOTS=512 VMX_AB_POOL_LEVELS=8 VMX_AB_POOL_BLOCKS=n] python tools/pool_probe.py [spp]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vermilion_amd as va
from vermilion_amd import scenes, _lib
lib = _lib.load()
fn = lib.vmx_debug_pool_stats
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
W, H = 1920, 1080
cam = va.make_camera(c["position"], c["rotation_deg"], W, H, spp, back_size=(3.6, 3.6 * H / W))
sc = va.Scene(pos, nrm, uv)
out = torch.empty((H, W, 5), device="cuda")
ref = None
for name, form in (("k_trace_w<1>", 0x100), ("k_trace_pool", 0x500)):
    o = va.make_opts(seed=1, early_stop=False, pipeline=form)
    sc.render_device(cam, o, out.data_ptr())
    fn(None, 1)
    ms, n = 0.0, 3
    for _ in range(n):
        sc.render_device(cam, o, out.data_ptr())
        ms += sc.timings()["trace_bounce"]["ms"] / n
    chk = int(out.view(torch.int32).to(torch.int64).sum().item())
    if ref is None:
        ref = chk
    print(f"{name:14s} trace_bounce {ms:8.3f} ms   frame checksum {'same' if chk == ref else 'DIFFERENT'}")
    if form & 0x400:
        st = np.zeros(16, np.uint64)
        fn(st.ctypes.data, 0)
        st = st.astype(np.float64) / n
        names = ("inner", "leaf", "finished/refill")
        tot_steps = st[0] + st[2] + st[4]
        for p in range(3):
            if st[p * 2]:
                print(f"  {names[p]:16s} {st[p*2]/1e6:9.2f} M wave-steps  {st[p*2+1]/1e6:10.2f} M lane-steps  lane utilisation {st[p*2+1]/st[p*2]/64:.3f}")
        print(f"  all phases       {tot_steps/1e6:9.2f} M wave-steps, traversal lane-steps {(st[1]+st[3])/1e6:.1f} M, "
              f"rounds per block {st[6]/max(st[8],1):.0f}, blocks {st[8]:.0f}, slots carried to the next round {st[7]/1e6:.1f} M")
sc.close()
