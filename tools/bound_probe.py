"""How much of a traversal lies beyond the hit?  For sample rays over the bench scene's reference-topology tree, the
share of inner-node visits and triangle tests whose node was entered with near > t_final — what a perfect upper bound
on the hit distance, known before the traversal starts, would prune (bvh.cpp:69 with t preset).  Statistics only:
float64 Python restatement of the traversal order (bvh.cpp:47-145) over the tree exported by vmx_scene_bvh."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, vermilion_amd as va
from vermilion_amd import scenes
pos, nrm, uv = scenes.sponza260k(); c = scenes.sponza_camera()
sc = va.Scene(pos, nrm, uv)
b = sc.bvh()
tri = np.asarray(pos, np.float64).reshape(-1, 3, 3)[b["prim_order"]]
v0, e1, e2 = tri[:, 0], tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
start, nprims, roff, bbox = b["start"], b["nprims"], b["right_offset"], b["bbox"].astype(np.float64)

def box(i, o, inv):
    lo = (bbox[i, :3] - o) * inv; hi = (bbox[i, 3:] - o) * inv
    tn = np.max(np.minimum(lo, hi)); tf = np.min(np.maximum(lo, hi))
    return tn <= tf, tn

def tri_hit(k, o, d):
    pv = np.cross(d, e2[k]); det = e1[k] @ pv
    if abs(det) <= 1e-8: return None
    inv = 1.0 / det; tv = o - v0[k]; u = (tv @ pv) * inv
    if u < 0 or u > 1: return None
    q = np.cross(tv, e1[k]); v = (d @ q) * inv
    if v < 0 or u + v > 1: return None
    t = (e2[k] @ q) * inv
    return t if t > 0 else None

def trace(o, d, best=999999999.0):
    with np.errstate(divide="ignore"):
        inv = 1.0 / d
    todo = [(0, -9999999.0)]; inner, leaf = [], []
    while todo:
        ni, near = todo.pop()
        if near > best: continue
        if roff[ni] == 0:
            for k in range(start[ni], start[ni] + nprims[ni]):
                leaf.append(near)
                t = tri_hit(k, o, d)
                if t is not None and t < best: best = t
        else:
            inner.append(near)
            h0, n0 = box(ni + 1, o, inv); h1, n1 = box(ni + roff[ni], o, inv)
            if h0 and h1:
                cl, ot, a, bb = ni + 1, ni + roff[ni], n0, n1
                if n1 < n0: cl, ot, a, bb = ot, cl, n1, n0
                todo.append((ot, bb)); todo.append((cl, a))
            elif h0: todo.append((ni + 1, n0))
            elif h1: todo.append((ni + roff[ni], n1))
    return best, np.array(inner), np.array(leaf)

rng = np.random.default_rng(3)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
lo, hi = bbox[0, :3], bbox[0, 3:]
cam = np.array(c["position"], np.float64)
def stats(name, rays):
    ti = tb = li = lb = hits = 0
    for o, d in rays:
        t, inner, leaf = trace(o, d)
        if t > 9e8: t = np.inf
        else: hits += 1
        ti += inner.size; tb += int((inner > t).sum()); li += leaf.size; lb += int((leaf > t).sum())
    print(f"{name}: {len(rays)} rays ({hits} hit), {ti / len(rays):.1f} inner visits per ray, {100 * tb / ti:.1f} % entered with near > t_final; "
          f"{li / len(rays):.1f} triangle tests per ray, {100 * lb / max(li, 1):.1f} % in leaves entered with near > t_final")
cam_rays, hitpts = [], []
for _ in range(N):
    p = lo + rng.random(3) * (hi - lo); d = p - cam; d /= np.linalg.norm(d)
    cam_rays.append((cam, d))
t0 = time.time(); stats("rays from the camera position", cam_rays)
for o, d in cam_rays[:N]:
    t, _, _ = trace(o, d)
    if t < 9e8: hitpts.append(o + d * (t - 1e-3))
bounce = []
for p in hitpts:
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    bounce.append((p, d))
stats("random rays from their hit points", bounce)
print(f"({time.time() - t0:.0f} s)")
