#!/usr/bin/env python3
"""Run ON THE GPU BOX: find the pixel-samples of a full-size scene whose radiance differs from the oracle's, and say WHY: the oracle logs
every ray of such a pixel (origin, direction, interval, what it found); the same rays go through the product's traversal
(pt_debug_intersect); the first ray with a different answer is printed -- or, if the traversal agrees on all of them, the bounce at which
the radiance parts (bounce-limited renders), which then is a shading difference.
usage: python tools/diag_flip.py [sponza|helmet|grid] [frames] [first_frame]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer
import oracle.pyoracle as po
from ray_hook import gpu_intersect, RF_CULL_BACK, RF_CULL_FRONT, RF_FORCE_NON_OPAQUE, RF_ACCEPT_FIRST

which = sys.argv[1] if len(sys.argv) > 1 else "sponza"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
s = {"sponza": scenes.sponza_class, "helmet": scenes.helmet_class, "grid": scenes.material_grid}[which]()
r = Renderer(); hg = s.upload(r)
o = po.Oracle(); ho = s.upload(o, env_raw=r.env_read(hg["env"]) if hg["env"] is not None else None)
og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
r.set_samples_per_trace(1)


def settings(max_bounces=None):
    st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.flags &= ~abi.FLAG_ACCUMULATE
    if max_bounces is not None: st.max_bounces = max_bounces; st.min_bounces = min(st.min_bounces, max_bounces)
    return st


def both(st, f, window=None):
    r.trace(st, s.execute_params(f, env_handle=hg["env"]), og)
    if window: o.set_window(*window)
    o.trace(st, s.execute_params(f, env_handle=ho["env"]), b)
    o.set_window()
    return r.readback(og)[..., :3].astype(np.float64), b[..., :3].astype(np.float64)


t0 = time.time(); events = []
for f in range(first, first + frames):
    A, B = both(settings(), f)
    rel = np.abs(A - B).max(axis=2) / np.maximum(np.abs(B).max(axis=2), 1e-4)
    for y, x in np.argwhere(rel > 1e-3): events.append((f, int(x), int(y), A[y, x].copy(), B[y, x].copy()))
print("%s %dx%d: %d pixel-samples of %d differ by more than 1e-3 (frames %d..%d, %.0f s)" % (s.name, s.width, s.height, len(events), frames * s.width * s.height, first, first + frames - 1, time.time() - t0), flush=True)
for f, x, y, ga, ob in events[:12]:
    print("frame %d pixel (%d, %d): gpu %s oracle %s" % (f, x, y, np.array2string(ga, precision=5), np.array2string(ob, precision=5)))
    o.ray_log(x, y); both(settings(), f, window=(x, y, x + 1, y + 1)); log = o.read_ray_log(); o.ray_log(-1, -1)
    found = False
    for k, q in enumerate(log):
        mode = int(q[8])
        dx = int(q[14])                                                              # the oracle's D3D12 RAY_FLAG_* -> the traversal's
        rf = (RF_CULL_BACK if dx & 0x10 else 0) | (RF_CULL_FRONT if dx & 0x20 else 0) | (RF_FORCE_NON_OPAQUE if dx & 0x2 else 0) | (RF_ACCEPT_FIRST if dx & 0x4 else 0)
        g = gpu_intersect(r, q[None, :8], rf, mode)[0]
        same = (g[0] == q[9]) and (mode == 1 or (g[1] == q[10] and g[4] == q[11] and g[5] == q[12]))
        if not same:
            print("   ray %d of %d (%s): origin %s direction %s interval (%g, %g)\n      oracle: hit %d t %.9g instance %d primitive %d   gpu: hit %d t %.9g instance %d primitive %d"
                  % (k, len(log), "closest" if mode == 0 else "shadow", q[0:3], q[4:7], q[3], q[7], q[9], q[10], q[11], q[12], g[0], g[1], g[4], g[5]))
            o.set_brute_force(True)
            bf = o.intersect_many(q[None, :8], dx, mode)[0]
            o.set_brute_force(False)
            print("      the oracle's exhaustive search over all triangles: hit %d t %.9g instance %d primitive %d" % (bf[0], bf[1], bf[4], bf[5]))
            found = True; break
    if not found:
        print("   the traversal agrees on all %d rays of the oracle's path: a shading difference; radiance by bounce limit:" % len(log))
        for mb in range(0, s.settings.max_bounces + 1):
            A, B = both(settings(mb), f, window=(x, y, x + 1, y + 1))
            print("      max_bounces %d: gpu %s oracle %s%s" % (mb, np.array2string(A[y, x], precision=6), np.array2string(B[y, x], precision=6), "" if np.allclose(A[y, x], B[y, x], rtol=1e-4, atol=1e-7) else "   <-- differ"))
r.close(); o.close()
