#!/usr/bin/env python3
"""Run ON THE GPU BOX: what would REORDERING rays buy the traversal?  The same 4 M secondary rays (from points on the bench scene's surfaces,
random directions) through the product's traversal (pt_debug_intersect, one ray per lane, no refill) in four orders: as generated (random),
sorted by direction octant, by origin cell (Morton), by octant then origin cell.  An upper bound for any in-pipeline binning scheme."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
os.environ["MIPT_DEBUG_INTERSECT_TIMING"] = "1"
import numpy as np
from gltf_renderer_amd import scenes
from gltf_renderer_amd.renderer import Renderer
import oracle.pyoracle as po
from ray_hook import gpu_intersect, surface_rays, RF_ACCEPT_FIRST
s = scenes.sponza_class()
r = Renderer(); s.upload(r)
o = po.Oracle(); s.upload(o)
first, rays = surface_rays(o, s, 8_000_000, 11)
rays = rays[(np.abs(rays[:, 4:7]) > 1e-3).all(axis=1)][:4_000_000]          # the random-direction class only
rays[:, 7] = 1000.0
print(len(rays), "rays", flush=True)
octant = (rays[:, 4] < 0).astype(np.uint32) | ((rays[:, 5] < 0).astype(np.uint32) << 1) | ((rays[:, 6] < 0).astype(np.uint32) << 2)
lo, hi = rays[:, 0:3].min(0), rays[:, 0:3].max(0)
q = np.minimum(((rays[:, 0:3] - lo) / (hi - lo) * 1024).astype(np.uint64), 1023)
def spread(v):
    v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
morton = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
rng = np.random.default_rng(1)
orders = {"random": rng.permutation(len(rays)), "screen order (as generated: coherent primary hits)": np.arange(len(rays)), "by octant": np.argsort(octant, kind="stable"),
          "by origin cell": np.argsort(morton, kind="stable"), "by octant, then origin cell": np.lexsort((morton, octant)),
          "by coarse origin cell (32^3), then octant": np.lexsort((octant, morton >> 15))}
for mode, rf in ((0, 0), (1, RF_ACCEPT_FIRST)):
    print("mode", "closest hit" if mode == 0 else "occlusion", flush=True)
    for name, idx in orders.items():
        sys.stderr.write("  %-52s " % name); sys.stderr.flush()
        gpu_intersect(r, rays[idx], rf, mode)
r.close(); o.close()
