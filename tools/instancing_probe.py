#!/usr/bin/env python3
"""Instancing at the reference's limit (Source/Config.h:24-25: 1000 TLAS instances; Pathtracer.cpp:185-257 rebuilds the TLAS every frame,
RayTracingAccelerationStructure.cpp:292-317), measured on the GPU box: N instances of ONE shared mesh (the streams are uploaded once, every
instance row names the same buffers, like upstream's BLAS sharing) flattened into the world-space tree of this library.

Reports: triangles, device bytes of the flattened structure, full build ms (first / warm), refit ms after ONE instance moved (upstream: a
free TLAS rebuild), trace ms and Mrays/s of a 1080p frame.  usage: instancing_probe.py [instances] [triangles per mesh] [width height]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def instanced_scene(n_inst, tris_per_mesh, width, height, seed=7):
    from gltf_renderer_amd import abi, camera, meshgen, scenes
    nu = max(8, int(round((tris_per_mesh / 2.0) ** 0.5)))
    mesh = meshgen.uv_sphere(nu, nu, 0.45)
    s = scenes.SceneData("instanced_%dx%d" % (n_inst, mesh.num_indices // 3))
    rng = np.random.default_rng(seed)
    mats = [s.add_material(scenes.material(base_color_factor=tuple(rng.uniform(0.2, 0.95, 3)) + (1.0,), metalness_factor=float(rng.uniform(0, 1)),
                                           roughness_factor=float(rng.uniform(0.15, 0.9)))) for _ in range(16)]
    side = int(np.ceil(n_inst ** (1.0 / 3.0)))
    first = None
    for k in range(n_inst):
        cx, cy, cz = k % side, (k // side) % side, k // (side * side)
        T = camera.trs((1.2 * cx + 0.1 * rng.normal(), 1.2 * cy + 0.1 * rng.normal(), 1.2 * cz + 0.5), scale=tuple([float(rng.uniform(0.7, 1.1))] * 3))
        if first is None:
            first = s.add_mesh(mesh, T, mats[k % 16])
        else:                                           # the same streams, another row: what BuildTlas does with a shared BLAS
            d = abi.PtInstanceDesc.from_buffer_copy(bytes(s.instances[first]))
            d.gpu.transform[:] = camera.cm(T)
            d.gpu.normal_transform[:] = camera.cm(camera.inverse_transpose(T))
            d.gpu.material_id = mats[k % 16]
            s.instances.append(d)
            s.mesh_records.append((mesh, T, mats[k % 16]))
            s.triangles += mesh.num_indices // 3
    m_floor = s.add_material(scenes.material(base_color_factor=(0.5, 0.5, 0.5, 1), metalness_factor=0.0, roughness_factor=0.8))
    ext = 1.2 * side
    s.add_mesh(meshgen.grid(8, 8, (-2, -2, 0), (ext + 4, 0, 0), (0, ext + 4, 0)), None, m_floor)
    s.add_light(abi.LIGHT_POINT, position=(ext / 2, -2.0, ext + 3.0), intensity=60.0 * side)
    s.env_image = scenes.sky_image(512, 256, 3.0e3)
    c = ext / 2
    s.world_to_view = camera.orbit_world_to_view((c, c, c * 0.8), 1.9 * ext, 0.55, -0.6)
    s.width, s.height = width, height
    st = abi.PtSettings.app_defaults(); st.min_bounces, st.max_bounces = 2, 5
    s.settings = st
    return s


def main():
    import torch
    from gltf_renderer_amd import abi, camera
    from gltf_renderer_amd.renderer import Renderer
    n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 999
    tpm = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
    t0 = time.time(); s = instanced_scene(n_inst, tpm, w, h); t_gen = time.time() - t0
    r = Renderer()
    torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
    hnd = s.upload(r)
    r.build_accel(); torch.cuda.synchronize()
    first_ms = r.stats().accel_ms
    free1 = torch.cuda.mem_get_info()[0]
    warm = []
    for _ in range(3):
        r.request_rebuild(); r.build_accel(); torch.cuda.synchronize(); warm.append(r.stats().accel_ms)
    # move ONE instance (upstream: the per-frame TLAS rebuild makes this free; here: a refit that rewrites that instance's packets)
    inst = [abi.PtInstanceDesc.from_buffer_copy(bytes(d)) for d in hnd["instances"]]
    refit = []
    for k in range(5):
        T = camera.from_cm(inst[3].gpu.transform[:]); T[:3, 3] += (0.05, 0.02, 0.01)
        inst[3].gpu.transform[:] = camera.cm(T); inst[3].gpu.normal_transform[:] = camera.cm(camera.inverse_transpose(T))
        r.set_instances(inst); r.build_accel(); torch.cuda.synchronize(); refit.append(r.stats().accel_ms)
    q = r.stats()
    out = r.create_output(s.width, s.height)
    st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.reset = 1
    r.set_samples_per_trace(1)
    ms = []
    for f in range(6):
        r.reset_stats(); r.trace(st, s.execute_params(f, env_handle=hnd["env"]), out); st.reset = 0; torch.cuda.synchronize()
        q2 = r.stats(); ms.append((q2.trace_ms, q2.rays))
    tm, rays = sorted(ms)[len(ms) // 2]
    print("instancing probe: %d instances x %d triangles = %d triangles (%d instance rows incl. floor); scene generated in %.1f s" % (n_inst, s.mesh_records[0][0].num_indices // 3, q.bvh_triangles, len(s.instances), t_gen))
    print("  device memory taken by upload + build: %.2f GB (%.0f B per flattened triangle); wide nodes %d; stack need %d" % ((free0 - free1) / 1e9, (free0 - free1) / max(q.bvh_triangles, 1), q.bvh_nodes, q.bvh_stack_need))
    print("  full build: first %.1f ms, warm %.1f ms (%.1f M triangles/s); builds %d refits %d fallbacks %d" % (first_ms, sorted(warm)[1], q.bvh_triangles / sorted(warm)[1] / 1e3, q.accel_builds, q.accel_refits, q.accel_builder_fallbacks))
    print("  one instance moved -> refit %.3f ms (upstream: TLAS rebuild of %d instances, every frame anyway)" % (sorted(refit)[2], len(s.instances)))
    print("  1-spp %dx%d frame, 5 bounces: %.2f ms, %.0f Mrays/s" % (s.width, s.height, tm, rays / tm / 1e3))
    r.close()


if __name__ == "__main__":
    main()
