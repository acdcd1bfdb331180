"""Render-level tests of the CPU oracle: config 1 (BASELINE.json configs[0]: single triangle, 256^2,
1 bounce, fixed seed, diffuse-white, CPU trace), furnace identities, PathtraceScene state semantics
(Source/Pathtracer.cpp:259-367), BVH-vs-brute-force agreement, tile sharding, and the committed
golden tile."""
import os

import numpy as np
import pytest

from gltf_renderer_amd import abi, scenes

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def copy_settings(s):
    return abi.PtSettings.from_buffer_copy(bytes(s))


def render(o, s, h, frames=1, settings=None, out=None, **kw):
    st = settings or s.settings
    if out is None:
        out = np.zeros((s.height, s.width, 4), np.float32)
    for f in range(frames):
        o.trace(st, s.execute_params(frame=f, env_handle=h["env"], **kw), out)
    return out


def test_config1_single_triangle_white_furnace(oracle_lib):
    s = scenes.single_triangle(256)
    o = oracle_lib.Oracle()
    h = s.upload(o)
    out = render(o, s, h)
    # diffuse-white in a constant (1,1,1) environment: every path carries weight 1 into the environment
    assert np.allclose(out[..., :3], 1.0, atol=1e-6) and np.all(out[..., 3] == 1.0)
    c = o.counters()
    assert c["primary"] == 256 * 256 and c["shadow"] == 0
    # coverage: triangle area 2 at distance 3 under a 90-degree fov (image plane 6x6 at that distance)
    assert abs(c["closest_hits"] / (256 * 256) - 2.0 / 36.0) < 0.002
    assert c["bounce"] == c["closest_hits"]           # min = max = 1 bounce: every hit bounces exactly once


def test_config1_hit_kind_and_barycentric_uv(oracle_lib):
    s = scenes.single_triangle(128)
    o = oracle_lib.Oracle()
    h = s.upload(o)
    st = copy_settings(s.settings); st.debug_output = abi.DEBUG_OUTPUT_HIT_KIND
    out = render(o, s, h, settings=st)
    hit = out[..., 2] == 0.0                 # the background is the (1,1,1) environment colour
    # camera at (0,-3,0) looking +Y sees the front (normal = cross(e1,e2) = -Y side): red
    assert np.all(out[hit][:, 0] == 1.0) and np.all(out[hit][:, 1] == 0.0)
    st.debug_output = abi.DEBUG_OUTPUT_TEXCOORD_0
    out = render(o, s, h, settings=st)
    uv = out[hit][:, :2]
    assert uv.min() >= -1e-6 and uv.max() <= 1 + 1e-6
    # apex (0,0,1) has uv (0.5, 0): the top-most covered row must have u ~ 0.5, v ~ 0
    rows = np.where(hit.any(axis=1))[0]
    top = out[rows[0]][hit[rows[0]]]
    assert abs(top[:, 0].mean() - 0.5) < 0.05 and top[:, 1].mean() < 0.05


def test_furnace_with_geometry(oracle_lib):
    """Diffuse-white + constant environment + min = max bounces, no NEE: every pixel is exactly the product of weights 1
    times the environment, or 0 when the path is still inside geometry after max bounces."""
    from gltf_renderer_amd import camera, meshgen
    s = scenes.SceneData("furnace")                      # no emissive materials: floor + two spheres + a box
    s.add_mesh(meshgen.grid(4, 4, (-3, -3, 0), (6, 0, 0), (0, 6, 0)), None, 0)
    s.add_mesh(meshgen.uv_sphere(16, 8, 0.7), camera.trs((-0.9, 0.3, 0.7)), 0)
    s.add_mesh(meshgen.uv_sphere(16, 8, 0.5), camera.trs((0.9, -0.2, 0.5), scale=(-1, 1, 1)), 0)
    s.add_mesh(meshgen.box((-0.3, 1.0, 0.0), (0.4, 1.6, 1.2)), None, 0)
    s.world_to_view = camera.orbit_world_to_view((0, 0, 0.5), 3.5, 0.3, -0.4)
    s.width = s.height = 48
    st = abi.PtSettings.defaults()
    st.flags = abi.FLAG_MATERIAL_DIFFUSE_WHITE
    st.min_bounces = st.max_bounces = 5
    st.environment_color[:] = (0.5, 0.25, 1.0)
    st.use_frame_as_seed = 0
    o = oracle_lib.Oracle()
    h = s.upload(o)
    out = render(o, s, h, settings=st)
    px = out[..., :3].reshape(-1, 3)
    env = np.array([0.5, 0.25, 1.0])
    is_env = np.all(np.abs(px - env) < 1e-5, axis=1)
    is_zero = np.all(px == 0, axis=1)
    assert np.all(is_env | is_zero)
    assert is_env.mean() > 0.6


def test_accumulation_state_machine(oracle_lib):
    s = scenes.test_scene(32, 16)
    o = oracle_lib.Oracle()
    h = s.upload(o)
    st = copy_settings(s.settings)
    st.max_accumulated_frames = 3
    out = np.zeros((s.height, s.width, 4), np.float32)
    singles = []
    for f in range(3):
        single = np.zeros_like(out)
        o2 = oracle_lib.Oracle(); h2 = s.upload(o2)
        st1 = copy_settings(st); st1.flags &= ~abi.FLAG_ACCUMULATE
        o2.trace(st1, s.execute_params(frame=f, env_handle=h2["env"]), single)
        singles.append(single[..., :3].astype(np.float64))
        o.trace(st, s.execute_params(frame=f, env_handle=h["env"]), out)
        assert o.counters(reset=False)["accumulated_frames"] == f + 1
    # running mean with weight 1/(n+1) (PathTracer.lib.hlsl:778-785)
    assert np.allclose(out[..., :3], np.mean(singles, axis=0), rtol=1e-4, atol=1e-5)
    # no-op once accumulated_frames >= max_accumulated_frames (Pathtracer.cpp:273)
    before = out.copy()
    o.counters()
    o.trace(st, s.execute_params(frame=3, env_handle=h["env"]), out)
    assert np.array_equal(before, out) and o.counters()["rays"] == 0
    # camera change resets (Pathtracer.cpp:267-271)
    s.world_to_view = s.world_to_view.copy(); s.world_to_view[0, 3] += 0.01
    o.trace(st, s.execute_params(frame=4, env_handle=h["env"]), out)
    assert o.counters(reset=False)["accumulated_frames"] == 1
    # settings.reset resets; without FLAG_ACCUMULATE the counter stays 0
    st.reset = 1
    o.trace(st, s.execute_params(frame=5, env_handle=h["env"]), out)
    assert o.counters(reset=False)["accumulated_frames"] == 1
    st.reset = 0; st.flags &= ~abi.FLAG_ACCUMULATE
    o.trace(st, s.execute_params(frame=6, env_handle=h["env"]), out)
    assert o.counters(reset=False)["accumulated_frames"] == 0


def test_bounce_clamp_matches_reference_limit(oracle_lib):
    s = scenes.test_scene(24, 16)
    st = copy_settings(s.settings); st.min_bounces = 0; st.max_bounces = 50; st.flags &= ~abi.FLAG_ACCUMULATE
    a = oracle_lib.Oracle(); ha = s.upload(a); a.set_bounce_limit(abi.REFERENCE_MAX_BOUNCES)
    st5 = copy_settings(st); st5.max_bounces = 5
    b = oracle_lib.Oracle(); hb = s.upload(b); b.set_bounce_limit(abi.REFERENCE_MAX_BOUNCES)
    ia, ib = render(a, s, ha, settings=st), render(b, s, hb, settings=st5)
    assert np.array_equal(ia, ib)             # max_bounces 50 is clamped to MAX_BOUNCES = 5 (Pathtracer.cpp:323-324)


def test_lbvh_agrees_with_brute_force(oracle_lib):
    s = scenes.test_scene(40, 16)
    o = oracle_lib.Oracle(); h = s.upload(o)
    b = oracle_lib.Oracle(); hb = s.upload(b); b.set_brute_force(True)
    for dbg in (abi.DEBUG_OUTPUT_HIT_KIND, abi.DEBUG_OUTPUT_TEXCOORD_0, abi.DEBUG_OUTPUT_SHADING_NORMAL, abi.DEBUG_OUTPUT_NONE):
        st = copy_settings(s.settings); st.debug_output = dbg; st.flags &= ~abi.FLAG_ACCUMULATE
        i1, i2 = render(o, s, h, settings=st), render(b, s, hb, settings=st)
        # closest hit is order independent except exact-t ties; radiance follows
        assert (np.abs(i1 - i2).max(axis=2) > 1e-6).mean() < 0.002
    nodes, tris = o.bvh_info()
    assert nodes == tris - 1 == s.triangles - 1


def test_tile_shards_compose_bit_exactly(oracle_lib):
    """pixels are independent, so N rank-shards summed == the 1-rank frame, bit for bit."""
    s = scenes.test_scene(50, 16)          # 50 is not a multiple of 16: ragged edge tiles
    o = oracle_lib.Oracle(); h = s.upload(o)
    st = copy_settings(s.settings); st.flags &= ~abi.FLAG_ACCUMULATE
    full = render(o, s, h, settings=st)
    for n in (2, 3, 5):
        acc = np.zeros_like(full)
        for rnk in range(n):
            part = np.zeros_like(full)
            o.trace(st, s.execute_params(frame=0, env_handle=h["env"], tile_rank=rnk, tile_rank_count=n), part)
            assert np.all((part != 0).any(axis=2) <= (full != 0).any(axis=2))
            acc += part
        assert np.array_equal(acc, full)


def test_flags_change_what_they_should(oracle_lib):
    s = scenes.test_scene(32, 16)
    o = oracle_lib.Oracle(); h = s.upload(o)
    base = copy_settings(s.settings); base.flags &= ~abi.FLAG_ACCUMULATE; base.use_frame_as_seed = 0; base.seed = 9
    ref = render(o, s, h, settings=base); c_ref = o.counters()
    st = copy_settings(base); st.flags &= ~abi.FLAG_SHADOW_RAYS
    render(o, s, h, settings=st); c = o.counters()
    assert c["shadow"] < c_ref["shadow"] and c["primary"] == c_ref["primary"]
    st = copy_settings(base); st.flags |= abi.FLAG_INDIRECT_ENVIRONMENT_ONLY
    render(o, s, h, settings=st); c = o.counters()
    assert c["shadow"] == 0 and c["closest_hits"] <= c_ref["primary"]          # bounce rays use instance mask 0: they all miss
    st = copy_settings(base); st.flags |= abi.FLAG_LUMINANCE_CLAMP; st.luminance_clamp = 0.5
    img = render(o, s, h, settings=st)
    lum = img[..., :3] @ np.array([0.2126, 0.7152, 0.0722])
    assert lum.max() <= 0.5 * (1 + 1e-5)
    st = copy_settings(base); st.flags |= abi.FLAG_CULL_BACKFACE
    img = render(o, s, h, settings=st)
    assert not np.array_equal(img, ref)
    st = copy_settings(base); st.max_bounces = 0; st.min_bounces = 0
    o.counters()
    render(o, s, h, settings=st); c = o.counters()
    assert c["bounce"] == 0


def test_golden_tile_regression(oracle_lib):
    """tests/golden/test_scene_32_spp4.npy was written by tools/make_golden.py from this oracle; it pins the oracle
    against accidental change (it is NOT reference output: parity with real DXR is unpinned)."""
    path = os.path.join(GOLDEN, "test_scene_32_spp4.npy")
    gold = np.load(path)
    s = scenes.test_scene(32, 16)
    o = oracle_lib.Oracle(); h = s.upload(o)
    out = render(o, s, h, frames=4)
    assert np.allclose(out, gold, rtol=2e-5, atol=2e-6)


def test_orthographic_camera_projects_in_parallel(oracle_lib):
    """Camera::Orthographic (Camera.h:31-40, GetViewToClip :91: orthoRH_ZO(-1/x_mag, 1/x_mag, -1/y_mag, 1/y_mag, far, near)): the image
    is the parallel projection of the scene onto the view plane, half extents 1 / mag (sic), top of the image = world +Z."""
    s = scenes.single_triangle(128)
    s.ortho = (0.5, 0.5)                                        # half extents 2 x 2
    o = oracle_lib.Oracle(); h = s.upload(o)
    st = copy_settings(s.settings); st.debug_output = abi.DEBUG_OUTPUT_HIT_KIND
    out = render(o, s, h, settings=st)
    hit = ((out[..., 0] == 1) & (out[..., 1] == 0)) | ((out[..., 0] == 0) & (out[..., 1] == 1))      # red front / green back; a miss is the white environment
    # triangle (-1,0,-1) (1,0,-1) (0,0,1) seen along +Y: x in [-2, 2] left to right, z in [2, -2] top to bottom
    ys, xs = np.mgrid[0:128, 0:128]
    x = ((xs + 0.5) / 128 * 2 - 1) * 2.0
    z = -((ys + 0.5) / 128 * 2 - 1) * 2.0
    inside = (z > -1) & (z < 1 - 2 * np.abs(x))
    assert (hit != inside).mean() < 0.01                        # the jittered silhouette pixels only
    assert abs(hit.mean() - 2.0 / 16.0) < 0.003                 # area 2 of a 4 x 4 window, independent of the distance


def test_skin_joint_ids_beyond_the_bone_array_read_zero(oracle_lib):
    """Skin.cs.hlsl:93-101 index StructuredBuffer<Bone> with the vertex's raw joint ids; beyond the buffer D3D12's robust access
    returns zeros, i.e. a zero matrix: the joint contributes nothing (it does not crash and it does not read a neighbour)."""
    s = scenes.skinned_figure(16, 16)
    sk = s.skins[0]
    nv = sk["mesh"].num_vertices
    res = []
    for beyond in (False, True):
        jw = s.buffers[sk["joint_weight"]][0].copy().reshape(-1, 8)
        if beyond:
            jw[:, 2] = 40000                                    # the third joint of every vertex: far beyond the 19 bones
        else:
            jw[:, 6] = 0                                        # reference: the third joint's weight set to zero instead
        s2 = scenes.skinned_figure(16, 16)
        s2.buffers[sk["joint_weight"]] = (jw, abi.FORMAT_JOINT_WEIGHT)
        o = oracle_lib.Oracle(); h = s2.upload(o)
        b = scenes.SkinBinding(o, s2, h, 0, 0)
        b.pose(0.6)
        res.append(o.buffer_read(b.out_position, np.float32, nv * 3))
        o.close()
    assert np.array_equal(res[0], res[1])
