"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI of libmipt.so,
against the CPU oracle on the same seeded inputs.

Tolerances.  Integer / index work (hit kind, ray counters, tile composition, accumulation counter) is compared
exactly or to a stated tiny fraction of flipped pixels.  Floating point: the radiance metric is the one
BASELINE.json states -- relative L2 <= 1e-3 per image after tone mapping, at matched scene / seed / sample
count.  CPU (glibc, no FMA contraction) and GPU (ocml, FMA) differ in the last bits, which flips a discrete
decision (lobe pick, Russian roulette, silhouette edge) in a few pixel-samples per million; those are reported
as a fraction and bounded, not hidden.  PARITY UNPINNED vs real DXR output."""
import ctypes as C

import numpy as np
import pytest

from gltf_renderer_amd import abi, camera, meshgen, scenes

pytestmark = pytest.mark.gpu


def copy_settings(s):
    return abi.PtSettings.from_buffer_copy(bytes(s))


@pytest.fixture(scope="module")
def R():
    from gltf_renderer_amd.renderer import Renderer
    return Renderer


class Pair:
    """The same scene uploaded to the GPU renderer and to the oracle (env maps preprocessed on the GPU and handed
    to the oracle raw, so tracer parity is not polluted by preprocessing differences; those have their own test)."""

    def __init__(self, R, oracle_lib, scene):
        self.s = scene
        self.r = R()
        self.hg = scene.upload(self.r)
        env_raw = self.r.env_read(self.hg["env"]) if self.hg["env"] is not None else None
        self.o = oracle_lib.Oracle()
        self.ho = scene.upload(self.o, env_raw=env_raw)

    def render(self, settings=None, frames=1, first_frame=0, **kw):
        st = settings or self.s.settings
        og = self.r.create_output(self.s.width, self.s.height)
        b = np.zeros((self.s.height, self.s.width, 4), np.float32)
        self.r.reset_stats(); self.o.counters()
        for f in range(first_frame, first_frame + frames):
            self.r.trace(st, self.s.execute_params(frame=f, env_handle=self.hg["env"], **kw), og)
            self.o.trace(st, self.s.execute_params(frame=f, env_handle=self.ho["env"], **kw), b)
        return og, b

    def close(self):
        self.r.close(); self.o.close()


def rel_l2(a, b):
    """Relative L2 over the pixels finite on both sides (AgX ends in pow(x, 2.2) of a value that can be slightly negative for very
    dark pixels, ToneMapper.ps.hlsl:75: NaN upstream, in the oracle and here alike -- those pixels must coincide)."""
    a = a.astype(np.float64); b = b.astype(np.float64)
    fa, fb = np.isfinite(a), np.isfinite(b)
    assert (fa != fb).mean() < 1e-3, float((fa != fb).mean())
    ok = fa & fb
    return float(np.sqrt(((a[ok] - b[ok]) ** 2).sum() / max((b[ok] ** 2).sum(), 1e-30)))


@pytest.fixture(scope="module")
def pair(R, oracle_lib):
    p = Pair(R, oracle_lib, scenes.test_scene(96, 64))
    yield p
    p.close()


def test_single_triangle_config1_exact(R, oracle_lib):
    p = Pair(R, oracle_lib, scenes.single_triangle(256))
    og, b = p.render()
    a = p.r.readback(og)
    assert np.array_equal(a, b)                                  # white furnace: exactly 1 on both sides
    st = p.r.stats(); c = p.o.counters()
    assert (st.rays_primary, st.rays_bounce, st.rays_shadow, st.closest_hits) == (c["primary"], c["bounce"], c["shadow"], c["closest_hits"])
    for dbg in (abi.DEBUG_OUTPUT_HIT_KIND, abi.DEBUG_OUTPUT_TEXCOORD_0, abi.DEBUG_OUTPUT_VERTEX_NORMAL):
        s2 = copy_settings(p.s.settings); s2.debug_output = dbg
        og, b = p.render(settings=s2)
        assert np.abs(p.r.readback(og) - b).max() < 1e-5
    p.close()


@pytest.mark.parametrize("dbg", list(range(1, 28)))
def test_debug_outputs_match(pair, dbg):
    """Deterministic per-pixel outputs (no Monte-Carlo noise beyond the jitter, which is seed matched)."""
    st = copy_settings(pair.s.settings); st.debug_output = dbg; st.flags &= ~abi.FLAG_ACCUMULATE
    st.use_frame_as_seed = 0; st.seed = 5
    og, b = pair.render(settings=st)
    a = pair.r.readback(og)
    err = np.abs(a[..., :3] - b[..., :3]).max(axis=2)
    tol = 2e-4 + 2e-3 * np.abs(b[..., :3]).max(axis=2)          # BSDF / pdf outputs span many decades: relative part
    bad = float((err > tol).mean())
    assert bad <= 0.003, (abi.DEBUG_OUTPUT_NAMES[dbg], bad, float(err.max()))
    assert np.median(err) < 1e-5


def test_radiance_parity_app_defaults(pair):
    import oracle.pyoracle as po
    og, b = pair.render(frames=32)
    ta, tb = pair.r.tonemap(og), po.tonemap(b)
    r = rel_l2(ta, tb)
    assert r <= 1e-3, r
    st = pair.r.stats(); c = pair.o.counters()
    assert int(st.rays) == c["rays"]          # every traversal started, counted on both sides
    assert st.accumulated_frames == c["accumulated_frames"] == 32


@pytest.mark.parametrize("name,set_flags,clear_flags", [
    ("cull_backface", abi.FLAG_CULL_BACKFACE, 0),
    ("alpha_shadows", abi.FLAG_ALPHA_SHADOWS, 0),
    ("indirect_env_only", abi.FLAG_INDIRECT_ENVIRONMENT_ONLY, 0),
    ("geometric_normals", abi.FLAG_MATERIAL_USE_GEOMETRIC_NORMALS, 0),
    ("cosine_only", 0, abi.FLAG_MATERIAL_MIS),
    ("no_env_mis", 0, abi.FLAG_ENVIRONMENT_MIS),
    ("no_shadow_rays", 0, abi.FLAG_SHADOW_RAYS),
    ("no_point_lights", 0, abi.FLAG_POINT_LIGHTS),
    ("no_normal_adaptation", 0, abi.FLAG_SHADING_NORMAL_ADAPTATION),
    ("diffuse_white", abi.FLAG_MATERIAL_DIFFUSE_WHITE, 0),
    ("luminance_clamp", abi.FLAG_LUMINANCE_CLAMP, 0),
])
def test_radiance_parity_flag_matrix(pair, name, set_flags, clear_flags):
    import oracle.pyoracle as po
    st = copy_settings(pair.s.settings)
    st.flags = (st.flags | set_flags) & ~clear_flags
    st.luminance_clamp = 2.0
    st.reset = 1
    og, b = pair.render(settings=st, frames=1)                  # first frame with reset ...
    st.reset = 0
    for f in range(1, 64):                                      # ... then accumulate
        pair.r.trace(st, pair.s.execute_params(frame=f, env_handle=pair.hg["env"]), og)
        pair.o.trace(st, pair.s.execute_params(frame=f, env_handle=pair.ho["env"]), b)
    r = rel_l2(pair.r.tonemap(og), po.tonemap(b))
    print("flag matrix %s: tone-mapped rel L2 %.3e at 64 spp" % (name, r))
    assert r <= 1e-3, (name, r)                                 # the north_star bar (64 spp: one flipped sample weighs 1/64 of a pixel)
    st2 = pair.r.stats(); c = pair.o.counters()
    assert int(st2.rays) == c["rays"], name


def test_constant_environment_no_envmap(R, oracle_lib):
    import oracle.pyoracle as po
    p = Pair(R, oracle_lib, scenes.test_scene(64, 32, with_env=False))
    og, b = p.render(frames=64)
    e = rel_l2(p.r.tonemap(og), po.tonemap(b))
    print("constant environment: tone-mapped rel L2 %.3e at 64 spp" % e)
    assert e <= 1e-3, e
    p.close()


def test_material_grid_deep_bounces(R, oracle_lib):
    """config-4 class (transmission / clearcoat / sheen / anisotropy sweeps), 16 bounces with the clamp lifted."""
    import oracle.pyoracle as po
    # Smooth transmissive spheres chain many discrete decisions (lobe pick, roulette), and a path that reaches the 2e3-radiance sun on one
    # side only leaves a saturated pixel: with an FMA-contracting build this scene sat at 1.1e-3 ... 2.8e-3 (different events each time) and
    # only its bulk met the bar.  Since the path-tracing kernels are built like the oracle -- no floating-point contraction (csrc/Makefile) --
    # the WHOLE image meets it: measured 4.8e-5, and 8e-6 (2.4e-7 without the 16 worst pixels, no pixel beyond 0.05) since sin / cos are
    # evaluated in double and rounded once (pt_math.h), and 1.8e-7 once the oracle defines them the same way (hlsl.h o_sin / o_cos).
    p = Pair(R, oracle_lib, scenes.material_grid(128, seg=12))
    og, b = p.render(frames=96)
    ta, tb = p.r.tonemap(og).astype(np.float64), po.tonemap(b).astype(np.float64)
    ok = np.isfinite(ta).all(axis=2) & np.isfinite(tb).all(axis=2)
    sq = np.where(ok, ((ta - tb) ** 2).sum(axis=2), 0.0).ravel()
    energy = float((tb[ok] ** 2).sum())
    e = float(np.sqrt(sq.sum() / energy))
    k = max(int(round(1e-3 * sq.size)), 1)
    trimmed = float(np.sqrt(np.sort(sq)[:-k].sum() / energy))
    d = np.abs(ta - tb).max(axis=2)[ok]
    bias = float((ta[ok] - tb[ok]).sum() / tb[ok].sum())
    print("material grid, 16 bounces, 128^2 at 96 spp: tone-mapped rel L2 %.3e whole image, %.3e without the %d worst pixels; pixels with |diff| > 0.05: %d; "
          "median |diff| %.2e; relative bias %.2e" % (e, trimmed, k, int((d > 0.05).sum()), float(np.median(d)), bias))
    st = p.r.stats(); c = p.o.counters()
    assert int(st.rays) == c["rays"]
    assert e <= 1e-3, e                                             # the north_star bar, whole image
    # (measured 1.8e-7, printed above.)  Regression guard on the arithmetic the oracle and the kernels define alike, two decades above the measured
    # figure and one below the contract; not a claim about a real DXR driver, which may fuse and approximate as it likes
    assert e <= 1e-4 and trimmed <= 1e-4 and (d > 0.05).sum() == 0 and abs(bias) < 1e-4, (e, trimmed, int((d > 0.05).sum()), float(np.median(d)), bias)
    p.close()


def test_env_preprocessing_matches_oracle(R, oracle_lib):
    """K10-K13 on the GPU vs the oracle's CPU restatement (cube RGBA16F texels and the 1024^2 sum pyramid)."""
    img = scenes.sky_image(512, 256, 1.0e4)
    r = R(); eg = r.env_create(img)
    n, cube, pyr = r.env_read(eg)
    o = oracle_lib.Oracle(); eo = o.env_create(img)
    n2, cube2, pyr2 = o.env_read(eo)
    assert n == n2 == 65
    # built without contraction (csrc/Makefile) the four kernels reproduce the oracle bit for bit: every RGBA16F cube texel, every entry of
    # the 1024^2 map and of its ten-level sum pyramid (with an FMA-contracting build 2 % of the texels differed in the last bit)
    assert np.array_equal(cube, cube2), int((cube != cube2).sum())
    assert np.array_equal(pyr.view(np.uint32), pyr2.view(np.uint32)), int((pyr.view(np.uint32) != pyr2.view(np.uint32)).sum())
    r.close(); o.close()


def test_gpu_lbvh_closest_hits_match_bruteforce(R, oracle_lib):
    """The on-device LBVH + traversal must return the same closest hits as a brute-force loop over all triangles."""
    s = scenes.test_scene(128, 16)
    r = R(); hg = s.upload(r)
    o = oracle_lib.Oracle(); ho = s.upload(o, env_raw=r.env_read(hg["env"])); o.set_brute_force(True)
    for dbg in (abi.DEBUG_OUTPUT_HIT_KIND, abi.DEBUG_OUTPUT_TEXCOORD_0, abi.DEBUG_OUTPUT_VERTEX_NORMAL):
        st = copy_settings(s.settings); st.debug_output = dbg; st.flags &= ~abi.FLAG_ACCUMULATE
        og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
        r.trace(st, s.execute_params(0, env_handle=hg["env"]), og)
        o.trace(st, s.execute_params(0, env_handle=ho["env"]), b)
        err = np.abs(r.readback(og)[..., :3] - b[..., :3]).max(axis=2)
        assert (err > 0).sum() == 0             # (round 3: the tree, the oracle's tree and its exhaustive search return the same hits)
    st = r.stats()
    # wide nodes collapsed greedily from the n-1 binary LBVH nodes
    assert st.bvh_triangles == s.triangles and s.triangles // 32 <= st.bvh_nodes <= s.triangles - 1      # wide nodes (<= 8 children) over leaves of <= 3 triangles
    r.close(); o.close()


def test_accumulation_semantics_on_gpu(R):
    s = scenes.test_scene(48, 16)
    r = R(); h = s.upload(r)
    st = copy_settings(s.settings); st.max_accumulated_frames = 2
    out = r.create_output(s.width, s.height)
    for f in range(2):
        r.trace(st, s.execute_params(f, env_handle=h["env"]), out)
    assert r.stats().accumulated_frames == 2
    before = r.readback(out)
    r.reset_stats()
    r.trace(st, s.execute_params(2, env_handle=h["env"]), out)      # no-op at max_accumulated_frames (Pathtracer.cpp:273)
    assert np.array_equal(before, r.readback(out)) and r.stats().rays == 0
    s.world_to_view = s.world_to_view.copy(); s.world_to_view[1, 3] += 0.02
    r.trace(st, s.execute_params(3, env_handle=h["env"]), out)      # camera moved -> reset
    assert r.stats().accumulated_frames == 1
    st.flags &= ~abi.FLAG_ACCUMULATE
    r.trace(st, s.execute_params(4, env_handle=h["env"]), out)
    assert r.stats().accumulated_frames == 0
    # same seed twice -> identical bits (determinism)
    a = r.create_output(s.width, s.height); b = r.create_output(s.width, s.height)
    r.trace(st, s.execute_params(7, env_handle=h["env"]), a); r.trace(st, s.execute_params(7, env_handle=h["env"]), b)
    assert np.array_equal(r.readback(a), r.readback(b))
    r.close()


@pytest.mark.parametrize("mode", [abi.MODE_WAVEFRONT, abi.MODE_MEGAKERNEL])
def test_sample_batch_equals_the_same_frames_one_by_one(R, mode):
    """pt_set_samples_per_trace: one launch carrying S samples must leave exactly the bits S consecutive calls leave --
    per-sample seeds (frame + k), blend order, accumulated_frames, the max_accumulated_frames cap, tile shards."""
    s = scenes.test_scene(72, 16)           # 4.5 tiles: ragged edge
    r = R(); h = s.upload(r)
    r.set_kernel_mode(mode)
    for use_frame, shard in ((1, (0, 1)), (0, (0, 1)), (1, (1, 3))):
        st = copy_settings(s.settings); st.use_frame_as_seed = use_frame; st.seed = 99; st.reset = 1
        kw = dict(env_handle=h["env"], tile_rank=shard[0], tile_rank_count=shard[1])
        one = r.create_output(s.width, s.height); many = r.create_output(s.width, s.height)
        r.set_samples_per_trace(1)
        for f in range(10, 17):                                     # 7 frames, one call each
            r.trace(st, s.execute_params(f, **kw), one); st.reset = 0
        rays_one = None
        r.reset_stats()
        st.reset = 1
        r.set_samples_per_trace(4)
        r.trace(st, s.execute_params(10, **kw), many); st.reset = 0  # frames 10..13
        assert r.stats().accumulated_frames == 4
        r.set_samples_per_trace(3)
        r.trace(st, s.execute_params(14, **kw), many)               # frames 14..16
        assert r.stats().accumulated_frames == 7
        assert np.array_equal(r.readback(one), r.readback(many)), (use_frame, shard)
    # the batch stops at max_accumulated_frames (the calls beyond it would have been no-ops, Pathtracer.cpp:273)
    st = copy_settings(s.settings); st.max_accumulated_frames = 5; st.reset = 1
    a = r.create_output(s.width, s.height); b = r.create_output(s.width, s.height)
    r.set_samples_per_trace(1)
    for f in range(5):
        r.trace(st, s.execute_params(f, env_handle=h["env"]), a); st.reset = 0
    st.reset = 1
    r.set_samples_per_trace(8)
    r.trace(st, s.execute_params(0, env_handle=h["env"]), b)
    assert r.stats().accumulated_frames == 5 and np.array_equal(r.readback(a), r.readback(b))
    # without accumulation the batch is ignored
    st = copy_settings(s.settings); st.flags &= ~abi.FLAG_ACCUMULATE
    r.set_samples_per_trace(1); r.trace(st, s.execute_params(3, env_handle=h["env"]), a)
    r.set_samples_per_trace(4); r.reset_stats(); r.trace(st, s.execute_params(3, env_handle=h["env"]), b)
    assert np.array_equal(r.readback(a), r.readback(b))
    from gltf_renderer_amd.renderer import MiptError
    with pytest.raises(MiptError):
        r.set_samples_per_trace(0)
    r.close()


@pytest.mark.parametrize("mode", [abi.MODE_WAVEFRONT, abi.MODE_MEGAKERNEL])
def test_null_shadow_culling_keeps_the_image_and_drops_rays(R, mode):
    """pt_set_null_shadow_culling: a shadow ray whose pending term is exactly zero adds T * 0; not tracing it must leave every bit
    of the image alone, lower rays_shadow and leave primary / bounce counts alone.  Off by default (reference ray counts)."""
    s = scenes.test_scene(96, 64)           # spot + point + directional lights, env map, alpha-tested and transmissive materials
    r = R(); h = s.upload(r)
    r.set_kernel_mode(mode)
    imgs, rays = [], []
    for cull in (0, 1, 0):
        r.set_null_shadow_culling(cull)
        out = r.create_output(s.width, s.height)
        st = copy_settings(s.settings); st.reset = 1
        r.reset_stats()
        for f in range(3):
            r.trace(st, s.execute_params(f, env_handle=h["env"]), out); st.reset = 0
        q = r.stats()
        imgs.append(r.readback(out)); rays.append((q.rays_primary, q.rays_bounce, q.rays_shadow))
    assert np.array_equal(imgs[0], imgs[1]) and np.array_equal(imgs[0], imgs[2])
    assert rays[0] == rays[2]
    assert rays[1][:2] == rays[0][:2] and rays[1][2] < rays[0][2]
    r.close()


def test_tile_shards_compose_bit_exactly_on_gpu(R):
    s = scenes.test_scene(200, 16)          # 200 = 12.5 tiles: ragged edges
    r = R(); h = s.upload(r)
    st = copy_settings(s.settings); st.flags &= ~abi.FLAG_ACCUMULATE
    full = r.create_output(s.width, s.height)
    r.trace(st, s.execute_params(1, env_handle=h["env"]), full)
    full_h = r.readback(full)
    for n in (2, 3, 8):
        acc = np.zeros_like(full_h)
        for rank in range(n):
            part = r.create_output(s.width, s.height)
            r.trace(st, s.execute_params(1, env_handle=h["env"], tile_rank=rank, tile_rank_count=n), part)
            acc += r.readback(part)
        assert np.array_equal(acc, full_h)
    r.close()


def test_tonemap_matches_oracle(pair):
    import oracle.pyoracle as po
    og, b = pair.render(frames=2)
    a = pair.r.readback(og)
    for tm in (abi.TONEMAPPER_NONE, abi.TONEMAPPER_AGX):
        for dither in (0, 1):
            cfg = abi.PtTonemapConfig(tm, 1.3, 7, dither)
            rgb_g, q_g = pair.r.tonemap(og, cfg, want_rgba8=True)
            rgb_o, q_o = po.tonemap(a, cfg, want_rgba8=True)          # same input image: isolates the tone mapper
            # log2 and pow are defined functions since round 3 (the AgX curve, the display transfer): the same bits as the oracle's for the same input
            print("tone mapper %d dither %d: floats not bit-identical %d of %d, rgba8 different %d" % (tm, dither, int((rgb_g.view(np.uint32) != rgb_o.view(np.uint32)).sum()), rgb_g.size, int((q_g != q_o).sum())))
            assert np.array_equal(rgb_g.view(np.uint32), rgb_o.view(np.uint32)) and np.array_equal(q_g, q_o)


def _skin_setup(backend, s, use_mfma, t):
    h = s.upload(backend)
    bind = scenes.SkinBinding(backend, s, h, 0, use_mfma)
    bind.pose(t)
    mesh, p, out_pos, out_ts = s.skins[0]["mesh"], bind.params, bind.out_position, bind.out_tangent_space
    return (backend.buffer_read(out_pos, np.float32, mesh.num_vertices * 3).reshape(-1, 3), backend.buffer_read(out_ts, np.uint32, mesh.num_vertices),
            h, p, out_pos, out_ts)


@pytest.mark.parametrize("use_mfma", [0, 1])
def test_gpu_skin_matches_oracle(R, oracle_lib, use_mfma):
    """GpuSkin::Run: per-vertex VALU blend and the v_mfma_f32_16x16x4_f32 joint-matrix blend vs Skin.cs.hlsl restated."""
    s = scenes.skinned_figure(64, 36)
    r = R(); o = oracle_lib.Oracle()
    pg, tg, *_ = _skin_setup(r, s, use_mfma, 0.37)
    po_, to_, *_ = _skin_setup(o, s, 0, 0.37)
    assert np.abs(pg - po_).max() < 2e-6 * max(1.0, np.abs(po_).max()) * (8 if use_mfma else 1)
    # packed tangent space: 10-bit fields may differ by one quantisation step where a value sits on a rounding edge
    d = [np.abs(((tg >> sh) & 0x3ff).astype(int) - ((to_ >> sh) & 0x3ff).astype(int)) for sh in (0, 10)]
    assert max(x.max() for x in d) <= 1 and np.mean(tg == to_) > 0.97
    ang = np.abs(((tg >> 20) & 0x3ff).astype(int) - ((to_ >> 20) & 0x3ff).astype(int)); ang = np.minimum(ang, 1023 - ang)
    assert ang.max() <= 2 and np.all((tg >> 30) == (to_ >> 30))
    # moved vertices: the pose is not the bind pose
    rest = s.skins[0]["mesh"].positions
    assert np.abs(pg - rest).max() > 0.05
    r.close(); o.close()


def test_skinned_frame_renders_like_oracle(R, oracle_lib):
    """config-5 class at test size: skin -> BVH rebuild -> trace, dynamic instance pointing at the skinned streams."""
    import oracle.pyoracle as po
    s = scenes.skinned_figure(96, 54)
    outs = []
    for backend in (R(), oracle_lib.Oracle()):
        pos, ts, h, p, out_pos, out_ts = _skin_setup(backend, s, 0, 0.8)
        outs.append((backend, h))
    (r, hg), (o, ho) = outs
    o2 = oracle_lib.Oracle()      # oracle must see the GPU-preprocessed env: re-create with raw maps
    og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
    env_raw = r.env_read(hg["env"])
    eo = o.env_create_raw(*env_raw)
    for f in range(64):
        r.trace(s.settings, s.execute_params(f, env_handle=hg["env"]), og)
        o.trace(s.settings, s.execute_params(f, env_handle=eo), b)
    e = rel_l2(r.tonemap(og), po.tonemap(b))
    print("skinned frame: tone-mapped rel L2 %.3e at 64 spp" % e)
    assert e <= 1e-3, e
    r.close(); o.close(); o2.close()


def test_argument_validation_returns_errors_not_faults(R):
    from gltf_renderer_amd.renderer import MiptError
    r = R()
    s = scenes.single_triangle(16)
    h = s.upload(r)
    bad = [abi.PtInstanceDesc.from_buffer_copy(bytes(d)) for d in h["instances"]]
    bad[0].gpu.position_descriptor = 9999
    with pytest.raises(MiptError):
        r.set_instances(bad)
    bad[0].gpu.position_descriptor = h["instances"][0].gpu.position_descriptor
    bad[0].num_of_indices = 3000                                       # more indices than the stream holds
    with pytest.raises(MiptError):
        r.set_instances(bad)
    bad[0].num_of_indices = 3; bad[0].gpu.material_id = 5
    with pytest.raises(MiptError):
        r.set_instances(bad)
    m = abi.PtMaterial.default(); m.albedo.descriptor = 12
    with pytest.raises(MiptError):
        r.set_materials([m])
    out = r.create_output(16, 16)
    p = s.execute_params(0); p.light_count = 3                          # no lights uploaded
    with pytest.raises(MiptError):
        r.trace(s.settings, p, out)
    with pytest.raises(MiptError):
        r.set_instances([abi.PtInstanceDesc() for _ in range(abi.MAX_TLAS_INSTANCES + 1)])
    r.close()


def test_edge_cases_empty_tiny_and_ragged(R, oracle_lib):
    """Empty and ragged inputs: a scene without geometry, the smallest trees (1, 2, 3 and 5 triangles: leaf root, one node, one node
    with a 3-triangle leaf, two levels), frames that are not multiples of the 16x16 tile or are smaller than one wave."""
    f32 = np.float32
    def tri_scene(n_tris, w, h):
        s = scenes.single_triangle(16)
        s.instances.clear(); s.mesh_records.clear(); s.buffers.clear(); s.triangles = 0
        for k in range(n_tris):
            z = 0.4 * k
            m = meshgen.Mesh(np.array([[-1 + 0.1 * k, 0.2 * k, -1 + z], [1, 0.2 * k, -1 + z], [0, 0.2 * k, 0.2 + z]], f32), np.array([0, 1, 2]),
                             normals=np.array([[0, -1, 0]] * 3, f32), uv0=np.array([[0, 1], [1, 1], [0.5, 0]], f32))
            s.add_mesh(m, None, 0)
        s.width, s.height = w, h
        return s
    cases = [(0, 40, 24), (1, 1, 1), (2, 17, 3), (3, 33, 65), (5, 16, 16), (5, 7, 130)]
    for n_tris, w, h in cases:
        p = Pair(R, oracle_lib, tri_scene(n_tris, w, h))
        for dbg in (abi.DEBUG_OUTPUT_NONE, abi.DEBUG_OUTPUT_HIT_KIND, abi.DEBUG_OUTPUT_TEXCOORD_0):
            st = copy_settings(p.s.settings); st.debug_output = dbg
            og, b = p.render(settings=st)
            a = p.r.readback(og)
            assert a.shape == b.shape == (h, w, 4)
            assert np.abs(a - b).max() < 1e-5, (n_tris, w, h, dbg)
        sg, so = p.r.stats(), p.o.counters()
        assert (sg.rays_primary, sg.rays_bounce, sg.rays_shadow) == (so["primary"], so["bounce"], so["shadow"]), (n_tris, w, h)
        assert sg.bvh_triangles == n_tris
        p.close()


def test_tables_larger_than_the_lds_caches(R, oracle_lib):
    """The shade stage keeps up to 96 materials, 128 instance rows and 32 lights in LDS and falls back to memory beyond that:
    a scene with 150 instances, 110 materials and 40 lights must still match the oracle (both sides of every limit are hit)."""
    f32 = np.float32
    rng = np.random.default_rng(11)
    s = scenes.test_scene(96, 32)
    n_extra_mat = 110 - len(s.materials)
    for k in range(n_extra_mat):
        s.add_material(scenes.material(base_color_factor=tuple(rng.uniform(0.2, 1.0, 3)) + (1.0,), roughness_factor=float(rng.uniform(0.2, 1.0)),
                                       metalness_factor=float(rng.uniform(0.0, 1.0))))
    for k in range(150 - len(s.instances)):
        c = np.array([rng.uniform(-2.5, 2.5), rng.uniform(-1.0, 3.0), rng.uniform(0.05, 2.5)], f32)
        a, b = rng.normal(0, 0.25, 3).astype(f32), rng.normal(0, 0.25, 3).astype(f32)
        n = np.cross(a, b); n = (n / max(np.linalg.norm(n), 1e-6)).astype(f32)
        m = meshgen.Mesh(np.stack([c, c + a, c + b]).astype(f32), np.array([0, 1, 2]), normals=np.stack([n] * 3), uv0=np.array([[0, 0], [1, 0], [0, 1]], f32))
        T = np.eye(4); T[:3, 3] = rng.uniform(-0.2, 0.2, 3)
        s.add_mesh(m, T, int(rng.integers(1, len(s.materials))))
    while len(s.lights) < 40:
        s.add_light(abi.LIGHT_POINT, position=tuple(rng.uniform(-3, 3, 3) + np.array([0, 0, 3.0])), color=tuple(rng.uniform(0.3, 1.0, 3)), intensity=float(rng.uniform(2, 10)))
    assert len(s.instances) >= 150 and len(s.materials) >= 110 and len(s.lights) == 40
    p = Pair(R, oracle_lib, s)
    for dbg in (abi.DEBUG_OUTPUT_COLOR, abi.DEBUG_OUTPUT_ROUGHNESS, abi.DEBUG_OUTPUT_VERTEX_NORMAL):
        st = copy_settings(s.settings); st.debug_output = dbg; st.flags &= ~abi.FLAG_ACCUMULATE; st.use_frame_as_seed = 0; st.seed = 4
        og, b = p.render(settings=st)
        err = np.abs(p.r.readback(og)[..., :3] - b[..., :3]).max(axis=2)
        assert (err > 0).sum() == 0, (abi.DEBUG_OUTPUT_NAMES[dbg], int((err > 0).sum()))
    import oracle.pyoracle as po
    og, b = p.render(frames=64)
    e = rel_l2(p.r.tonemap(og), po.tonemap(b))
    print("tables beyond the LDS caches: tone-mapped rel L2 %.3e at 64 spp" % e)
    sg, so = p.r.stats(), p.o.counters()
    assert int(sg.rays) == so["rays"]
    assert e <= 1e-3, e
    p.close()


# ---- BASELINE-size property tests (no oracle at this size: size-independent identities) --------------------------
@pytest.fixture(scope="module")
def sponza(R):
    s = scenes.sponza_class(tex=256)
    r = R(); h = s.upload(r)
    yield s, r, h
    r.close()


def test_fullsize_linearity_in_environment_intensity(sponza):
    """Radiance is linear in the environment: with punctual lights off, doubling environment_intensity doubles every
    pixel exactly (power-of-two scaling commutes with every fp32 operation on the path)."""
    s, r, h = sponza
    st = copy_settings(s.settings); st.flags &= ~(abi.FLAG_ACCUMULATE | abi.FLAG_POINT_LIGHTS); st.use_frame_as_seed = 0; st.seed = 11
    a = r.create_output(s.width, s.height); b = r.create_output(s.width, s.height)
    st.environment_intensity = 1.0; r.trace(st, s.execute_params(0, env_handle=h["env"]), a)
    st.environment_intensity = 2.0; r.trace(st, s.execute_params(0, env_handle=h["env"]), b)
    ia, ib = r.readback(a), r.readback(b)
    assert np.array_equal(ia[..., :3] * 2.0, ib[..., :3])
    assert ia[..., :3].mean() > 0.01


def test_fullsize_scene_hits_and_radiance_match_the_oracle(R, oracle_lib):
    """The 257 k-triangle scene goes through the large-scene build (level-per-launch collapse) and deep quantised-node traversal,
    which the small parity scenes do not reach: same hits and same radiance as the oracle's own CPU BVH on a reduced frame."""
    s = scenes.sponza_class(width=320, height=180, tex=64)
    p = Pair(R, oracle_lib, s)
    for dbg in (abi.DEBUG_OUTPUT_HIT_KIND, abi.DEBUG_OUTPUT_TEXCOORD_0, abi.DEBUG_OUTPUT_VERTEX_NORMAL):
        st = copy_settings(s.settings); st.debug_output = dbg; st.flags &= ~abi.FLAG_ACCUMULATE; st.use_frame_as_seed = 0; st.seed = 3
        og, b = p.render(settings=st)
        err = np.abs(p.r.readback(og)[..., :3] - b[..., :3]).max(axis=2)
        assert (err > 0).sum() == 0, (abi.DEBUG_OUTPUT_NAMES[dbg], int((err > 0).sum()))
    # Radiance per pixel-sample, seed matched.  The primary hits are identical (above).  This scene amplifies rounding -- it tiles its textures
    # (texture coordinates of tens of units) and glossy lobes turn a perturbed shading normal into a different path -- and an FMA-contracting
    # build parted from the plain-arithmetic oracle in 0.3 % of the pixel-samples at one bounce and 2 % at four.  Built without contraction
    # (csrc/Makefile) the kernels agreed in every pixel-sample at one bounce and in all but 0.04 % at four; with sin / cos evaluated in
    # double and rounded once, like the oracle's libm (pt_math.h), in every pixel-sample at four bounces too.  Counted and bounded.
    for mb, frac_1pc in ((1, 1e-5), (4, 1e-5)):
        st = copy_settings(s.settings); st.max_bounces = mb; st.min_bounces = min(st.min_bounces, mb); st.flags &= ~abi.FLAG_ACCUMULATE
        st.use_frame_as_seed = 0; st.seed = 9
        og, b = p.render(settings=st)
        a = p.r.readback(og)[..., :3].astype(np.float64); bb = b[..., :3].astype(np.float64)
        rel = np.abs(a - bb).max(axis=2) / np.maximum(np.abs(bb).max(axis=2), 1e-6)
        print("full-size scene, max_bounces %d: pixel-samples beyond 1e-2 of the oracle's: %.5f, median relative difference %.2e" % (mb, float((rel > 1e-2).mean()), float(np.median(rel))))
        assert np.median(rel) < 1e-6 and (rel > 1e-2).mean() < frac_1pc, (mb, float(np.median(rel)), float((rel > 1e-2).mean()))
        sg, so = p.r.stats(), p.o.counters()
        assert int(sg.rays) == so["rays"], mb
    assert sg.bvh_triangles - 1 > 32768 and sg.bvh_nodes > 8192       # more radix-tree nodes than the single-launch collapse takes
    p.close()


def test_fullsize_shards_and_determinism(sponza):
    s, r, h = sponza
    st = copy_settings(s.settings); st.flags &= ~abi.FLAG_ACCUMULATE
    full = r.create_output(s.width, s.height)
    r.reset_stats(); r.trace(st, s.execute_params(5, env_handle=h["env"]), full)
    rays_full = r.stats().rays
    full_h = r.readback(full)
    acc = np.zeros_like(full_h); r.reset_stats()
    for rank in range(8):
        part = r.create_output(s.width, s.height)
        r.trace(st, s.execute_params(5, env_handle=h["env"], tile_rank=rank, tile_rank_count=8), part)
        acc += r.readback(part)
    assert np.array_equal(acc, full_h)                     # 8 shards == 1 frame, bit for bit
    assert r.stats().rays == rays_full                     # and exactly the same rays were traced
    assert np.isfinite(full_h).all()
    # rays per pixel-sample bound: 1 primary + 8 bounce + 8 env-shadow + 9 light-shadow
    assert rays_full <= 26 * s.width * s.height and rays_full >= s.width * s.height
