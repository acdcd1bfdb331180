"""GPU tests of the round-2 work (run with -m gpu on an MI355X), all through the C-ABI of libmipt.so:

  * BVH refit (UpdateDynamicBlas) against a full rebuild, on a 257 k-triangle static scene with one skinned figure;
  * instance-table diffing (identical table -> nothing, moved instance -> refit), resource destroy calls, two kinds of misuse;
  * the exchange of the sharded renderer: tile pack / unpack, and pt_exchange_frame on a world of one;
  * BASELINE.json configs that round 1 never ran on the GPU: config 2 (helmet class) against the oracle, config 3 at its own
    8 bounces against the oracle, configs 4 and 5 at full size through size-independent properties, the megakernel mode
    against the oracle, FLAG_SHOW_NAN / FLAG_SHOW_INF and the orthographic camera on both sides.

Radiance tolerance: the north_star figure, relative L2 <= 1e-3 per image after tone mapping at matched seeds and sample count.
PARITY UNPINNED vs real DXR output: the checker is the CPU restatement of the reference's HLSL."""
import numpy as np
import pytest

from gltf_renderer_amd import abi, camera, meshgen, scenes

pytestmark = pytest.mark.gpu

f32 = np.float32


def copy_settings(s):
    return abi.PtSettings.from_buffer_copy(bytes(s))


def rel_l2(a, b, nan_mismatch=1e-3):
    """Relative L2 over the pixels that are finite on both sides.  The AgX curve ends in pow(x, 2.2) of a value that can be slightly
    negative for very dark pixels (ToneMapper.ps.hlsl:75): NaN upstream, in the oracle and here alike -- those pixels must coincide."""
    a = a.astype(np.float64); b = b.astype(np.float64)
    fa, fb = np.isfinite(a), np.isfinite(b)
    assert (fa != fb).mean() < nan_mismatch, float((fa != fb).mean())
    ok = fa & fb
    return float(np.sqrt(((a[ok] - b[ok]) ** 2).sum() / max((b[ok] ** 2).sum(), 1e-30)))


@pytest.fixture(scope="module")
def R():
    from gltf_renderer_amd.renderer import Renderer
    return Renderer


class Pair:
    """The same scene on the GPU renderer and on the oracle (env maps preprocessed on the GPU and handed over raw)."""

    def __init__(self, R, oracle_lib, scene):
        self.s = scene
        self.r = R()
        self.hg = scene.upload(self.r)
        env_raw = self.r.env_read(self.hg["env"]) if self.hg["env"] is not None else None
        self.o = oracle_lib.Oracle()
        self.ho = scene.upload(self.o, env_raw=env_raw)

    def render(self, settings=None, frames=1, first_frame=0):
        st = settings or self.s.settings
        og = self.r.create_output(self.s.width, self.s.height)
        b = np.zeros((self.s.height, self.s.width, 4), np.float32)
        self.r.reset_stats(); self.o.counters()
        for f in range(first_frame, first_frame + frames):
            self.r.trace(st, self.s.execute_params(frame=f, env_handle=self.hg["env"]), og)
            self.o.trace(st, self.s.execute_params(frame=f, env_handle=self.ho["env"]), b)
        return og, b

    def close(self):
        self.r.close(); self.o.close()


def debug_image(r, s, h, dbg, seed=3):
    st = copy_settings(s.settings); st.debug_output = dbg; st.flags &= ~abi.FLAG_ACCUMULATE; st.use_frame_as_seed = 0; st.seed = seed
    out = r.create_output(s.width, s.height)
    r.trace(st, s.execute_params(0, env_handle=h["env"]), out)
    return r.readback(out)


def radiance_image(r, s, h, frames=2, seed_base=40):
    st = copy_settings(s.settings); st.reset = 1
    out = r.create_output(s.width, s.height)
    r.reset_stats()
    for f in range(frames):
        r.trace(st, s.execute_params(seed_base + f, env_handle=h["env"]), out); st.reset = 0
    return r.readback(out), r.stats()


# ---- BVH refit -------------------------------------------------------------------------------------------------------------
def test_refit_matches_full_rebuild_static_scene_plus_skinned_figure(R):
    """A 257 k-triangle static background with one skinned character: after the pose changes, pt_build_accel refits (packets of
    the figure rewritten, every box re-derived, topology kept) instead of rebuilding.  The hits must be those of a tree built from
    scratch for the new pose; only exact-t ties on shared edges may resolve differently (the visiting order differs)."""
    s = scenes.sponza_with_figure(320, 180, tex=64)
    r = R(); h = s.upload(r)
    bind = scenes.SkinBinding(r, s, h, 0, use_mfma=1)
    bind.pose(0.0)
    r.build_accel()
    st0 = r.stats()
    assert st0.accel_builds == 1 and st0.accel_refits == 0 and st0.bvh_triangles == s.triangles
    assert 0 < st0.bvh_stack_need <= 64
    build_ms = st0.accel_ms
    refit_ms = []
    for t in (0.37, 0.9):
        builds_before = r.stats().accel_builds
        bind.pose(t)
        r.build_accel()                                        # -> refit
        q = r.stats()
        refit_ms.append(q.accel_ms)
        assert q.accel_builds == builds_before, "a pose change must not trigger a full build"
        dbg_refit = [debug_image(r, s, h, d) for d in (abi.DEBUG_OUTPUT_HIT_KIND, abi.DEBUG_OUTPUT_TEXCOORD_0, abi.DEBUG_OUTPUT_VERTEX_NORMAL)]
        rad_refit, s_refit = radiance_image(r, s, h)
        r.request_rebuild()
        r.build_accel()                                        # -> full build of the same pose
        q2 = r.stats()
        assert q2.accel_builds == q.accel_builds + 1
        rebuild_ms = q2.accel_ms
        dbg_full = [debug_image(r, s, h, d) for d in (abi.DEBUG_OUTPUT_HIT_KIND, abi.DEBUG_OUTPUT_TEXCOORD_0, abi.DEBUG_OUTPUT_VERTEX_NORMAL)]
        rad_full, s_full = radiance_image(r, s, h)
        for a, b in zip(dbg_refit, dbg_full):
            assert np.array_equal(a, b)        # (round 3: a ray's hit does not depend on the tree -- box gate + equal-distance rule, pt_traverse.h)
        assert np.array_equal(rad_refit, rad_full), float((np.abs(rad_refit - rad_full).max(axis=2) > 0).mean())
        assert int(s_refit.rays) == int(s_full.rays)
        # the figure is in the picture and it moved: the refitted frames of the two poses are different images
    print("accel: first build %.3f ms, rebuild %.3f ms, refits %s ms (%d triangles)" % (build_ms, rebuild_ms, ["%.3f" % x for x in refit_ms], s.triangles))
    assert max(refit_ms) < 0.6 * rebuild_ms, (refit_ms, rebuild_ms)
    assert r.stats().accel_refits == 2
    r.close()


def test_builders_agree_and_ploc_visits_fewer_nodes(R):
    """pt_set_accel_builder: the radix tree, the PLOC tree and the PLOC tree after the reinsertion passes are different trees over
    the same triangles -- identical closest hits (bit-identical debug images: the triangle test does not know the tree), identical
    ray counts, and on the config-4 class scene (spheres over a floor: where spatial-median splits are at their worst) clearly
    fewer node visits per ray with PLOC, fewer again after reinsertion."""
    s = scenes.material_grid(256, seg=24)
    imgs, per_ray, rays = {}, {}, {}
    for name, b in (("lbvh", abi.BUILDER_LBVH), ("ploc", abi.BUILDER_PLOC), ("reins", abi.BUILDER_PLOC_REINSERT)):
        r = R()
        if b != abi.BUILDER_PLOC_REINSERT: r.set_accel_builder(b)              # (the default: a fresh context must already use it)
        h = s.upload(r)
        imgs[name] = [debug_image(r, s, h, d) for d in (abi.DEBUG_OUTPUT_HIT_KIND, abi.DEBUG_OUTPUT_TEXCOORD_0, abi.DEBUG_OUTPUT_VERTEX_NORMAL)]
        q = r.stats()
        assert q.bvh_triangles == s.triangles and 0 < q.bvh_stack_need <= 64 and q.accel_builds == 1
        r.enable_counters(True); r.reset_stats()
        st = copy_settings(s.settings); st.flags &= ~abi.FLAG_ACCUMULATE; st.use_frame_as_seed = 0; st.seed = 17
        out = r.create_output(s.width, s.height)
        r.trace(st, s.execute_params(0, env_handle=h["env"]), out)
        c = r.stats()
        per_ray[name] = c.nodes_visited / c.rays; rays[name] = int(c.rays)
        imgs[name].append(r.readback(out))
        # switching the builder on a live context rebuilds; the picture stays
        r.enable_counters(False)
        r.set_accel_builder(abi.BUILDER_LBVH if b == abi.BUILDER_PLOC else abi.BUILDER_PLOC)
        again = debug_image(r, s, h, abi.DEBUG_OUTPUT_TEXCOORD_0)
        assert r.stats().accel_builds == 2 and np.array_equal(again, imgs[name][1])
        r.close()
    for other in ("ploc", "reins"):
        for a, b in zip(imgs["lbvh"][:3], imgs[other][:3]):
            assert np.array_equal(a, b)                                   # three different trees, the same hits: bit-identical debug images ...
        assert np.array_equal(imgs["lbvh"][3], imgs[other][3]) and rays["lbvh"] == rays[other]      # ... and radiance, ray for ray
    print("nodes per ray, config-4 class: radix tree %.2f, PLOC %.2f, PLOC + reinsertion %.2f" % (per_ray["lbvh"], per_ray["ploc"], per_ray["reins"]))
    assert per_ray["ploc"] < 0.9 * per_ray["lbvh"] and per_ray["reins"] < 0.98 * per_ray["ploc"]


def test_coincident_and_garbage_triangles_build_or_fail_loudly(R):
    """Thousands of triangles with one centroid give every Morton code the same value: the radix tree degenerates into a chain that
    no traversal stack can serve -- the build must refuse it (PT_ERR_CAPACITY), never drop geometry silently -- while PLOC pairs
    equal boxes by index and stays balanced.  A mesh with NaN / huge vertices must not hang either builder."""
    from gltf_renderer_amd.renderer import MiptError
    base = scenes.single_triangle(64)
    tri = np.array([[-1, 0, -1], [1, 0, -1], [0, 0, 1]], f32)
    n = 6000
    pos = np.tile(tri, (n, 1))
    many = meshgen.Mesh(pos, np.arange(3 * n), normals=np.tile(np.array([[0, -1, 0]], f32), (3 * n, 1)), uv0=np.tile(np.array([[0, 1], [1, 1], [0.5, 0]], f32), (n, 1)))
    ref_r = R(); ref_h = base.upload(ref_r)
    ref = debug_image(ref_r, base, ref_h, abi.DEBUG_OUTPUT_HIT_KIND)
    ref_r.close()
    s = scenes.single_triangle(64)
    s.instances.clear(); s.mesh_records.clear(); s.buffers.clear(); s.triangles = 0
    s.add_mesh(many, None, 0)
    for b in (abi.BUILDER_PLOC, abi.BUILDER_PLOC_REINSERT):
        r = R(); r.set_accel_builder(b); h = s.upload(r)
        img = debug_image(r, s, h, abi.DEBUG_OUTPUT_HIT_KIND)
        q = r.stats()
        assert q.bvh_triangles == n and q.bvh_stack_need <= 64
        assert np.array_equal(img, ref)                         # the same silhouette, whichever copy is hit
        if b == abi.BUILDER_PLOC: r.close()
    r.set_accel_builder(abi.BUILDER_LBVH)
    try:
        img2 = debug_image(r, s, h, abi.DEBUG_OUTPUT_HIT_KIND)
        assert np.array_equal(img2, ref) and r.stats().bvh_stack_need <= 64
    except MiptError as e:
        assert "traversal stack" in str(e)
    r.close()
    # garbage vertices: NaN and 1e30 among ordinary ones
    bad = pos[: 3 * 300].copy()
    bad += np.random.default_rng(5).normal(0, 0.3, bad.shape).astype(f32)
    bad[7] = np.nan; bad[100] = 1e30; bad[203, 1] = -np.inf
    s2 = scenes.single_triangle(32)
    s2.instances.clear(); s2.mesh_records.clear(); s2.buffers.clear(); s2.triangles = 0
    s2.add_mesh(meshgen.Mesh(bad, np.arange(900), normals=np.tile(np.array([[0, -1, 0]], f32), (900, 1))), None, 0)
    for b in (abi.BUILDER_PLOC_REINSERT, abi.BUILDER_PLOC, abi.BUILDER_LBVH):
        r = R(); r.set_accel_builder(b); h = s2.upload(r)
        try:
            out = debug_image(r, s2, h, abi.DEBUG_OUTPUT_HIT_KIND)
            assert out.shape == (32, 32, 4)
        except MiptError:
            pass                                                # refusing garbage is fine; hanging or faulting is not
        r.close()


def test_refit_of_the_figure_scene_matches_the_oracle(R, oracle_lib):
    """config-5 class: build at pose A, refit to pose B, render; the oracle (which always rebuilds) at pose B must agree."""
    import oracle.pyoracle as po
    s = scenes.skinned_figure(96, 54)
    r = R(); hg = s.upload(r)
    bg = scenes.SkinBinding(r, s, hg, 0, use_mfma=0)
    bg.pose(0.1); r.build_accel()
    bg.pose(0.8); r.build_accel()
    assert r.stats().accel_refits == 1 and r.stats().accel_builds == 1
    o = oracle_lib.Oracle(); ho = s.upload(o)
    bo = scenes.SkinBinding(o, s, ho, 0, use_mfma=0)
    bo.pose(0.8)
    eo = o.env_create_raw(*r.env_read(hg["env"]))
    og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
    for f in range(48):
        r.trace(s.settings, s.execute_params(f, env_handle=hg["env"]), og)
        o.trace(s.settings, s.execute_params(f, env_handle=eo), b)
    e = rel_l2(r.tonemap(og), po.tonemap(b))
    assert e <= 1e-3, e
    r.close(); o.close()


def test_instance_table_diffing_identical_moved_and_reshaped(R):
    s = scenes.test_scene(64, 32)
    r = R(); h = s.upload(r)
    r.build_accel()
    assert r.stats().accel_builds == 1
    ref = debug_image(r, s, h, abi.DEBUG_OUTPUT_VERTEX_NORMAL)
    # the same table again (what gs_frame sends every frame of a static scene): nothing to do
    r.set_instances(h["instances"]); r.build_accel()
    q = r.stats()
    assert (q.accel_builds, q.accel_refits) == (1, 0)
    assert np.array_equal(ref, debug_image(r, s, h, abi.DEBUG_OUTPUT_VERTEX_NORMAL))
    # one instance moved and mirrored: refit; the image must be that of a context that builds the moved table from scratch
    moved = [abi.PtInstanceDesc.from_buffer_copy(bytes(d)) for d in h["instances"]]
    T = camera.trs((0.3, 0.6, 1.1), scale=(1, -1, 1.3))
    moved[3].gpu.transform[:] = camera.cm(T); moved[3].gpu.normal_transform[:] = camera.cm(camera.inverse_transpose(T))
    r.set_instances(moved); r.build_accel()
    q = r.stats()
    assert (q.accel_builds, q.accel_refits) == (1, 1)
    imgs = [debug_image(r, s, h, d) for d in (abi.DEBUG_OUTPUT_HIT_KIND, abi.DEBUG_OUTPUT_VERTEX_NORMAL, abi.DEBUG_OUTPUT_COLOR)]
    r2 = R(); h2 = s.upload(r2)
    moved2 = [abi.PtInstanceDesc.from_buffer_copy(bytes(d)) for d in h2["instances"]]
    moved2[3].gpu.transform[:] = camera.cm(T); moved2[3].gpu.normal_transform[:] = camera.cm(camera.inverse_transpose(T))
    r2.set_instances(moved2)
    for img, d in zip(imgs, (abi.DEBUG_OUTPUT_HIT_KIND, abi.DEBUG_OUTPUT_VERTEX_NORMAL, abi.DEBUG_OUTPUT_COLOR)):
        other = debug_image(r2, s, h2, d)
        assert np.array_equal(img, other)
    assert np.abs(imgs[1] - ref).max() > 0.1                    # and it did move
    # a material swap alone touches no packet: neither build nor refit
    swapped = [abi.PtInstanceDesc.from_buffer_copy(bytes(d)) for d in moved]
    swapped[2].gpu.material_id = 3                              # the textured sphere becomes the clearcoat material
    r.set_instances(swapped); r.build_accel()
    q = r.stats()
    assert (q.accel_builds, q.accel_refits) == (1, 1)
    assert np.abs(debug_image(r, s, h, abi.DEBUG_OUTPUT_COLOR) - imgs[2]).max() > 0.01
    # an instance dropped: the set of triangles changed -> full build
    r.set_instances(swapped[:-1]); r.build_accel()
    q = r.stats()
    assert q.accel_builds == 2 and q.bvh_triangles < s.triangles
    # pt_buffer_update of a position stream -> refit
    pos_handle = swapped[4].gpu.position_descriptor
    n = swapped[4].num_of_vertices
    pos = r.buffer_read(pos_handle, np.float32, n * 3).reshape(-1, 3)
    r.buffer_update(pos_handle, (pos * f32(1.25)).astype(f32))
    before = r.stats().accel_refits
    bigger = debug_image(r, s, h, abi.DEBUG_OUTPUT_HIT_KIND)
    assert r.stats().accel_refits == before + 1 and r.stats().accel_builds == 2
    r.request_rebuild()
    assert np.array_equal(bigger, debug_image(r, s, h, abi.DEBUG_OUTPUT_HIT_KIND))
    r.close(); r2.close()


def test_resource_destroy_and_scene_swap(R, oracle_lib):
    """Gltf::Unload semantics: resources in use cannot go; after the tables are replaced they can, their handles are reused, and a
    second scene loaded into the same context renders exactly like in a fresh one."""
    from gltf_renderer_amd.renderer import MiptError
    r = R()
    s1 = scenes.test_scene(48, 16)
    h1 = s1.upload(r)
    img1 = debug_image(r, s1, h1, abi.DEBUG_OUTPUT_COLOR)
    with pytest.raises(MiptError):
        r.buffer_destroy(h1["instances"][0].gpu.position_descriptor)        # still in the instance table
    with pytest.raises(MiptError):
        r.texture_destroy(h1["textures"][0])                                 # still in the material table
    with pytest.raises(MiptError):
        r.set_materials([abi.PtMaterial.default()])                          # would leave dangling material ids
    r.set_instances([]); r.set_materials([abi.PtMaterial.default()])
    for b in h1["buffers"]:
        r.buffer_destroy(b)
    for t in h1["textures"]:
        r.texture_destroy(t)
    r.env_destroy(h1["env"])
    with pytest.raises(MiptError):
        r.buffer_destroy(h1["buffers"][0])                                    # twice
    with pytest.raises(MiptError):
        r.buffer_read(h1["buffers"][0], np.float32, 3)
    out = r.create_output(s1.width, s1.height)
    with pytest.raises(MiptError):
        r.trace(s1.settings, s1.execute_params(0, env_handle=h1["env"]), out)  # destroyed environment
    s2 = scenes.skinned_figure(48, 32)
    h2 = s2.upload(r)
    assert min(h2["buffers"]) < len(h1["buffers"])                           # handles are reused
    a = debug_image(r, s2, h2, abi.DEBUG_OUTPUT_VERTEX_NORMAL)
    r2 = R(); h3 = s2.upload(r2)
    assert np.array_equal(a, debug_image(r2, s2, h3, abi.DEBUG_OUTPUT_VERTEX_NORMAL))
    assert img1.shape[0] == 48
    r.close(); r2.close()


def test_skin_joint_ids_beyond_the_bone_array_contribute_nothing(R, oracle_lib):
    """A joint id >= bone_count reads zeros upstream (D3D12 robust buffer access): no contribution.  Both kernels and the oracle
    must agree and nothing may be read out of bounds."""
    s = scenes.skinned_figure(32, 18)
    sk = s.skins[0]
    jw = s.buffers[sk["joint_weight"]][0].copy().reshape(-1, 8)          # u16 x 8: four joints, four unorm16 weights
    assert jw.dtype == np.uint16
    jw[::3, 2] = 19 + (np.arange(len(jw[::3])) % 40000)         # third joint of every third vertex: beyond the 19 bones
    jw[5::7, 0] = 65535
    s.buffers[sk["joint_weight"]] = (jw, abi.FORMAT_JOINT_WEIGHT)
    outs = []
    for backend, mfma in ((R(), 0), (R(), 1), (oracle_lib.Oracle(), 0)):
        h = s.upload(backend)
        b = scenes.SkinBinding(backend, s, h, 0, mfma)
        b.pose(0.6)
        n = sk["mesh"].num_vertices
        outs.append((backend.buffer_read(b.out_position, np.float32, n * 3).reshape(-1, 3), backend.buffer_read(b.out_tangent_space, np.uint32, n)))
        backend.close()
    (pv, tv), (pm, tm), (po_, to_) = outs
    scale = max(1.0, float(np.abs(po_).max()))
    assert np.abs(pv - po_).max() < 2e-6 * scale and np.abs(pm - po_).max() < 2e-5 * scale
    assert np.isfinite(pv).all() and np.isfinite(pm).all()
    assert np.mean(tv == to_) > 0.97 and np.mean(tm == to_) > 0.95


# ---- the exchange ------------------------------------------------------------------------------------------------------------
def test_tile_pack_unpack_composes_shards_bit_exactly(R):
    import torch
    s = scenes.test_scene(200, 16); s.height = 72               # 12.5 x 4.5 tiles: ragged on both axes
    r = R(); h = s.upload(r)
    st = copy_settings(s.settings); st.flags &= ~abi.FLAG_ACCUMULATE
    full = r.create_output(s.width, s.height)
    r.trace(st, s.execute_params(1, env_handle=h["env"]), full)
    for world in (2, 3, 8):
        frame = torch.full_like(full, float("nan"))
        total = 0
        for rank in range(world):
            part = r.create_output(s.width, s.height)
            part.fill_(123.0)                                   # foreign pixels hold garbage: the pack must not read them
            r.trace(st, s.execute_params(1, env_handle=h["env"], tile_rank=rank, tile_rank_count=world), part)
            packed = r.tiles_pack(part, rank, world)
            assert packed.numel() * 4 == r.tiles_packed_bytes(s.width, s.height, rank, world)
            total += packed.shape[0]
            r.tiles_unpack(packed, frame, rank, world)
        torch.cuda.synchronize()
        assert total == 13 * 5 * 256
        assert torch.equal(frame, full), world
    r.close()


@pytest.mark.parametrize("mode", [abi.EXCHANGE_GATHER, abi.EXCHANGE_REDUCE])
def test_exchange_frame_world_of_one_runs_rccl(R, mode):
    """One GPU per lease: the N > 1 path cannot run here, but a communicator of one can be created from a real ncclUniqueId and
    the frame call (pack -> grouped send/recv to self -> unpack, or ncclReduce of the masked copy) must return the image."""
    import torch
    s = scenes.test_scene(72, 16)
    r = R(); h = s.upload(r)
    uid = r.exchange_unique_id()
    assert len(uid) == abi.EXCHANGE_ID_BYTES and any(uid)
    r.exchange_create(0, 1, uid)
    acc = r.create_output(s.width, s.height)
    st = copy_settings(s.settings); st.reset = 1
    for f in range(3):                                          # accumulate, exchange after every frame
        r.trace(st, s.execute_params(f, env_handle=h["env"]), acc); st.reset = 0
        frame = torch.full_like(acc, -1.0)
        r.exchange_frame(acc, frame, mode=mode)
        torch.cuda.synchronize()
        assert torch.equal(frame, acc), f
    r.exchange_frame(acc, None, mode=mode)                      # in place
    torch.cuda.synchronize()
    assert r.stats().accumulated_frames == 3
    r.exchange_destroy()
    r.exchange_create(0, 1, None)                               # no communicator at all
    frame = torch.zeros_like(acc)
    r.exchange_frame(acc, frame, mode=mode)
    torch.cuda.synchronize()
    assert torch.equal(frame, acc)
    r.close()


def test_two_contexts_on_one_device_are_independent(R):
    """Every entry point makes its context's device current and works on its own tables (two contexts in one process)."""
    s1, s2 = scenes.test_scene(48, 16), scenes.skinned_figure(48, 32)
    ra, rb = R(), R()
    ha, hb = s1.upload(ra), s2.upload(rb)
    a1 = debug_image(ra, s1, ha, abi.DEBUG_OUTPUT_COLOR); b1 = debug_image(rb, s2, hb, abi.DEBUG_OUTPUT_COLOR)
    a2 = debug_image(ra, s1, ha, abi.DEBUG_OUTPUT_COLOR); b2 = debug_image(rb, s2, hb, abi.DEBUG_OUTPUT_COLOR)
    assert np.array_equal(a1, a2) and np.array_equal(b1, b2)
    ra.close()
    assert np.array_equal(b1, debug_image(rb, s2, hb, abi.DEBUG_OUTPUT_COLOR))
    rb.close()


# ---- configs round 1 left untested on the GPU --------------------------------------------------------------------------------
def test_config2_helmet_class_matches_the_oracle(R, oracle_lib):
    """BASELINE config 2 (DamagedHelmet class: one ~20 k-triangle displaced icosphere at this size, five textures, env-map IBL with a
    1e4 sun, 4 bounces, no punctual lights) at a reduced frame against the oracle."""
    import oracle.pyoracle as po
    s = scenes.helmet_class(240, 136, subdiv=5, tex=512)
    p = Pair(R, oracle_lib, s)
    for dbg in (abi.DEBUG_OUTPUT_HIT_KIND, abi.DEBUG_OUTPUT_SHADING_NORMAL, abi.DEBUG_OUTPUT_COLOR, abi.DEBUG_OUTPUT_ROUGHNESS):
        st = copy_settings(s.settings); st.debug_output = dbg; st.flags &= ~abi.FLAG_ACCUMULATE; st.use_frame_as_seed = 0; st.seed = 2
        og, b = p.render(settings=st)
        err = np.abs(p.r.readback(og)[..., :3] - b[..., :3]).max(axis=2)
        assert (err > 0).sum() == 0, (abi.DEBUG_OUTPUT_NAMES[dbg], int((err > 0).sum()))
    og, b = p.render(frames=64)
    e = rel_l2(p.r.tonemap(og), po.tonemap(b))
    sg, so = p.r.stats(), p.o.counters()
    assert int(sg.rays) == so["rays"]
    assert sg.rays_shadow > 0 and sg.rays_bounce > 0
    assert e <= 1e-3, e
    p.close()


def test_config2_helmet_class_fullsize_properties(R):
    """config 2 at its own size (1920x1080, 81,920 triangles, 2048^2 textures): determinism, 8 shards == 1 frame bit for bit, finite,
    ray-count bound 1 primary + 4 bounce + 4 env-shadow (no punctual lights)."""
    s = scenes.helmet_class()
    r = R(); h = s.upload(r)
    st = copy_settings(s.settings); st.flags &= ~abi.FLAG_ACCUMULATE
    full = r.create_output(s.width, s.height)
    r.reset_stats(); r.trace(st, s.execute_params(5, env_handle=h["env"]), full)
    q = r.stats()
    full_h = r.readback(full)
    again = r.create_output(s.width, s.height)
    r.trace(st, s.execute_params(5, env_handle=h["env"]), again)
    assert np.array_equal(full_h, r.readback(again))
    acc = np.zeros_like(full_h); r.reset_stats()
    for rank in range(8):
        part = r.create_output(s.width, s.height)
        r.trace(st, s.execute_params(5, env_handle=h["env"], tile_rank=rank, tile_rank_count=8), part)
        acc += r.readback(part)
    assert np.array_equal(acc, full_h) and r.stats().rays == q.rays
    assert np.isfinite(full_h).all() and full_h[..., :3].mean() > 0.01
    n = s.width * s.height
    assert q.rays_primary == n and q.rays <= 9 * n and q.rays_shadow <= 4 * n + q.closest_hits * 0 and q.bvh_triangles == 81920
    r.close()


def test_config3_sponza_class_at_8_bounces_matches_the_oracle(R, oracle_lib):
    """config 3 at ITS OWN bounce settings (max 8, min 2, RR 0.1-0.9) on a reduced frame: per-sample agreement and the tone-mapped
    image metric at 64 accumulated samples."""
    import oracle.pyoracle as po
    s = scenes.sponza_class(width=320, height=180, tex=64)
    assert s.settings.max_bounces == 8
    p = Pair(R, oracle_lib, s)
    st = copy_settings(s.settings); st.flags &= ~abi.FLAG_ACCUMULATE; st.use_frame_as_seed = 0; st.seed = 9
    og, b = p.render(settings=st)
    a = p.r.readback(og)[..., :3].astype(np.float64); bb = b[..., :3].astype(np.float64)
    rel = np.abs(a - bb).max(axis=2) / np.maximum(np.abs(bb).max(axis=2), 1e-6)
    sg, so = p.r.stats(), p.o.counters()
    assert int(sg.rays) == so["rays"]
    frac = float((rel > 1e-2).mean())
    assert np.median(rel) < 1e-6 and frac == 0.0, (float(np.median(rel)), frac)      # no pixel-sample beyond 1e-2 (0.0004 with the fp32 library sin / cos)
    # The image metric.  This scene amplifies rounding: tiled textures put texture coordinates at tens of units, glossy lobes turn a
    # perturbed shading normal into a different path, and a path that then reaches the 1e4-radiance sun on one side only saturates its
    # pixel whatever the sample count.  History of this figure (rel L2 at 64 spp): 2.3e-2 with an FMA-contracting build (1.1 % of the
    # pixel-samples on another path than the oracle's); 4.9e-3 built like the oracle, without contraction (0.04 %); 8.4e-4 with sin / cos
    # evaluated in double and rounded once (none beyond 1e-2); and 3.4e-7 once the oracle defines sin / cos the same way instead of taking
    # the host libm's float routines (hlsl.h o_sin / o_cos): NO pixel-sample of 3.7 M beyond 1e-3 (tools/diag_config3_events.py), every
    # debug output of the first vertex bit-identical except the BSDF value, whose pow is v_exp_f32 / v_log_f32 here and libm there
    # (tools/diag_bit_audit.py).  Asserted with a margin of 30.  
    og = p.r.create_output(s.width, s.height)
    b = np.zeros((s.height, s.width, 4), np.float32)
    st = copy_settings(s.settings); st.reset = 1
    errs, bulk = {}, {}

    def bulk_l2(ta, tb):                                      # rel L2 without the 1 % of the pixels that differ most
        ta = ta.astype(np.float64); tb = tb.astype(np.float64)
        ok = np.isfinite(ta).all(axis=2) & np.isfinite(tb).all(axis=2)
        sq = np.where(ok, ((np.nan_to_num(ta) - np.nan_to_num(tb)) ** 2).sum(axis=2), 0.0).ravel()
        return float(np.sqrt(np.sort(sq)[: -max(sq.size // 100, 1)].sum() / (tb[ok] ** 2).sum()))

    for f in range(64):
        p.r.trace(st, s.execute_params(frame=f, env_handle=p.hg["env"]), og)
        p.o.trace(st, s.execute_params(frame=f, env_handle=p.ho["env"]), b)
        st.reset = 0
        if f + 1 in (16, 64):
            ta, tb = p.r.tonemap(og), po.tonemap(b)
            errs[f + 1] = rel_l2(ta, tb, nan_mismatch=1e-2)   # (this scene has near-black pixels: AgX's pow of +-1e-7)
            bulk[f + 1] = bulk_l2(ta, tb)
    ok = np.isfinite(ta).all(axis=2) & np.isfinite(tb).all(axis=2)
    d = np.abs(ta - tb).max(axis=2)[ok]
    bias = float((ta[ok].astype(np.float64) - tb[ok]).sum() / tb[ok].astype(np.float64).sum())
    print("config 3, 8 bounces: tone-mapped rel L2 %.3e at 16 spp, %.3e at 64 spp; without the 1 %% worst pixels %.3e -> %.3e; relative bias %.2e; median |diff| %.2e; "
          "pixels with |diff| > 0.05: %.4f; 1-spp pixel-samples beyond 1e-2: %.4f" % (errs[16], errs[64], bulk[16], bulk[64], bias, float(np.median(d)), float((d > 0.05).mean()), frac))
    # THE CONTRACT is the north_star's 1e-3 on the whole image (measured 3.4e-7 / 3.3e-7: printed above).  Agreement far below it holds only
    # because the oracle and the kernels define sin / cos, division and contraction alike -- a real DXR driver is free not
    # to; the 1e-4 lines are a regression guard on that co-definition, not a claim about the reference.
    assert errs[64] <= 1e-3 and errs[16] <= 1e-3, errs
    assert errs[64] <= 1e-4 and bulk[64] <= 1e-4 and bulk[16] <= 1e-4, (errs, bulk)
    assert abs(bias) < 1e-4 and (d > 0.05).sum() == 0
    p.close()


def _fullsize_properties(R, s, frame, shards, rays_per_pixel_bound):
    r = R(); h = s.upload(r)
    st = copy_settings(s.settings); st.flags &= ~abi.FLAG_ACCUMULATE
    full = r.create_output(s.width, s.height)
    r.reset_stats(); r.trace(st, s.execute_params(frame, env_handle=h["env"]), full)
    q = r.stats()
    full_h = r.readback(full)
    again = r.create_output(s.width, s.height)
    r.trace(st, s.execute_params(frame, env_handle=h["env"]), again)
    assert np.array_equal(full_h, r.readback(again))                         # determinism
    acc = np.zeros_like(full_h); r.reset_stats()
    for rank in range(shards):
        part = r.create_output(s.width, s.height)
        r.trace(st, s.execute_params(frame, env_handle=h["env"], tile_rank=rank, tile_rank_count=shards), part)
        acc += r.readback(part)
    assert np.array_equal(acc, full_h) and r.stats().rays == q.rays          # shards compose, same rays
    assert np.isfinite(full_h).all() and full_h[..., :3].mean() > 0.005
    n = s.width * s.height
    assert q.rays_primary == n and n <= q.rays <= rays_per_pixel_bound * n
    # linearity in the environment with punctual lights off (power-of-two scaling commutes with every fp32 operation)
    st2 = copy_settings(st); st2.flags &= ~abi.FLAG_POINT_LIGHTS; st2.use_frame_as_seed = 0; st2.seed = 11
    a = r.create_output(s.width, s.height); b = r.create_output(s.width, s.height)
    st2.environment_intensity = 1.0; r.trace(st2, s.execute_params(0, env_handle=h["env"]), a)
    st2.environment_intensity = 2.0; r.trace(st2, s.execute_params(0, env_handle=h["env"]), b)
    assert np.array_equal(r.readback(a)[..., :3] * 2.0, r.readback(b)[..., :3])
    r.close()
    return q


def test_config4_material_grid_fullsize_properties(R):
    """config 4 at its own size: 1024x1024, 32-segment spheres (~76 k triangles), 16 bounces with the clamp lifted."""
    s = scenes.material_grid(1024, seg=32)
    assert s.settings.max_bounces == 16 and s.width == 1024
    q = _fullsize_properties(R, s, 3, 8, 1 + 16 + 16 + 17)
    assert q.rays_bounce > 0.5 * s.width * s.height


def test_config5_skinned_figure_4k_properties(R):
    """config 5 at its own size: 3840x2160, 8 bounces, the figure skinned to a walk-cycle pose (skin -> refit -> trace)."""
    s = scenes.skinned_figure()
    assert (s.width, s.height) == (3840, 2160)
    r = R(); h = s.upload(r)
    bind = scenes.SkinBinding(r, s, h, 0, use_mfma=1)
    bind.pose(0.0); r.build_accel(); bind.pose(0.55)
    st = copy_settings(s.settings); st.flags &= ~abi.FLAG_ACCUMULATE
    full = r.create_output(s.width, s.height)
    r.reset_stats(); r.trace(st, s.execute_params(7, env_handle=h["env"]), full)
    q = r.stats()
    assert q.accel_refits == 1 and q.accel_builds == 1
    full_h = r.readback(full)
    acc = np.zeros_like(full_h); r.reset_stats()
    for rank in range(8):
        part = r.create_output(s.width, s.height)
        r.trace(st, s.execute_params(7, env_handle=h["env"], tile_rank=rank, tile_rank_count=8), part)
        acc += r.readback(part)
        del part
    assert np.array_equal(acc, full_h) and r.stats().rays == q.rays
    assert np.isfinite(full_h).all()
    n = s.width * s.height
    assert q.rays_primary == n and n <= q.rays <= 26 * n
    r.close()


def test_megakernel_mode_matches_the_oracle(R, oracle_lib):
    """The megakernel arrangement (pt_kernel.hip) against the oracle, not only against the wavefront mode."""
    import oracle.pyoracle as po
    s = scenes.test_scene(96, 64)
    p = Pair(R, oracle_lib, s)
    p.r.set_kernel_mode(abi.MODE_MEGAKERNEL)
    for dbg in (abi.DEBUG_OUTPUT_HIT_KIND, abi.DEBUG_OUTPUT_SHADING_NORMAL, abi.DEBUG_OUTPUT_BOUNCE_WEIGHT):
        st = copy_settings(s.settings); st.debug_output = dbg; st.flags &= ~abi.FLAG_ACCUMULATE; st.use_frame_as_seed = 0; st.seed = 5
        og, b = p.render(settings=st)
        err = np.abs(p.r.readback(og)[..., :3] - b[..., :3]).max(axis=2)
        tol = 2e-4 + 2e-3 * np.abs(b[..., :3]).max(axis=2)
        assert (err > tol).mean() <= 0.003, abi.DEBUG_OUTPUT_NAMES[dbg]
    og, b = p.render(frames=32)
    e = rel_l2(p.r.tonemap(og), po.tonemap(b))
    sg, so = p.r.stats(), p.o.counters()
    assert int(sg.rays) == so["rays"]
    assert e <= 1e-3, e
    p.close()


@pytest.mark.parametrize("mode", [abi.MODE_WAVEFRONT, abi.MODE_MEGAKERNEL])
def test_show_nan_and_show_inf_paint_red_on_both_sides(R, oracle_lib, mode):
    """RayGeneration's scrub (PathTracer.lib.hlsl:760-766): a NaN sample becomes 0, or red with FLAG_SHOW_NAN; then an Inf sample
    becomes 0, or red with FLAG_SHOW_INF.  An infinite constant environment colour makes every escaping path Inf; a NaN
    environment intensity makes it NaN."""
    s = scenes.test_scene(64, 32, with_env=False)
    p = Pair(R, oracle_lib, s)
    p.r.set_kernel_mode(mode)
    red = lambda im: (im[..., 0] == 1) & (im[..., 1] == 0) & (im[..., 2] == 0)
    for case, color, intensity, flag in (("inf", (float("inf"), 1.0, 0.5), 1.0, abi.FLAG_SHOW_INF), ("nan", (1.0, 1.0, 1.0), float("nan"), abi.FLAG_SHOW_NAN)):
        base = copy_settings(s.settings); base.flags &= ~abi.FLAG_ACCUMULATE; base.use_frame_as_seed = 0; base.seed = 21
        base.environment_color[:] = color; base.environment_intensity = intensity
        results = {}
        for name, fl in (("scrub", 0), ("nan", abi.FLAG_SHOW_NAN), ("inf", abi.FLAG_SHOW_INF), ("both", abi.FLAG_SHOW_NAN | abi.FLAG_SHOW_INF)):
            st = copy_settings(base); st.flags |= fl
            og, b = p.render(settings=st)
            a = p.r.readback(og)
            assert np.isfinite(a).all() and np.isfinite(b).all(), (case, name)
            mism = float((np.abs(a[..., :3] - b[..., :3]).max(axis=2) > 1e-3 * (1 + np.abs(b[..., :3]).max(axis=2))).mean())
            assert mism < 0.01, (case, name, mism)
            assert np.array_equal(red(a), red(b)) or (red(a) != red(b)).mean() < 0.005, (case, name)
            results[name] = a
        shown = "inf" if flag == abi.FLAG_SHOW_INF else "nan"
        assert red(results[shown]).mean() > 0.3, case                 # the sky and everything an escaping path reaches
        assert red(results["both"]).sum() >= red(results[shown]).sum()
        assert red(results["scrub"]).sum() == 0, case                 # scrubbed to 0 instead
    p.close()


def test_orthographic_camera_matches_the_oracle(R, oracle_lib):
    """Camera::Orthographic (Camera.h:31-40, GetViewToClip :91): parallel rays from the reversed-Z ortho matrix."""
    import oracle.pyoracle as po
    s = scenes.test_scene(96, 64)
    s.ortho = (0.3, 0.45)                                         # half extents 1 / mag (sic)
    p = Pair(R, oracle_lib, s)
    st = copy_settings(s.settings); st.debug_output = abi.DEBUG_OUTPUT_HIT_KIND; st.flags &= ~abi.FLAG_ACCUMULATE; st.use_frame_as_seed = 0; st.seed = 1
    og, b = p.render(settings=st)
    a = p.r.readback(og)
    assert (np.abs(a - b).max(axis=2) > 0).sum() == 0
    hit = ((a[..., 0] == 1) & (a[..., 1] == 0)) | ((a[..., 0] == 0) & (a[..., 1] == 1))
    assert 0.2 < hit.mean() < 1.0                                 # the scene is in view, and so is the sky
    persp = scenes.test_scene(96, 64)
    pp = R(); hp = persp.upload(pp)
    assert np.abs(debug_image(pp, persp, hp, abi.DEBUG_OUTPUT_HIT_KIND, seed=1) - a).max() > 0.5     # not the perspective image
    pp.close()
    og, b = p.render(frames=32)
    e = rel_l2(p.r.tonemap(og), po.tonemap(b))
    assert e <= 1e-3, e
    p.close()


def interleave_scene(size=160):
    """A wall of quads whose materials exercise the interleaved albedo / normal / metal-rough footprint (pt_types.h RM_TRIO) and
    every reason to stay on the general path: all three textures alike; two of the three; a tiling WRAP transform (footprints that
    cross the texture's last column); MIRROR / CLAMP and point samplers; a 1-texel-wide texture; and normal maps of another size,
    another sampler, another UV transform, the second UV set -- side by side, so waves mix both paths."""
    import math
    rng = np.random.default_rng(11)
    s = scenes.SceneData("interleave_cases")
    TS = abi.PtTextureSample
    def tex_set(n, w=None):
        w = w or n
        h = scenes.value_noise(rng, n, 4, 2)[:, :w] if w != n else scenes.value_noise(rng, n, 4, 2)
        g = scenes.value_noise(rng, n, 3, 3)[:, :w] if w != n else scenes.value_noise(rng, n, 3, 3)
        b = s.add_texture(scenes.rgba(0.2 + 0.8 * h, 0.3 + 0.6 * g, 0.9 - 0.7 * h, 0.5 + 0.5 * g), True)
        mr = s.add_texture(scenes.rgba(np.ones_like(h), 0.2 + 0.7 * g, 0.8 * h), False)
        nm_full = scenes.normal_map_from_height(scenes.value_noise(rng, n, 4, 2), 3.0)
        nm = s.add_texture(nm_full[:, :w] if w != n else nm_full, False)
        return b, mr, nm
    b64, mr64, nm64 = tex_set(64)
    b32, mr32, nm32 = tex_set(32)
    b1, mr1, nm1 = tex_set(16, 1)                       # 1 texel wide, 16 high
    em64 = s.add_texture(scenes.rgba(scenes.value_noise(rng, 64, 3, 2) ** 3, 0.5 * scenes.value_noise(rng, 64, 3, 2) ** 2, 0.2 * scenes.value_noise(rng, 64, 2, 2)), True)
    em32 = s.add_texture(scenes.rgba(scenes.value_noise(rng, 32, 3, 2) ** 2, scenes.value_noise(rng, 32, 3, 2) ** 3, 0.1 * scenes.value_noise(rng, 32, 2, 2)), False)
    smp_mc = s.add_sampler(abi.ADDRESS_MIRROR, abi.ADDRESS_CLAMP, abi.FILTER_LINEAR, abi.FILTER_LINEAR)
    smp_pt = s.add_sampler(abi.ADDRESS_WRAP, abi.ADDRESS_WRAP, abi.FILTER_POINT, abi.FILTER_POINT)
    tile = dict(rotation=0.4, offset=(0.13, 0.27), scale=(3.0, 2.0))
    M = scenes.material
    mats = [
        M(albedo=TS(b64), metallic_roughness=TS(mr64), normal=TS(nm64)),                                             # interleaved: all three
        M(albedo=TS(b64), normal=TS(nm64), roughness_factor=0.4),                                                    # interleaved: albedo + normal
        M(albedo=TS(b32), metallic_roughness=TS(mr32), metalness_factor=1.0),                                        # interleaved: albedo + metal-rough
        M(albedo=TS(b64, 0, 0, **tile), metallic_roughness=TS(mr64, 0, 0, **tile), normal=TS(nm64, 0, 0, **tile)),   # interleaved, tiled + rotated
        M(albedo=TS(b32, smp_mc), metallic_roughness=TS(mr32, smp_mc), normal=TS(nm32, smp_mc, 0, 0.0, (0, 0), (2.5, 2.5))),   # general: transform differs
        M(albedo=TS(b32, smp_mc, 0, 0.0, (0.3, 0.1), (2.5, 2.5)), metallic_roughness=TS(mr32, smp_mc, 0, 0.0, (0.3, 0.1), (2.5, 2.5)),
          normal=TS(nm32, smp_mc, 0, 0.0, (0.3, 0.1), (2.5, 2.5))),                                                  # interleaved, MIRROR / CLAMP
        M(albedo=TS(b64, smp_pt, 0, 0.0, (0, 0), (2, 2)), metallic_roughness=TS(mr64, smp_pt, 0, 0.0, (0, 0), (2, 2)), normal=TS(nm64, smp_pt, 0, 0.0, (0, 0), (2, 2))),   # interleaved, point
        M(albedo=TS(b1, 0, 0, 0.0, (0, 0), (4, 4)), metallic_roughness=TS(mr1, 0, 0, 0.0, (0, 0), (4, 4)), normal=TS(nm1, 0, 0, 0.0, (0, 0), (4, 4))),   # interleaved, width 1
        M(albedo=TS(b64), metallic_roughness=TS(mr64), normal=TS(nm32)),                                             # general: size differs
        M(albedo=TS(b64), metallic_roughness=TS(mr64), normal=TS(nm64, smp_mc)),                                     # general: sampler differs
        M(albedo=TS(b64), metallic_roughness=TS(mr64), normal=TS(nm64, 0, 1)),                                       # general: UV set differs
        M(normal=TS(nm64), metallic_roughness=TS(mr64)),                                                             # general: no albedo texture
        M(albedo=TS(b64)),                                                                                           # general: albedo only
        M(albedo=TS(b64), metallic_roughness=TS(mr64), normal=TS(nm64), emissive=TS(em64), emissive_factor=(1.5, 1.2, 2.0)),   # interleaved, emissive rides along (sRGB)
        M(albedo=TS(b32, 0, 0, **tile), metallic_roughness=TS(mr32, 0, 0, **tile), emissive=TS(em32, 0, 0, **tile), emissive_factor=(2.0, 2.0, 2.0)),   # same, linear, tiled
        M(albedo=TS(b64), normal=TS(nm64), emissive=TS(em32), emissive_factor=(1.0, 2.0, 1.0)),                       # interleaved; emissive of another size fetched on its own
        M(albedo=TS(b64), emissive=TS(em64), emissive_factor=(1.0, 1.0, 1.0)),                                       # general: an emissive texture alone does not make a copy
    ]
    ids = [s.add_material(m) for m in mats]
    cols = 5
    for k, mid in enumerate(ids):
        cx, cz = (k % cols - (cols - 1) / 2) * 1.05, (k // cols - 1.5) * 1.05
        g = meshgen.grid(6, 6, (cx - 0.5, 0.0, cz - 0.5), (1, 0, 0), (0, 0, 1), (1, 1))
        g.uv1 = (g.uv0 * f32(1.7) + f32(0.2)).astype(f32)
        s.add_mesh(g, None, mid)
    s.add_light(abi.LIGHT_POINT, position=(0.5, -2.0, 0.3), color=(1, 0.9, 0.8), intensity=30.0)
    s.add_light(abi.LIGHT_DIRECTIONAL, direction=(0.2, 1.0, -0.3), color=(1, 1, 1), intensity=2.0)
    s.world_to_view = camera.orbit_world_to_view((0, 0, 0), 3.6, 0.0, 0.0)
    s.width, s.height = size, size * 3 // 4
    s.settings.max_bounces = 3
    s.settings.flags &= ~abi.FLAG_ENVIRONMENT_MAP
    return s, 10, 2


@pytest.mark.parametrize("mode", ["wavefront", "megakernel"])
def test_interleaved_texel_footprint_is_bit_identical_to_the_general_path(R, oracle_lib, mode, monkeypatch):
    """pt_scene_set_materials interleaves the albedo / normal / metal-rough texels of the materials whose three footprints coincide;
    the shade stage then reads two 32-B pieces a row pair instead of six 8-B pieces.  Same texels, same weights: the image must not
    change by one bit, whatever mix of materials a wave holds, and must match the oracle (which knows nothing of it)."""
    s, expect_interleaved, expect_emissive = interleave_scene()
    frames = 6
    def render(env):
        if env is None: monkeypatch.delenv("MIPT_TEXTURE_INTERLEAVE", raising=False)
        else: monkeypatch.setenv("MIPT_TEXTURE_INTERLEAVE", env)
        r = R()
        if mode == "megakernel": r.set_kernel_mode(1)
        h = s.upload(r)
        n = (r.L.pt_debug_interleaved_materials(r.h), r.L.pt_debug_interleaved_emissive(r.h))
        out = r.create_output(s.width, s.height)
        for f in range(frames):
            r.trace(s.settings, s.execute_params(frame=f), out)
        img = r.readback(out)
        r.close()
        return img, n
    a, na = render(None)
    b, nb = render("0")
    assert na == (expect_interleaved, expect_emissive) and nb == (0, 0), (na, nb)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "interleaved footprint changed %d pixels" % int((a != b).any(axis=2).sum())
    o = oracle_lib.Oracle()
    s.upload(o)
    ref = np.zeros((s.height, s.width, 4), np.float32)
    for f in range(frames):
        o.trace(s.settings, s.execute_params(frame=f), ref)
    err = rel_l2(oracle_lib.tonemap(a), oracle_lib.tonemap(ref))
    assert err <= 1e-3, err


def test_interleaved_copies_follow_the_material_table(R):
    """A new material table releases the copies it no longer names and reuses the ones it does; textures stay destroyable."""
    s, n, _ = interleave_scene(64)
    r = R()
    h = s.upload(r)
    assert r.L.pt_debug_interleaved_materials(r.h) == n
    out = r.create_output(s.width, s.height)
    r.trace(s.settings, s.execute_params(frame=0), out)
    first = r.readback(out).copy()
    r.set_materials([abi.PtMaterial.default() for _ in s.materials])          # no texture named any more: every copy goes
    assert r.L.pt_debug_interleaved_materials(r.h) == 0
    r.trace(s.settings, s.execute_params(frame=0), out)
    r.close()
    r = R()
    s.upload(r)
    out = r.create_output(s.width, s.height)
    r.trace(s.settings, s.execute_params(frame=0), out)
    assert np.array_equal(first.view(np.uint32), r.readback(out).view(np.uint32))
    r.close()


@pytest.mark.parametrize("which", ["all_features", "sponza_class", "material_grid", "helmet_class"])
def test_first_vertex_quantities_are_bit_identical_to_the_oracle(R, oracle_lib, which):
    """Every debug output of the reference (Pathtracer.h:19-49: the per-pixel deterministic quantities of the first path vertex) rendered on
    both sides and compared BIT FOR BIT.  Built without contraction, with sin / cos correctly rounded and -- since round 3 -- pow, exp and
    atan2 as defined functions on both sides, the HIP path reproduces all 27 of them exactly: hit kind, vertex attributes, texture coordinates,
    albedo, the shading frame, every material scalar, the sampled bounce direction, the BSDF value, its pdf and the weight.  (The one
    tolerated difference is the sign of a zero pdf: 0 * x with x of either sign.)"""
    s = {"all_features": lambda: scenes.test_scene(160, 64), "sponza_class": lambda: scenes.sponza_class(width=320, height=180, tex=64),
         "material_grid": lambda: scenes.material_grid(256, seg=16), "helmet_class": lambda: scenes.helmet_class(width=320, height=180, subdiv=4, tex=256)}[which]()
    r = R(); hg = s.upload(r)
    if s.bounce_limit != 5: r.set_bounce_limit(s.bounce_limit)
    o = oracle_lib.Oracle(); ho = s.upload(o, env_raw=r.env_read(hg["env"]) if hg["env"] is not None else None)
    og = r.create_output(s.width, s.height); b = np.zeros((s.height, s.width, 4), np.float32)
    signed_zero = 0
    for dbg in range(1, 28):
        st = copy_settings(s.settings); st.debug_output = dbg; st.flags &= ~abi.FLAG_ACCUMULATE; st.use_frame_as_seed = 0; st.seed = 5
        r.trace(st, s.execute_params(frame=0, env_handle=hg["env"]), og)
        o.trace(st, s.execute_params(frame=0, env_handle=ho["env"]), b)
        a = r.readback(og)[..., :3]; bb = b[..., :3]
        both_nan = np.isnan(a) & np.isnan(bb)
        bits_differ = (a.view(np.uint32) != bb.view(np.uint32)) & ~both_nan
        zeros = bits_differ & (a == 0) & (bb == 0)                      # +0 against -0
        signed_zero += int(zeros.sum())
        assert not (bits_differ & ~zeros).any(), (abi.DEBUG_OUTPUT_NAMES[dbg], int((bits_differ & ~zeros).sum()))
    print("%s: all 27 debug outputs bit-identical (%d zeros of opposite sign)" % (which, signed_zero))
    r.close(); o.close()
