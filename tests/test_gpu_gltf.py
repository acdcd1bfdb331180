"""End to end on the GPU: file -> C++ loader (include/mipt_scene.h) -> path tracer must render what the directly uploaded
procedural scene renders.  The two paths share no host code above the C-ABI: one builds streams with numpy (scenes.py), the other
parses the exported .glb, converts accessors, encodes tangent spaces, decodes PNGs, walks the node tree and gathers per frame."""
import numpy as np
import pytest

from gltf_renderer_amd import scenes
from tests.scene_export import scene_to_builder, skinned_figure_to_builder

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def render_direct(R, s, frames, prepare=None):
    r = R()
    h = s.upload(r)
    if prepare:
        prepare(r, h)
    out = r.create_output(s.width, s.height)
    for f in range(frames):
        r.trace(s.settings, s.execute_params(f, env_handle=h["env"]), out)
    img = r.tonemap(out)
    st = r.stats()
    r.close()
    return img, st


def render_loaded(R, s, path, frames, animate_time=None):
    from gltf_renderer_amd.gltf import GltfScene
    r = R()
    sc = GltfScene(path)
    sc.upload(r)
    if animate_time is not None:
        sc.animate(0, animate_time)
    sc.calculate_global_transforms(0)
    lights = sc.frame(r, 0)
    assert lights == len(s.lights)
    env = r.env_create(s.env_image) if s.env_image is not None else None
    r.set_bounce_limit(s.bounce_limit)
    out = r.create_output(s.width, s.height)
    for f in range(frames):
        r.trace(s.settings, s.execute_params(f, env_handle=env), out)
    img = r.tonemap(out)
    st = r.stats()
    r.close(); sc.close()
    return img, st


@pytest.fixture(scope="module")
def R():
    from gltf_renderer_amd.renderer import Renderer
    return Renderer


def test_loader_fed_scene_renders_like_the_direct_upload(R, tmp_path):
    s = scenes.test_scene(128, 64)
    path = scene_to_builder(s).write_glb(str(tmp_path / "test_scene.glb"))
    a, sa = render_direct(R, s, 16)
    b, sb = render_loaded(R, s, path, 16)
    assert sa.bvh_triangles == sb.bvh_triangles
    # same streams up to: node matrices that went through decompose -> T*R*S (1e-7), tangent-space packing on rounding edges
    print("loader-fed scene against the direct upload: rel L2 %.3e" % rel_l2(b, a))
    assert rel_l2(b, a) <= 1e-4, rel_l2(b, a)                      # measured 1.8e-6


def test_animated_skinned_gltf_drives_the_dynamic_path(R, tmp_path):
    """config-5 class: walk cycle as glTF animation channels -> gs_animate -> bones -> pt_skin_run -> BVH rebuild -> trace,
    against the analytic pose of the generator fed through the same skinning kernel."""
    s = scenes.skinned_figure(160, 90)
    path = skinned_figure_to_builder(s).write_glb(str(tmp_path / "figure.glb"))
    t = 0.8
    def prepare(r, h):
        scenes.SkinBinding(r, s, h, 0, 1).pose(t)
    a, _ = render_direct(R, s, 16, prepare)
    b, _ = render_loaded(R, s, path, 16, animate_time=t)
    rest, _ = render_direct(R, s, 16)
    print("animated glTF against the analytic pose: rel L2 %.3e (bind pose: %.3e)" % (rel_l2(b, a), rel_l2(rest, a)))
    assert rel_l2(b, a) <= 1e-4, rel_l2(b, a)                      # measured 3.4e-7
    assert rel_l2(rest, a) > 5 * rel_l2(b, a)          # the pose matters: the bind pose is measurably a different image


def test_unload_then_load_another_file_in_the_same_context(R, tmp_path):
    """LoadGltf's flow (Main.cpp:43-54): Gltf::Unload, then the next file, in ONE context: the second scene must render exactly as in a
    fresh context, the first scene's handles are gone and reused, and unloading twice is harmless."""
    from gltf_renderer_amd.gltf import GltfScene
    from gltf_renderer_amd import abi
    s1, s2 = scenes.test_scene(96, 32), scenes.skinned_figure(96, 54)
    p1 = scene_to_builder(s1).write_glb(str(tmp_path / "a.glb"))
    p2 = skinned_figure_to_builder(s2).write_glb(str(tmp_path / "b.glb"))

    def frame_of(r, sc, s, frames=4):
        sc.calculate_global_transforms(0)
        lights = sc.frame(r, 0)
        assert lights == len(s.lights)
        r.set_bounce_limit(s.bounce_limit)
        out = r.create_output(s.width, s.height)
        st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.reset = 1
        for f in range(frames):
            r.trace(st, s.execute_params(f), out); st.reset = 0          # no environment map: the constant colour of the settings
        return r.readback(out)

    r = R()
    a = GltfScene(p1); a.upload(r)
    img_a = frame_of(r, a, s1)
    tris_a = r.stats().bvh_triangles
    a.unload(r); a.unload(r)
    b = GltfScene(p2); b.upload(r)
    img_b = frame_of(r, b, s2)
    assert r.stats().bvh_triangles != tris_a
    fresh = R()
    b2 = GltfScene(p2); b2.upload(fresh)
    assert np.array_equal(img_b, frame_of(fresh, b2, s2))
    # and back again: the first file once more, into the handles the second one leaves behind
    b.unload(r)
    a2 = GltfScene(p1); a2.upload(r)
    assert np.array_equal(img_a, frame_of(r, a2, s1))
    for sc in (a, b, b2, a2):
        sc.close()
    r.close(); fresh.close()
