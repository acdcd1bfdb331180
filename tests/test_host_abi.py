"""CPU-only tests of the boundary and host logic: the C-ABI library loads and exports every symbol that
include/mipt.h declares (no compute calls without a GPU), struct layouts, camera conventions
(Source/Camera.h:80-92, Source/CameraController.h:42-49), scene generators."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

from gltf_renderer_amd import abi, camera, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mipt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", text)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from gltf_renderer_amd import renderer
    L = renderer.load_library()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "libmipt.so does not export %s" % s
    assert sorted(renderer.EXPORTS) == syms
    assert L.pt_abi_version() == 2
    # the scene side (include/mipt_scene.h): loader, animation, image readers
    from gltf_renderer_amd import gltf
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mipt_scene.h")).read(), flags=re.S)
    scene_syms = sorted(set(re.findall(r"\b((?:gs|img)_[a-z_0-9]+)\s*\(", text)))
    assert len(scene_syms) >= 25 and sorted(gltf.SCENE_EXPORTS) == scene_syms
    for s in scene_syms:
        assert hasattr(L, s), "libmipt.so does not export %s" % s


def test_product_fails_loudly_without_gpu():
    import torch
    from gltf_renderer_amd.renderer import Renderer, MiptError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(MiptError):
        Renderer()


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under gltf_renderer_amd/ may reference it."""
    pkg = os.path.join(ROOT, "gltf_renderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in txt and "liboracle" not in txt and "oracle/" not in txt.replace("the CPU oracle", ""), f


def test_struct_layouts_match_reference_sizes():
    assert C.sizeof(abi.PtSettings) == 64          # Pathtracer::Settings
    assert C.sizeof(abi.PtLight) == 64             # GpuLight / Light
    assert C.sizeof(abi.PtTextureSample) == 32     # TextureSample / TextureAddress
    assert C.sizeof(abi.PtMaterial) == 640         # GpuMaterial / Material
    assert C.sizeof(abi.PtMeshInstance) == 156     # GpuMeshInstance / Instance
    assert C.sizeof(abi.PtBone) == 128             # GpuSkin::Bone
    # field offsets the HLSL struct implies (Material.hlsli:23-66)
    M = abi.PtMaterial
    assert M.normal.offset == 64 and M.albedo.offset == 96 and M.emissive.offset == 192
    assert M.specular_factor.offset == 224 and M.specular.offset == 240 and M.clearcoat_factor.offset == 304
    assert M.clearcoat.offset == 320 and M.anisotropy_strength.offset == 416 and M.anisotropy.offset == 432
    assert M.sheen_color_factor.offset == 464 and M.sheen_color.offset == 480 and M.transmission_factor.offset == 544
    assert M.transmission.offset == 560 and M.attenuation_distance.offset == 592 and M.thickness.offset == 608
    L = abi.PtLight
    assert L.position.offset == 4 and L.cutoff.offset == 16 and L.direction.offset == 20 and L.intensity.offset == 32
    assert L.color.offset == 36 and L.inner_angle.offset == 48 and L.outer_angle.offset == 52
    I = abi.PtMeshInstance
    assert I.normal_transform.offset == 64 and I.index_descriptor.offset == 128 and I.material_id.offset == 152
    S = abi.PtSettings
    assert S.debug_output.offset == 12 and S.flags.offset == 16 and S.environment_color.offset == 20 and S.seed.offset == 40
    assert S.max_accumulated_frames.offset == 56 and S.max_ray_length.offset == 60


def test_settings_defaults_follow_reference():
    s = abi.PtSettings.defaults()               # Pathtracer.h:71-84
    assert (s.min_bounces, s.max_bounces, s.environment_intensity, s.luminance_clamp) == (2, 2, 1.0, 1000.0)
    assert s.flags == abi.FLAG_ACCUMULATE | abi.FLAG_POINT_LIGHTS | abi.FLAG_ENVIRONMENT_MAP
    assert s.max_accumulated_frames == 65536 and s.use_frame_as_seed == 1
    a = abi.PtSettings.app_defaults()           # Main.cpp:462-474
    assert a.flags == 0x9364 and a.luminance_clamp == 20.0 and a.max_accumulated_frames == 8196
    assert abi.FLAG_NONE == 1                    # FLAG_NONE really is bit 0 (quirk q1)


def test_reversed_z_projection():
    P = camera.view_to_clip(16 / 9, math.pi / 2, 0.01, 100.0)
    near = P @ np.array([0, 0, -0.01, 1.0]); far = P @ np.array([0, 0, -100.0, 1.0])
    assert abs(near[2] / near[3] - 1.0) < 1e-9 and abs(far[2] / far[3]) < 1e-9       # near -> 1, far -> 0
    assert abs(P[1, 1] - 1.0) < 1e-12 and abs(P[0, 0] - 9 / 16) < 1e-12               # fov 90 deg
    Pinf = camera.view_to_clip(1.0, math.pi / 2, 0.01, 0.0)                            # z_far == 0 -> 100000
    f = Pinf @ np.array([0, 0, -100000.0, 1.0])
    assert abs(f[2] / f[3]) < 1e-9


def test_orbit_and_free_camera_conventions():
    V = camera.orbit_world_to_view()            # the app's g_orbit(centre 0, radius 1, angles 0)
    eye = np.linalg.inv(V) @ np.array([0, 0, 0, 1.0])
    assert np.allclose(eye[:3], (0, -1, 0))      # eye at world (0,-1,0)
    fwd = np.linalg.inv(V) @ np.array([0, 0, -1, 0.0]); up = np.linalg.inv(V) @ np.array([0, 1, 0, 0.0])
    assert np.allclose(fwd[:3], (0, 1, 0)) and np.allclose(up[:3], (0, 0, 1))          # looking +Y, up +Z (Z-up world)
    V = camera.orbit_world_to_view((1, 2, 3), 2.0, 0.0, -0.5)
    eye = (np.linalg.inv(V) @ np.array([0, 0, 0, 1.0]))[:3]
    assert abs(np.linalg.norm(eye - np.array([1, 2, 3])) - 2.0) < 1e-9 and eye[2] > 3   # negative inclination looks down from above
    F = camera.free_world_to_view((0, -1, 0), 0, 0)
    assert np.allclose(F, camera.orbit_world_to_view())
    F = camera.free_world_to_view((0, 0, 0), -math.pi / 2, 0)
    fwd = np.linalg.inv(F) @ np.array([0, 0, -1, 0.0])
    assert np.allclose(fwd[:3], (1, 0, 0), atol=1e-9)


def test_y_up_to_z_up_root():
    v = camera.Y_UP_TO_Z_UP @ np.array([0, 1, 0, 0.0])       # glTF +Y (up) becomes world +Z
    assert np.allclose(v[:3], (0, 0, 1))
    assert abs(np.linalg.det(camera.Y_UP_TO_Z_UP[:3, :3]) - 1) < 1e-12


def test_scene_generators_match_baseline_classes():
    s1 = scenes.single_triangle()
    assert s1.triangles == 1 and (s1.width, s1.height) == (256, 256) and s1.settings.max_bounces == 1
    assert s1.settings.flags == abi.FLAG_MATERIAL_DIFFUSE_WHITE and s1.settings.use_frame_as_seed == 0
    s5 = scenes.skinned_figure()
    assert 4000 < s5.triangles < 6000 and (s5.width, s5.height) == (3840, 2160) and len(s5.skins) == 1
    assert len(scenes.JOINT_NAMES) == 19
    fig = s5.skins[0]["mesh"]
    assert np.allclose(fig.weights.sum(axis=1), 1.0) and fig.joints.max() < 19
    s4 = scenes.material_grid(64)
    assert len(s4.materials) == 38 and s4.settings.max_bounces == 16 and s4.bounce_limit == 16
    # instance flags derive from the material exactly as Pathtracer::BuildTlas does (Pathtracer.cpp:216-228)
    t = scenes.test_scene(16, 8)
    for d in t.instances:
        m = t.materials[d.gpu.material_id]
        assert bool(d.instance_flags & abi.INSTANCE_FLAG_TRIANGLE_CULL_DISABLE) == bool(m.flags & abi.MATERIAL_FLAG_DOUBLE_SIDED)
        assert bool(d.instance_flags & abi.INSTANCE_FLAG_FORCE_NON_OPAQUE) == (m.alpha_mode == abi.ALPHA_MODE_MASK)
        assert d.instance_mask == (abi.MASK_ALPHA_BLEND if m.alpha_mode == abi.ALPHA_MODE_BLEND else abi.MASK_NONE)
        nt = np.array(d.gpu.normal_transform).reshape(4, 4).T
        tt = np.array(d.gpu.transform).reshape(4, 4).T
        assert np.allclose(nt, np.linalg.inv(tt).T, atol=1e-5)


def test_sponza_class_scene_shape():
    s = scenes.sponza_class(tex=16)
    assert 240000 <= s.triangles <= 290000
    assert 95 <= len(s.instances) <= 110 and len(s.materials) == 26 and len(s.textures) == 40
    assert len(s.lights) == 6 and sum(1 for l in s.lights if l.type == abi.LIGHT_POINT) == 4
    assert sum(1 for d in s.instances if d.instance_flags & abi.INSTANCE_FLAG_FORCE_NON_OPAQUE) == 10
    assert (s.width, s.height) == (1920, 1080) and s.settings.max_bounces == 8 and s.bounce_limit == 8


def test_unorm_division_shortcut_is_exact(tmp_path):
    """pt_math.h unorm_div<C>: q = x * RN(1/C); q += fma(-C, q, x) * RN(1/C) must equal the IEEE quotient x / C bit for bit for
    every integer x the decoders feed it (8-, 10- and 16-bit unorm).  Checked on the host with the same three operations."""
    import shutil, subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "divc.c"
    src.write_text(r'''
#include <math.h>
#include <stdio.h>
#include <string.h>
static int check(int C, int maxx) {
    const float c = (float)C, inv = 1.0f / c;
    int bad = 0;
    for (int x = 0; x <= maxx; x++) {
        float fx = (float)x, q = fx * inv, r = fmaf(fmaf(-c, q, fx), inv, q), ref = fx / c;
        if (memcmp(&r, &ref, 4)) bad++;
    }
    return bad;
}
int main(void) { int b = check(255, 65535) + check(1023, 65535) + check(65535, 65535); printf("%d\n", b); return b != 0; }
''')
    exe = tmp_path / "divc"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"])
    assert subprocess.run([str(exe)], capture_output=True, text=True).stdout.strip() == "0"


def test_reciprocal_division_of_slot_numbers_is_exact(tmp_path):
    """pt_types.h FastDiv (the kernels' slot / pixel_slots and tile / tiles_x): mulhi(x, ceil(2^(31+L) / d)) >> (L - 1) must equal x / d for
    every x below 2^31.  The struct's own host constructor is compiled from the header; divisors = frame sizes the configs produce plus
    the awkward ones (1, powers of two and their neighbours, the largest)."""
    import shutil, subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    src = tmp_path / "fastdiv.cpp"
    src.write_text(r'''
#include <cstdint>
#include <cstdio>
struct float2 { float x, y; }; struct float4 { float x, y, z, w; }; struct uint2 { unsigned x, y; }; struct uint4 { unsigned x, y, z, w; };
#define PT_TYPES_HOST_TEST 1
#include "pt_types_fastdiv.h"
static uint32_t dv(const FastDiv& f, uint32_t x) { return f.d == 1u ? x : (uint32_t)(((uint64_t)x * f.mul) >> 32) >> f.shift; }
int main() {
    const uint32_t ds[] = {1, 2, 3, 5, 7, 8, 15, 16, 17, 120, 240, 64, 255, 256, 257, 1000, 8160u * 256u, 32400u * 256u, 4096u * 256u, 65535, 65536, 65537,
                           1u << 20, (1u << 20) + 1, 0x7fffffffu, 0x40000000u, 0x40000001u, 12345679u};
    unsigned long long bad = 0;
    for (uint32_t d : ds) {
        const FastDiv f = FastDiv::make(d);
        for (uint64_t x = 0; x < 0x80000000ull; x++) {
            if (dv(f, (uint32_t)x) != (uint32_t)x / d) bad++;
            if (x > 2000000 && x < 0x7ffe0000ull) x += 1021;          // dense at both ends, a stride in between
        }
        for (uint64_t k = 1; k * d < 0x80000000ull && k < 200000; k++) {   // and around every multiple of d
            const uint32_t m = (uint32_t)(k * d);
            if (dv(f, m) != m / d || dv(f, m - 1) != (m - 1) / d) bad++;
        }
    }
    printf("%llu\n", bad);
    return bad != 0;
}
''')
    # the struct, cut from the header so that the test compiles exactly what the library compiles
    import re
    hdr = open(os.path.join(ROOT, "gltf_renderer_amd", "csrc", "pt_types.h")).read()
    m = re.search(r"struct FastDiv \{.*?\n\};", hdr, re.S)
    assert m, "FastDiv not found in pt_types.h"
    (tmp_path / "pt_types_fastdiv.h").write_text(m.group(0) + "\n")
    exe = tmp_path / "fastdiv"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", str(tmp_path), "-o", str(exe), str(src)])
    assert subprocess.run([str(exe)], capture_output=True, text=True).stdout.strip() == "0"


def test_rccl_that_cannot_be_loaded_is_reported_not_crashed():
    """rccl() built its message from two dlerror() calls (the second returns NULL: undefined behaviour) -- a host without a
    loadable librccl must get PT_ERR_NOT_READY from pt_exchange_probe / pt_exchange_unique_id, as include/mipt.h documents.
    MIPT_RCCL_LIBRARY points the loader at a file that does not exist; a fresh process, because the binding is made once."""
    import subprocess
    import sys
    code = ("import ctypes as C, sys; sys.path.insert(0, %r)\n"
            "from gltf_renderer_amd import renderer\n"
            "L = renderer.load_library()\n"
            "buf = (C.c_ubyte * 128)()\n"
            "print(L.pt_exchange_probe(), L.pt_exchange_unique_id(buf), L.pt_exchange_probe())\n" % ROOT)
    env = dict(os.environ, MIPT_RCCL_LIBRARY="/nonexistent/librccl.so.1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["-6", "-6", "-6"], out.stdout           # PT_ERR_NOT_READY, three times, no crash
    # and with the default search list the library is found here (ROCm image): the probe does not touch a GPU
    out = subprocess.run([sys.executable, "-c", code], env={k: v for k, v in os.environ.items() if k != "MIPT_RCCL_LIBRARY"}, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[0] == "0", out.stdout
