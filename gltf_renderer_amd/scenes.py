"""Deterministic synthetic scenes standing in for BASELINE.json's configs.

No glTF / HDR assets exist offline, so every config is restated as a seeded procedural scene of the
same class (triangle count, primitive count, texture set, light set, material features).  A scene is
plain data in exactly the layouts of include/mipt.h; `SceneData.upload(backend)` feeds it to any
object with the Renderer method surface (the product's Renderer, or the CPU oracle in tests).
"""
import math

import numpy as np

from . import abi, camera, meshgen
from .abi import PtMaterial, PtTextureSample, PtLight, PtInstanceDesc, PtSettings, PtExecuteParams

f32 = np.float32


class SceneData:
    def __init__(self, name):
        self.name = name
        self.buffers = []      # (ndarray, format)
        self.textures = []     # (rgba8 HxWx4, srgb)
        self.samplers = []     # (au, av, minf, magf); scene-local index i+1 (0 = default sampler)
        self.materials = [PtMaterial.default()]   # index 0 = default material (Gltf.cpp:470-475)
        self.lights = []
        self.instances = []    # PtInstanceDesc with SCENE-LOCAL buffer indices
        self.env_image = None  # equirect float32 HxWx3
        self.world_to_view = camera.orbit_world_to_view()
        self.y_fov, self.z_near, self.z_far = math.pi / 2, 0.01, 100.0
        self.ortho = None      # (x_mag, y_mag): Camera::Orthographic (Camera.h:31-40) instead of the perspective camera
        self.width, self.height = 256, 256
        self.settings = PtSettings.app_defaults()
        self.bounce_limit = abi.REFERENCE_MAX_BOUNCES
        self.skins = []        # dynamic-mesh records (config 5)
        self.mesh_records = [] # (Mesh, 4x4 transform, material_id) per instance
        self.triangles = 0

    # ---- building -------------------------------------------------------------------------------
    def add_buffer(self, arr, fmt):
        self.buffers.append((np.ascontiguousarray(arr), fmt))
        return len(self.buffers) - 1

    def add_texture(self, rgba8, srgb):
        self.textures.append((np.ascontiguousarray(rgba8, dtype=np.uint8), bool(srgb)))
        return len(self.textures) - 1

    def add_sampler(self, au=abi.ADDRESS_WRAP, av=abi.ADDRESS_WRAP, minf=abi.FILTER_LINEAR, magf=abi.FILTER_LINEAR):
        self.samplers.append((au, av, minf, magf))
        return len(self.samplers)            # scene-local handle (0 is the default sampler)

    def add_material(self, m):
        self.materials.append(m)
        return len(self.materials) - 1

    def add_mesh(self, mesh, transform=None, material_id=0, dynamic=False):
        """Gltf::LoadPrimitive streams (Gltf.cpp:182-321) + one BuildTlas instance (Pathtracer.cpp:196-250)."""
        T = np.eye(4) if transform is None else np.asarray(transform, dtype=np.float64)
        d = PtInstanceDesc()
        g = d.gpu
        g.transform[:] = camera.cm(T)
        g.normal_transform[:] = camera.cm(camera.inverse_transpose(T))
        idx, ifmt = mesh.index_stream()
        g.index_descriptor = self.add_buffer(idx, ifmt) if idx is not None else -1
        g.position_descriptor = self.add_buffer(mesh.positions, abi.FORMAT_R32G32B32_FLOAT)
        ts = mesh.tangent_space_stream()
        g.tangent_space_descriptor = self.add_buffer(ts, abi.FORMAT_R10G10B10A2_UNORM) if ts is not None else -1
        g.texcoord_descriptors[0] = self.add_buffer(mesh.uv0, abi.FORMAT_R32G32_FLOAT) if mesh.uv0 is not None else -1
        g.texcoord_descriptors[1] = self.add_buffer(mesh.uv1, abi.FORMAT_R32G32_FLOAT) if mesh.uv1 is not None else -1
        g.color_descriptor = self.add_buffer(meshgen.pack_unorm16(mesh.colors), abi.FORMAT_R16G16B16A16_UNORM) if mesh.colors is not None else -1
        g.material_id = material_id
        mat = self.materials[material_id]
        d.instance_flags = 0
        if mat.flags & abi.MATERIAL_FLAG_DOUBLE_SIDED:
            d.instance_flags |= abi.INSTANCE_FLAG_TRIANGLE_CULL_DISABLE
        if mat.alpha_mode == abi.ALPHA_MODE_MASK:
            d.instance_flags |= abi.INSTANCE_FLAG_FORCE_NON_OPAQUE
        d.instance_mask = abi.MASK_ALPHA_BLEND if mat.alpha_mode == abi.ALPHA_MODE_BLEND else abi.MASK_NONE
        d.num_of_vertices = mesh.num_vertices
        d.num_of_indices = mesh.num_indices
        d.dynamic = 1 if dynamic else 0
        self.instances.append(d)
        self.mesh_records.append((mesh, T, material_id))       # sources kept so a scene can be written out as glTF (tests/scene_export.py)
        self.triangles += mesh.num_indices // 3
        return len(self.instances) - 1

    def add_light(self, type_, position=(0, 0, 0), direction=(0, 0, -1), color=(1, 1, 1), intensity=1.0, cutoff=0.0,
                  inner=0.0, outer=math.pi / 4):
        l = PtLight()
        l.type = type_
        l.position[:] = position
        l.cutoff = cutoff
        d = np.asarray(direction, dtype=np.float64)
        l.direction[:] = d / np.linalg.norm(d)
        l.intensity = intensity
        l.color[:] = color
        l.inner_angle, l.outer_angle = inner, outer
        self.lights.append(l)

    # ---- camera / params ---------------------------------------------------------------------------
    def execute_params(self, frame=0, tile_rank=0, tile_rank_count=1, env_handle=None):
        p = PtExecuteParams()
        p.world_to_view[:] = camera.cm(self.world_to_view)
        if self.ortho is not None:
            p.view_to_clip[:] = camera.cm(camera.ortho_view_to_clip(self.ortho[0], self.ortho[1], self.z_near, self.z_far))
        else:
            p.view_to_clip[:] = camera.cm(camera.view_to_clip(self.width / self.height, self.y_fov, self.z_near, self.z_far))
        p.width, p.height = self.width, self.height
        p.frame = frame
        p.light_count = len(self.lights)
        p.environment_map = -1 if env_handle is None else env_handle
        p.output = None
        p.tile_rank, p.tile_rank_count = tile_rank, tile_rank_count
        return p

    # ---- upload ---------------------------------------------------------------------------------
    def upload(self, backend, env_raw=None):
        """Returns a dict with the backend's handles.  env_raw = (N, cube, pyramid) feeds maps that were
        preprocessed elsewhere (tracer-only parity tests); otherwise backend.env_create runs K10-K13."""
        bmap = [backend.buffer_create(a, fmt) for a, fmt in self.buffers]
        tmap = [backend.texture_create(t, srgb) for t, srgb in self.textures]
        smap = [0] + [backend.sampler_create(*s) for s in self.samplers]
        mats = []
        for m in self.materials:
            c = PtMaterial.from_buffer_copy(bytes(m))
            for slot in PtMaterial.TEXTURE_SLOTS:
                ts = getattr(c, slot)
                if ts.descriptor != -1:
                    ts.descriptor = tmap[ts.descriptor]
                ts.sampler = smap[ts.sampler]
            mats.append(c)
        backend.set_materials(mats)
        backend.set_lights(self.lights)
        insts = []
        for d in self.instances:
            c = PtInstanceDesc.from_buffer_copy(bytes(d))
            g = c.gpu
            for name in ("index_descriptor", "position_descriptor", "tangent_space_descriptor", "color_descriptor"):
                v = getattr(g, name)
                if v != -1:
                    setattr(g, name, bmap[v])
            for k in range(2):
                if g.texcoord_descriptors[k] != -1:
                    g.texcoord_descriptors[k] = bmap[g.texcoord_descriptors[k]]
            insts.append(c)
        backend.set_instances(insts)
        env = None
        if env_raw is not None:
            env = backend.env_create_raw(*env_raw)
        elif self.env_image is not None:
            env = backend.env_create(self.env_image)
        if hasattr(backend, "set_bounce_limit"):
            backend.set_bounce_limit(self.bounce_limit)
        return {"buffers": bmap, "textures": tmap, "samplers": smap, "env": env, "instances": insts}


# ---- procedural images ----------------------------------------------------------------------------
def value_noise(rng, size, octaves=5, base=4):
    out = np.zeros((size, size), np.float32)
    amp, tot = 1.0, 0.0
    for o in range(octaves):
        n = base << o
        g = rng.random((n + 1, n + 1), dtype=np.float32)
        g[-1, :] = g[0, :]; g[:, -1] = g[:, 0]                     # tileable
        t = np.linspace(0, n, size, endpoint=False, dtype=np.float32)
        i = np.floor(t).astype(np.int32); f = t - i
        f = f * f * (3 - 2 * f)
        a = g[i][:, i]; b = g[i][:, i + 1]; c = g[i + 1][:, i]; d = g[i + 1][:, i + 1]
        fx, fy = f[None, :], f[:, None]
        out += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
        tot += amp; amp *= 0.5
    return out / tot


def to_u8(x):
    return np.clip(np.asarray(x) * 255.0 + 0.5, 0, 255).astype(np.uint8)


def rgba(r, g, b, a=None):
    a = np.ones_like(r) if a is None else a
    return np.stack([to_u8(r), to_u8(g), to_u8(b), to_u8(a)], axis=2)


def normal_map_from_height(h, strength=4.0):
    dx = (np.roll(h, -1, axis=1) - np.roll(h, 1, axis=1)) * strength * h.shape[1] / 64.0
    dy = (np.roll(h, -1, axis=0) - np.roll(h, 1, axis=0)) * strength * h.shape[0] / 64.0
    n = np.stack([-dx, -dy, np.ones_like(h)], axis=2)
    n /= np.linalg.norm(n, axis=2, keepdims=True)
    return rgba(n[..., 0] * 0.5 + 0.5, n[..., 1] * 0.5 + 0.5, n[..., 2] * 0.5 + 0.5)


def sky_image(width=2048, height=1024, sun_radiance=1.0e4, sun_dir=(0.3, -0.5, 0.8), sun_cos=0.9995, seed=7):
    """Procedural sky in the reference's env parameterisation (quirk q8): u = atan2(y,x)/2pi (wrap),
    v = 1 - (z+1)/2 (equal-area in z, Z-up): gradient + one bright sun disc."""
    u = (np.arange(width, dtype=np.float64) + 0.5) / width
    v = (np.arange(height, dtype=np.float64) + 0.5) / height
    z = 1.0 - 2.0 * v
    phi = 2 * np.pi * u
    r = np.sqrt(np.maximum(1 - z * z, 0))
    d = np.stack([r[:, None] * np.cos(phi)[None, :], r[:, None] * np.sin(phi)[None, :], np.repeat(z[:, None], width, 1)], axis=2)
    t = np.clip(d[..., 2] * 0.5 + 0.5, 0, 1)
    horizon = np.array([0.9, 0.85, 0.8]); zenith = np.array([0.15, 0.35, 0.9]); ground = np.array([0.25, 0.22, 0.2])
    sky = np.where((d[..., 2] >= 0)[..., None], horizon + (zenith - horizon) * (np.clip(d[..., 2], 0, 1) ** 0.5)[..., None],
                   ground[None, None] * (0.4 + 0.6 * t[..., None]))
    rng = np.random.default_rng(seed)
    clouds = value_noise(rng, height, 4, 2)
    clouds = np.repeat(clouds, width // height, axis=1) if width > height else clouds[:, :width]
    sky = sky * (0.85 + 0.3 * clouds[..., None])
    s = np.asarray(sun_dir, dtype=np.float64); s /= np.linalg.norm(s)
    cosang = d @ s
    sky = sky + (cosang > sun_cos)[..., None] * sun_radiance * np.array([1.0, 0.95, 0.85])
    return np.ascontiguousarray(sky.astype(np.float32))


# ---- config 1: single triangle ------------------------------------------------------------------------
def single_triangle(size=256):
    """BASELINE config 1: one triangle (-1,0,-1)(1,0,-1)(0,0,1), default material, 256^2, 1 bounce,
    fixed seed, diffuse-white override, no env map -> constant environment colour (1,1,1)."""
    s = SceneData("config1_single_triangle")
    m = meshgen.Mesh(np.array([[-1, 0, -1], [1, 0, -1], [0, 0, 1]], f32), np.array([0, 1, 2]),
                     normals=np.array([[0, -1, 0]] * 3, f32), uv0=np.array([[0, 1], [1, 1], [0.5, 0]], f32))
    s.add_mesh(m, None, 0)
    s.world_to_view = camera.orbit_world_to_view((0, 0, 0), 3.0, 0.0, 0.0)
    s.width = s.height = size
    st = PtSettings.defaults()
    st.min_bounces = st.max_bounces = 1
    st.flags = abi.FLAG_MATERIAL_DIFFUSE_WHITE
    st.environment_color[:] = (1, 1, 1)
    st.use_frame_as_seed = 0
    st.seed = 1
    s.settings = st
    return s


# ---- small all-feature scene for parity tests ---------------------------------------------------------
def material(**kw):
    m = PtMaterial.default()
    for k, v in kw.items():
        cur = getattr(m, k)
        if hasattr(cur, "__len__") and not isinstance(cur, PtTextureSample):
            cur[:] = v
        else:
            setattr(m, k, v)
    return m


def test_scene(size=128, tex=64, with_env=True, seed=3):
    """Small scene exercising every code path: textured/normal-mapped sphere, metal, clearcoat, sheen,
    anisotropy, transmission, MASK cut-out quad, BLEND quad, vertex colours, texture transform,
    mirrored instance, non-indexed mesh, mesh without tangent space, point/spot/directional lights."""
    rng = np.random.default_rng(seed)
    s = SceneData("test_scene")
    h = value_noise(rng, tex, 4, 2)
    t_base = s.add_texture(rgba(0.3 + 0.7 * h, 0.4 + 0.5 * value_noise(rng, tex, 3, 2), 0.5 * h, 0.2 + 0.8 * value_noise(rng, tex, 2, 4)), True)
    t_mr = s.add_texture(rgba(np.ones_like(h), 0.2 + 0.8 * value_noise(rng, tex, 3, 2), value_noise(rng, tex, 2, 2)), False)
    t_nrm = s.add_texture(normal_map_from_height(h, 2.0), False)
    t_em = s.add_texture(rgba(h ** 4, 0.5 * h ** 4, 0.1 * h), True)
    checker = ((np.indices((tex, tex)).sum(axis=0) // max(tex // 8, 1)) % 2).astype(np.float32)
    t_mask = s.add_texture(rgba(0.2 + 0.6 * checker, 0.8 * np.ones_like(h), 0.3 * np.ones_like(h), checker), True)
    t_misc = s.add_texture(rgba(value_noise(rng, tex, 2, 2), value_noise(rng, tex, 2, 2), value_noise(rng, tex, 2, 2), value_noise(rng, tex, 2, 2)), False)
    smp_clamp = s.add_sampler(abi.ADDRESS_CLAMP, abi.ADDRESS_MIRROR, abi.FILTER_LINEAR, abi.FILTER_LINEAR)
    smp_point = s.add_sampler(abi.ADDRESS_WRAP, abi.ADDRESS_WRAP, abi.FILTER_POINT, abi.FILTER_POINT)
    TS = PtTextureSample
    m_tex = s.add_material(material(albedo=TS(t_base), metallic_roughness=TS(t_mr), normal=TS(t_nrm, 0, 0, 0.3, (0.1, 0.2), (2.0, 1.5)),
                                    emissive=TS(t_em), emissive_factor=(2.0, 2.0, 2.0), occlusion=TS(t_mr), normal_scale=0.8))
    m_metal = s.add_material(material(base_color_factor=(0.95, 0.8, 0.4, 1), metalness_factor=1.0, roughness_factor=0.25,
                                      anisotropy_strength=0.7, anisotropy_rotation=0.6, anisotropy=TS(t_misc)))
    m_coat = s.add_material(material(base_color_factor=(0.7, 0.05, 0.05, 1), metalness_factor=0.0, roughness_factor=0.6, clearcoat_factor=1.0,
                                     clearcoat_roughness_factor=0.1, clearcoat_normal=TS(t_nrm), clearcoat=TS(t_misc), clearcoat_roughness=TS(t_misc, smp_point)))
    m_sheen = s.add_material(material(base_color_factor=(0.1, 0.1, 0.5, 1), metalness_factor=0.0, roughness_factor=0.9,
                                      sheen_color_factor=(0.9, 0.8, 1.0), sheen_roughness_factor=0.5, sheen_color=TS(t_base), sheen_roughness=TS(t_base)))
    m_glass = s.add_material(material(base_color_factor=(0.9, 1.0, 0.95, 1), metalness_factor=0.0, roughness_factor=0.15, transmission_factor=0.9,
                                      ior=1.45, transmission=TS(t_misc), thickness=TS(t_misc), specular_factor=0.8, specular_color_factor=(1, 0.9, 0.8),
                                      specular=TS(t_base), specular_color=TS(t_base)))
    m_mask = s.add_material(material(flags=abi.MATERIAL_FLAG_DOUBLE_SIDED, alpha_mode=abi.ALPHA_MODE_MASK, alpha_cutoff=0.5, albedo=TS(t_mask, smp_clamp),
                                     metalness_factor=0.0, roughness_factor=0.8))
    m_blend = s.add_material(material(flags=abi.MATERIAL_FLAG_DOUBLE_SIDED, alpha_mode=abi.ALPHA_MODE_BLEND, base_color_factor=(0.2, 0.9, 0.3, 0.5),
                                      metalness_factor=0.0, roughness_factor=0.5))
    m_floor = s.add_material(material(albedo=TS(t_mask, 0, 0, 0, (0, 0), (4, 4)), metalness_factor=0.0, roughness_factor=0.7))
    # floor z=0, back wall
    s.add_mesh(meshgen.grid(8, 8, (-3, -3, 0), (6, 0, 0), (0, 6, 0)), None, m_floor)
    wall = meshgen.grid(4, 4, (-3, 3, 0), (6, 0, 0), (0, 0, 3))
    wall.colors = np.concatenate([np.random.default_rng(1).random((wall.num_vertices, 3)), np.ones((wall.num_vertices, 1))], axis=1)
    wall.uv1 = wall.uv0 * 3.0
    s.add_mesh(wall, None, 0)
    sp = meshgen.uv_sphere(24, 12, 0.6)
    s.add_mesh(sp, camera.trs((-1.6, 0.5, 0.6)), m_tex)
    s.add_mesh(sp, camera.trs((0.0, 0.8, 0.6), scale=(1, 1, 1.2)), m_metal)
    s.add_mesh(sp, camera.trs((1.6, 0.5, 0.6)), m_coat)
    s.add_mesh(sp, camera.trs((-0.9, -0.8, 0.45), scale=(0.75, 0.75, 0.75)), m_sheen)
    s.add_mesh(sp, camera.trs((0.9, -0.8, 0.45), scale=(-0.75, 0.75, 0.75)), m_glass)      # mirrored instance
    s.add_mesh(meshgen.grid(2, 2, (-0.6, -1.6, 0.1), (1.2, 0, 0), (0, 0.2, 1.2)), None, m_mask)
    s.add_mesh(meshgen.grid(1, 1, (1.6, -1.4, 0.1), (0.9, 0.3, 0), (0, 0, 1.0)), None, m_blend)
    # non-indexed mesh without tangent space or uvs (GenerateTangent path)
    tri = meshgen.Mesh(np.array([[-2.6, -1.5, 0.05], [-1.8, -1.7, 0.05], [-2.2, -1.2, 1.1]], f32), None)
    s.add_mesh(tri, None, 0)
    # normals-only mesh (EncodeNormal path: angle bits 0)
    b = meshgen.box((2.0, 1.2, 0.0), (2.6, 1.8, 0.9))
    b.tangents = None
    s.add_mesh(b, None, m_coat)
    s.add_light(abi.LIGHT_POINT, position=(0.0, -1.0, 2.5), color=(1, 0.9, 0.8), intensity=12.0, cutoff=0.0)
    s.add_light(abi.LIGHT_POINT, position=(-2.0, 1.0, 1.5), color=(0.4, 0.5, 1.0), intensity=6.0, cutoff=6.0)
    s.add_light(abi.LIGHT_SPOT, position=(2.0, -2.0, 2.5), direction=(-0.5, 0.6, -0.7), color=(1, 1, 1), intensity=30.0, inner=0.2, outer=0.5)
    s.add_light(abi.LIGHT_DIRECTIONAL, direction=(0.3, 0.4, -0.8), color=(1, 1, 0.9), intensity=1.5)
    if with_env:
        s.env_image = sky_image(256, 128, 300.0)
    s.world_to_view = camera.orbit_world_to_view((0, 0, 0.6), 3.6, 0.35, -0.45)
    s.width = s.height = size
    st = PtSettings.app_defaults()
    st.min_bounces, st.max_bounces = 2, 4
    if not with_env:
        st.flags &= ~(abi.FLAG_ENVIRONMENT_MAP | abi.FLAG_ENVIRONMENT_MIS)
        st.environment_color[:] = (0.6, 0.7, 0.9)
    s.settings = st
    return s


# ---- config 2: "Helmet-class" ---------------------------------------------------------------------------
def helmet_class(width=1920, height=1080, subdiv=6, tex=2048, seed=2):
    """BASELINE config 2 stand-in for DamagedHelmet: ONE mesh, displaced icosphere (subdiv 6 = 81,920
    triangles), 5 procedural RGBA8 textures (base sRGB, metal-rough, normal, occlusion, emissive sRGB),
    tangents present, 2048x1024 procedural sky with a 1e4-radiance sun, 4 bounces, app-default flags
    minus POINT_LIGHTS."""
    rng = np.random.default_rng(seed)
    s = SceneData("config2_helmet_class")
    v, f = meshgen.icosphere(subdiv)
    # smooth displacement from low-frequency noise on the sphere
    disp = np.zeros(len(v))
    for k in range(1, 5):
        w = rng.normal(size=(6, 3)) * k
        ph = rng.uniform(0, 2 * np.pi, 6)
        disp += (np.sin(v @ w.T + ph).sum(axis=1)) * 0.03 / k
    p = v * (1.0 + disp)[:, None]
    n = meshgen.vertex_normals(p, f)
    t, uv = meshgen.sphere_tangent_frame(v)
    t[:, :3] = t[:, :3] - n * (t[:, :3] * n).sum(axis=1, keepdims=True)
    t[:, :3] /= np.maximum(np.linalg.norm(t[:, :3], axis=1, keepdims=True), 1e-9)
    mesh = meshgen.Mesh(p, f.reshape(-1), n, t, uv * np.array([2.0, 1.0]))
    h = value_noise(rng, tex, 6, 4)
    h2 = value_noise(rng, tex, 5, 8)
    t_base = s.add_texture(rgba(0.25 + 0.6 * h, 0.22 + 0.5 * h2, 0.2 + 0.4 * h * h2), True)
    t_mr = s.add_texture(rgba(np.ones_like(h), 0.15 + 0.7 * h2, (h > 0.5).astype(np.float32)), False)
    t_nrm = s.add_texture(normal_map_from_height(h, 3.0), False)
    t_occ = s.add_texture(rgba(0.5 + 0.5 * h, 0.5 + 0.5 * h, 0.5 + 0.5 * h), False)
    em = (h2 > 0.72).astype(np.float32)
    t_em = s.add_texture(rgba(0.2 * em, 0.7 * em, em), True)
    TS = PtTextureSample
    m = s.add_material(material(albedo=TS(t_base), metallic_roughness=TS(t_mr), normal=TS(t_nrm), occlusion=TS(t_occ), emissive=TS(t_em),
                                emissive_factor=(1.0, 1.0, 1.0)))
    s.add_mesh(mesh, camera.Y_UP_TO_Z_UP, m)
    s.env_image = sky_image(2048, 1024, 1.0e4)
    s.world_to_view = camera.orbit_world_to_view((0, 0, 0), 2.5, 0.6, -0.25)
    s.width, s.height = width, height
    st = PtSettings.app_defaults()
    st.min_bounces, st.max_bounces = 2, 4
    st.flags &= ~abi.FLAG_POINT_LIGHTS
    s.settings = st
    return s


# ---- config 3: "Sponza-class" --------------------------------------------------------------------------
def sponza_class(width=1920, height=1080, tex=1024, detail=1.083, seed=3, n_textures=40):
    """BASELINE config 3 stand-in for Sponza: an atrium, ~262 k triangles in ~100 primitives (tessellated
    floor, two storeys of box columns with arches, walls, ceiling beams, hanging cloth quads), 25
    materials, ~40 1024^2 textures, 10 MASK-mode foliage quads, 4 point + 1 spot + 1 directional light,
    free camera inside the atrium, 8 bounces + RR (clamp lifted), app-default flags."""
    rng = np.random.default_rng(seed)
    s = SceneData("config3_sponza_class")
    TS = PtTextureSample
    texs = []
    n_sets = max(n_textures // 3, 1)
    for k in range(n_sets):                          # sets of (base sRGB, metal-rough, normal)
        h = value_noise(rng, tex, 5, 2 + (k % 4))
        g = value_noise(rng, tex, 4, 4)
        tint = rng.uniform(0.3, 1.0, 3)
        b = s.add_texture(rgba(tint[0] * (0.3 + 0.7 * h), tint[1] * (0.3 + 0.7 * g), tint[2] * (0.3 + 0.7 * h * g)), True)
        mr = s.add_texture(rgba(np.ones_like(h), 0.35 + 0.6 * g, 0.0 * h), False)
        nm = s.add_texture(normal_map_from_height(h, 2.0 + k % 3), False)
        texs.append((b, mr, nm))
    leaf = value_noise(rng, tex, 4, 8)
    t_leaf = s.add_texture(rgba(0.1 + 0.3 * leaf, 0.35 + 0.5 * leaf, 0.08 + 0.1 * leaf, (leaf > 0.45).astype(np.float32)), True)
    mats = []
    for k in range(23):
        b, mr, nm = texs[k % len(texs)]
        scale = (1.0 + (k % 3), 1.0 + (k % 3))
        mats.append(s.add_material(material(albedo=TS(b, 0, 0, 0.0, (0, 0), scale), metallic_roughness=TS(mr, 0, 0, 0.0, (0, 0), scale),
                                            normal=TS(nm, 0, 0, 0.0, (0, 0), scale), roughness_factor=0.5 + 0.5 * (k % 2),
                                            metalness_factor=1.0 if k % 7 == 3 else 0.0)))
    m_leaf = s.add_material(material(flags=abi.MATERIAL_FLAG_DOUBLE_SIDED, alpha_mode=abi.ALPHA_MODE_MASK, alpha_cutoff=0.5, albedo=TS(t_leaf),
                                     metalness_factor=0.0, roughness_factor=0.8))
    m_cloth = s.add_material(material(flags=abi.MATERIAL_FLAG_DOUBLE_SIDED, albedo=TS(texs[0][0]), metalness_factor=0.0, roughness_factor=0.9,
                                      sheen_color_factor=(0.6, 0.5, 0.4), sheen_roughness_factor=0.6))
    d = lambda n: max(int(round(n * detail)), 1)
    L, W, H = 24.0, 10.0, 12.0                      # atrium extents (x, y, z)
    mi = iter(range(10 ** 6))
    nm_ = lambda: mats[next(mi) % len(mats)]
    # floor, ceiling, walls  (6 primitives)
    s.add_mesh(meshgen.grid(d(160), d(64), (-L / 2, -W / 2, 0), (L, 0, 0), (0, W, 0), (12, 5)), None, nm_())
    # the atrium is open to the sky like Sponza's: only the two gallery roofs are covered.  (Punctual-light shadow
    # rays run 1000 units past the light, quirk q3, so in a closed room no punctual light would ever be visible.)
    s.add_mesh(meshgen.grid(d(20), d(160), (-L / 2, -W / 2, H), (0, 2.6, 0), (L, 0, 0), (2, 12)), None, nm_())
    s.add_mesh(meshgen.grid(d(20), d(160), (-L / 2, W / 2 - 2.6, H), (0, 2.6, 0), (L, 0, 0), (2, 12)), None, nm_())
    s.add_mesh(meshgen.grid(d(160), d(80), (-L / 2, W / 2, 0), (L, 0, 0), (0, 0, H), (12, 6)), None, nm_())
    s.add_mesh(meshgen.grid(d(80), d(160), (-L / 2, -W / 2, 0), (0, 0, H), (L, 0, 0), (6, 12)), None, nm_())
    s.add_mesh(meshgen.grid(d(80), d(64), (-L / 2, -W / 2, 0), (0, W, 0), (0, 0, H), (5, 6)), None, nm_())
    s.add_mesh(meshgen.grid(d(64), d(80), (L / 2, -W / 2, 0), (0, 0, H), (0, W, 0), (6, 5)), None, nm_())
    # two storeys of columns along both sides + arches between them (2 x 2 x 10 columns, 2 x 2 x 9 arches)
    xs = np.linspace(-L / 2 + 1.5, L / 2 - 1.5, 10)
    for storey in range(2):
        z0 = storey * 5.5
        for side in (-1, 1):
            y = side * (W / 2 - 2.2)
            for x in xs:
                s.add_mesh(meshgen.box((x - 0.35, y - 0.35, z0), (x + 0.35, y + 0.35, z0 + 4.2), (d(6), d(6), d(36)), 2.0), None, nm_())
            for x0, x1 in zip(xs[:-1], xs[1:]):
                na = d(40)
                def arch(sv, tv, x0=x0, x1=x1):
                    return 0.0 * sv
                g = meshgen.grid(na, d(6), (x0 + 0.35, y - 0.3, z0 + 4.2), (x1 - x0 - 0.7, 0, 0), (0, 0.6, 0), (3, 1))
                # bend the strip into an arch (semi-ellipse), recompute normals / tangents
                sv = (g.positions[:, 0] - (x0 + 0.35)) / (x1 - x0 - 0.7)
                g.positions[:, 2] = (z0 + 4.2 + 0.9 * np.sin(np.pi * sv)).astype(f32)
                tx = np.stack([np.full_like(sv, (x1 - x0 - 0.7)), np.zeros_like(sv), 0.9 * np.pi * np.cos(np.pi * sv)], axis=1)
                tx /= np.linalg.norm(tx, axis=1, keepdims=True)
                nrm = np.cross(tx, np.array([0, 1.0, 0])[None])
                g.normals = (-nrm).astype(f32)
                g.tangents = np.concatenate([tx, np.ones((len(tx), 1))], axis=1).astype(f32)
                # grid() wound for +Z normal; after bending the normal is -cross(tx, y) = up-ish: consistent
                s.add_mesh(g, None, nm_())
        # gallery floor slabs on each side (2 per storey)
        if storey == 1:
            for side in (-1, 1):
                y0 = side * (W / 2) if side < 0 else W / 2 - 2.6
                s.add_mesh(meshgen.box((-L / 2, min(y0, y0 + 2.6) if side < 0 else y0, 5.0), (L / 2, (y0 + 2.6) if side < 0 else W / 2, 5.5),
                                       (d(80), d(8), 1), 6.0), None, nm_())
    # hanging cloth (10 wavy double-sided quads)
    for k in range(10):
        x = -L / 2 + 2.5 + k * (L - 5) / 9
        ph = rng.uniform(0, 6.28)
        g = meshgen.grid(d(24), d(48), (x, -1.2, 6.0), (0, 2.4, 0), (0, 0, 4.0), (1, 2),
                         displace=lambda sv, tv, ph=ph: 0.12 * np.sin(6.0 * sv + ph) * (0.3 + tv))
        g.normals = meshgen.vertex_normals(g.positions.astype(np.float64), g.indices.reshape(-1, 3)).astype(f32)
        s.add_mesh(g, None, m_cloth)
    # foliage: 10 MASK-mode quads near the floor
    for k in range(10):
        x = rng.uniform(-L / 2 + 2, L / 2 - 2); y = rng.uniform(-1.5, 1.5); a = rng.uniform(0, np.pi)
        s.add_mesh(meshgen.grid(d(4), d(4), (x - 0.6 * np.cos(a), y - 0.6 * np.sin(a), 0.0), (1.2 * np.cos(a), 1.2 * np.sin(a), 0), (0, 0, 1.4)), None, m_leaf)
    # lights: 4 point + 1 spot + 1 directional
    for k, x in enumerate(np.linspace(-L / 2 + 3, L / 2 - 3, 4)):
        s.add_light(abi.LIGHT_POINT, position=(x, (-1) ** k * 1.5, 4.0), color=(1.0, 0.85, 0.6), intensity=40.0, cutoff=0.0)
    s.add_light(abi.LIGHT_SPOT, position=(0, 0, 11.0), direction=(0.1, 0.0, -1.0), color=(1, 1, 1), intensity=300.0, inner=0.3, outer=0.6)
    s.add_light(abi.LIGHT_DIRECTIONAL, direction=(0.3, -0.2, -0.9), color=(1, 0.95, 0.9), intensity=2.0)
    s.env_image = sky_image(2048, 1024, 1.0e4)
    s.world_to_view = camera.free_world_to_view((-L / 2 + 2.0, -0.5, 2.2), yaw=-math.pi / 2 + 0.15, pitch=0.12)
    s.width, s.height = width, height
    st = PtSettings.app_defaults()
    st.min_bounces, st.max_bounces = 2, 8
    st.min_russian_roulette_continue_prob, st.max_russian_roulette_continue_prob = 0.1, 0.9
    s.settings = st
    s.bounce_limit = 8
    return s


# ---- config 4: material test grid ----------------------------------------------------------------------
def material_grid(size=1024, seg=32, seed=4):
    """BASELINE config 4 stand-in for TransmissionTest + ClearcoatTest: 6x6 spheres (~2 k triangles each)
    sweeping transmission x roughness x ior and clearcoat x clearcoat-roughness (+ clearcoat normal map),
    plus a sheen row and an anisotropy row; 16 bounces (clamp lifted), full flags."""
    rng = np.random.default_rng(seed)
    s = SceneData("config4_material_grid")
    TS = PtTextureSample
    h = value_noise(rng, 512, 5, 4)
    t_nrm = s.add_texture(normal_map_from_height(h, 3.0), False)
    t_chk = s.add_texture(rgba(*(3 * [0.2 + 0.6 * ((np.indices((512, 512)).sum(axis=0) // 32) % 2).astype(np.float32)])), True)
    sp = meshgen.uv_sphere(seg, seg, 0.42)
    iors = [1.0, 1.33, 1.5]
    for r in range(6):
        for c in range(6):
            a, b = c / 5.0, (r % 2 * 3 + c % 3) / 5.0
            if r < 2:      # transmission sweep
                m = material(base_color_factor=(0.9, 0.95, 1.0, 1), metalness_factor=0.0, transmission_factor=a, roughness_factor=0.05 + 0.9 * b,
                             ior=iors[(r * 6 + c) % 3])
            elif r < 4:    # clearcoat sweep
                m = material(base_color_factor=(0.8, 0.1, 0.1, 1), metalness_factor=0.0, roughness_factor=0.7, clearcoat_factor=a,
                             clearcoat_roughness_factor=b, clearcoat_normal=TS(t_nrm) if r == 3 else TS())
            elif r == 4:   # sheen
                m = material(base_color_factor=(0.1, 0.1, 0.4, 1), metalness_factor=0.0, roughness_factor=0.8, sheen_color_factor=(a, a, 1.0 - 0.5 * a),
                             sheen_roughness_factor=0.1 + 0.8 * b)
            else:          # anisotropy
                m = material(base_color_factor=(0.9, 0.7, 0.3, 1), metalness_factor=1.0, roughness_factor=0.3, anisotropy_strength=a,
                             anisotropy_rotation=b * math.pi)
            s.add_mesh(sp, camera.trs(((c - 2.5) * 1.0, 0.0, (r - 2.5) * 1.0 + 3.0)), s.add_material(m))
    m_floor = s.add_material(material(albedo=TS(t_chk, 0, 0, 0, (0, 0), (8, 8)), metalness_factor=0.0, roughness_factor=0.6))
    s.add_mesh(meshgen.grid(32, 32, (-6, -4, 0), (12, 0, 0), (0, 12, 0)), None, m_floor)
    s.add_mesh(meshgen.grid(32, 16, (-6, 1.5, 0), (12, 0, 0), (0, 0, 7)), None, m_floor)
    s.add_light(abi.LIGHT_POINT, position=(0, -4, 6), intensity=60.0)
    s.add_light(abi.LIGHT_DIRECTIONAL, direction=(0.2, 0.5, -0.8), intensity=2.0)
    s.env_image = sky_image(1024, 512, 2.0e3)
    s.world_to_view = camera.orbit_world_to_view((0, 0, 3.0), 5.2, 0.0, -0.05)
    s.width = s.height = size
    st = PtSettings.app_defaults()
    st.min_bounces, st.max_bounces = 2, 16
    s.settings = st
    s.bounce_limit = 16
    return s


# ---- config 5: skinned figure ---------------------------------------------------------------------------
JOINT_NAMES = ["hips", "spine", "chest", "neck", "head", "l_shoulder", "l_elbow", "l_hand", "r_shoulder", "r_elbow", "r_hand",
               "l_hip", "l_knee", "l_foot", "l_toe", "r_hip", "r_knee", "r_foot", "r_toe"]
JOINT_PARENT = [-1, 0, 1, 2, 3, 2, 5, 6, 2, 8, 9, 0, 11, 12, 13, 0, 15, 16, 17]
JOINT_REST = np.array([[0, 0, 1.0], [0, 0, 1.2], [0, 0, 1.45], [0, 0, 1.62], [0, 0, 1.75],
                       [0.2, 0, 1.5], [0.48, 0, 1.5], [0.74, 0, 1.5], [-0.2, 0, 1.5], [-0.48, 0, 1.5], [-0.74, 0, 1.5],
                       [0.11, 0, 0.95], [0.11, 0, 0.52], [0.11, 0, 0.08], [0.11, -0.16, 0.03],
                       [-0.11, 0, 0.95], [-0.11, 0, 0.52], [-0.11, 0, 0.08], [-0.11, -0.16, 0.03]], dtype=np.float64)


def _axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64); axis /= np.linalg.norm(axis)
    s = math.sin(angle / 2)
    return (axis[0] * s, axis[1] * s, axis[2] * s, math.cos(angle / 2))


def skinned_figure_pose(t):
    """Joint global matrices of the 2 s walk cycle at time t: LINEAR keyframes at 30 Hz, i.e. the pose
    is sampled at the two neighbouring keyframes and blended (Animation.cpp LINEAR / slerp path)."""
    def key(tk):
        ph = 2 * math.pi * tk / 2.0
        swing = 0.6 * math.sin(ph)
        rot = {i: (0, 0, 0, 1) for i in range(19)}
        rot[11] = _axis_angle((1, 0, 0), swing); rot[15] = _axis_angle((1, 0, 0), -swing)
        rot[12] = _axis_angle((1, 0, 0), -0.5 * max(0.0, math.sin(ph + 0.6))); rot[16] = _axis_angle((1, 0, 0), -0.5 * max(0.0, -math.sin(ph + 0.6)))
        rot[5] = _axis_angle((0, 1, 0), 1.1); rot[8] = _axis_angle((0, 1, 0), -1.1)
        rot[6] = _axis_angle((1, 0, 0), -0.5 * swing - 0.3); rot[9] = _axis_angle((1, 0, 0), 0.5 * swing - 0.3)
        rot[1] = _axis_angle((0, 0, 1), 0.08 * math.sin(ph))
        return rot, 0.03 * abs(math.sin(ph))
    k0 = math.floor(t * 30.0) / 30.0
    k1 = k0 + 1.0 / 30.0
    a = (t - k0) * 30.0
    r0, b0 = key(k0); r1, b1 = key(k1)
    glob = [None] * 19
    for j in range(19):
        q0, q1 = np.array(r0[j]), np.array(r1[j])
        if np.dot(q0, q1) < 0:
            q1 = -q1
        q = q0 * (1 - a) + q1 * a
        q /= np.linalg.norm(q)
        p = JOINT_PARENT[j]
        local_t = JOINT_REST[j] - (JOINT_REST[p] if p >= 0 else 0)
        if p < 0:
            local_t = local_t + np.array([0, 0, b0 * (1 - a) + b1 * a])
        local = camera.trs(local_t, q)
        glob[j] = local if p < 0 else glob[p] @ local
    return glob


def add_skinned_figure(s, transform=None, seed=5):
    """The CesiumMan-class figure as ONE dynamic instance of scene `s` (+ its skin record in s.skins): capsule-limbed body,
    ~4.7 k triangles / ~3.3 k vertices, 19 joints, 4 weights per vertex.  Returns the instance index."""
    bones = [(0, 1, 0.13), (1, 2, 0.15), (2, 3, 0.07), (3, 4, 0.11), (2, 5, 0.07), (5, 6, 0.055), (6, 7, 0.045), (2, 8, 0.07), (8, 9, 0.055),
             (9, 10, 0.045), (0, 11, 0.08), (11, 12, 0.075), (12, 13, 0.06), (13, 14, 0.045), (0, 15, 0.08), (15, 16, 0.075), (16, 17, 0.06), (17, 18, 0.045)]
    parts, joints, weights = [], [], []
    for a, b, r in bones:
        m, tf = meshgen.capsule_tube(JOINT_REST[a], JOINT_REST[b], r, nseg=10, nring=6)
        parts.append(m)
        pa = JOINT_PARENT[a] if JOINT_PARENT[a] >= 0 else a
        j = np.tile(np.array([a, b, pa, 0], dtype=np.uint16), (m.num_vertices, 1))
        w = np.stack([(1 - tf) * 0.85, tf * 0.85 + 0.05, np.full_like(tf, 0.10), np.zeros_like(tf)], axis=1)
        w /= w.sum(axis=1, keepdims=True)
        joints.append(j); weights.append(w)
    fig = meshgen.merge(parts)
    fig.joints, fig.weights = np.concatenate(joints), np.concatenate(weights)
    rng = np.random.default_rng(seed)
    h = value_noise(rng, 512, 4, 4)
    t_base = s.add_texture(rgba(0.5 + 0.4 * h, 0.3 + 0.3 * h, 0.25 + 0.2 * h), True)
    m_fig = s.add_material(material(albedo=PtTextureSample(t_base), metalness_factor=0.0, roughness_factor=0.6))
    inst = s.add_mesh(fig, transform, m_fig, dynamic=True)
    inv_bind = [np.linalg.inv(camera.translate(JOINT_REST[j])) for j in range(19)]
    d = s.instances[inst]
    jw = s.add_buffer(meshgen.pack_joint_weight(fig.joints, fig.weights), abi.FORMAT_JOINT_WEIGHT)
    s.skins.append({"instance": inst, "mesh": fig, "joint_weight": jw, "inverse_bind": inv_bind,
                    "input_position": d.gpu.position_descriptor, "input_tangent_space": d.gpu.tangent_space_descriptor})
    return inst


def skinned_figure(width=3840, height=2160, seed=5):
    """BASELINE config 5 stand-in for CesiumMan: capsule-limbed figure (~4.7 k triangles / ~3.3 k
    vertices), 19 joints, 4 weights per vertex, 2 s walk cycle, ground plane; 4K, 8 bounces."""
    s = SceneData("config5_skinned_figure")
    add_skinned_figure(s, None, seed)
    m_floor = s.add_material(material(base_color_factor=(0.5, 0.5, 0.5, 1), metalness_factor=0.0, roughness_factor=0.8))
    s.add_mesh(meshgen.grid(16, 16, (-4, -4, 0), (8, 0, 0), (0, 8, 0)), None, m_floor)
    s.add_light(abi.LIGHT_POINT, position=(1.5, -2.0, 3.0), intensity=30.0)
    s.add_light(abi.LIGHT_DIRECTIONAL, direction=(-0.3, 0.4, -0.85), intensity=2.5)
    s.env_image = sky_image(1024, 512, 5.0e3)
    s.world_to_view = camera.orbit_world_to_view((0, 0, 1.0), 2.6, 0.5, -0.2)
    s.width, s.height = width, height
    st = PtSettings.app_defaults()
    st.min_bounces, st.max_bounces = 2, 8
    s.settings = st
    s.bounce_limit = 8
    return s


def sponza_with_figure(width=1920, height=1080, tex=1024):
    """A Sponza-class static background (~257 k triangles) with ONE animated character standing in the atrium: the case the
    reference's per-frame flow is made for -- static BLASes kept, the dynamic BLAS refitted, the TLAS rebuilt (Pathtracer.cpp:138-257)."""
    s = sponza_class(width, height, tex)
    s.name = "sponza_class_with_skinned_figure"
    add_skinned_figure(s, camera.trs((-6.5, -0.4, 0.0), scale=(1.6, 1.6, 1.6)))
    return s


def bones_for_pose(skin, node_global, joint_globals):
    """Renderer::PerformSkinning bone matrices (Renderer.cpp:408-417):
    bones[i] = affineInverse(node.global) * joint.global * inverseBind[i]; inverse_transpose(mat3)."""
    out = []
    ninv = np.linalg.inv(node_global)
    for j, g in enumerate(joint_globals):
        T = ninv @ g @ skin["inverse_bind"][j]
        b = abi.PtBone()
        b.transform[:] = camera.cm(T)
        it = np.eye(4)
        it[:3, :3] = np.linalg.inv(T[:3, :3]).T
        b.inverse_transpose[:] = camera.cm(it)
        out.append(b)
    return out


MORPH_TARGET_KINDS = ("position+normal+tangent", "position", "position+normal", "normal+tangent", "position+normal+tangent", "position")


def add_morph_targets(s, skin_index=0, seed=11):
    """Six morph targets for the figure's primitive, as Gltf::CreateMorphTarget uploads them (Gltf.cpp:323-367): a float3
    position-delta stream when the target has POSITION, and -- when it has NORMAL -- a 10-10-10-2 stream made by
    EncodeTangentSpace(normal delta, tangent delta) (with TANGENT) or EncodeNormal(normal delta) (without).  One of each kind
    (MORPH_TARGET_KINDS), so that Skin.cs.hlsl:70-88's two descriptor tests are taken both ways.  Records go to
    s.skins[i]["targets"] = [{"position": buffer or -1, "tangent_space": buffer or -1, "sources": {...}}]."""
    sk = s.skins[skin_index]
    mesh = sk["mesh"]
    nv = mesh.num_vertices
    rng = np.random.default_rng(seed)
    P = mesh.positions.astype(np.float64)
    targets = []
    for k, kind in enumerate(MORPH_TARGET_KINDS):
        src, rec = {}, {"position": -1, "tangent_space": -1}
        if "position" in kind:
            ph = rng.uniform(0, 2 * math.pi, 3)
            amp = 0.04 + 0.02 * k
            dp = amp * np.stack([np.sin(7.0 * P[:, 2] + ph[0]), np.cos(5.0 * P[:, 0] + ph[1]), np.sin(9.0 * P[:, 1] + 3.0 * P[:, 2] + ph[2])], axis=1)
            src["POSITION"] = dp.astype(np.float32)
            rec["position"] = s.add_buffer(src["POSITION"], abi.FORMAT_R32G32B32_FLOAT)
        if "normal" in kind:
            dn = rng.normal(0, 0.3, (nv, 3)) + np.array([0.2, -0.1, 0.3])
            src["NORMAL"] = dn.astype(np.float32)
            if "tangent" in kind:
                dt = rng.normal(0, 0.3, (nv, 3)) + np.array([-0.3, 0.2, 0.1])
                # a glTF morph TANGENT is a VEC3 delta; upstream iterates it as 4 floats, the missing w filled with 1 (TinyGltfTools.h:217-219)
                src["TANGENT"] = np.concatenate([dt, np.ones((nv, 1))], axis=1).astype(np.float32)
                packed = meshgen.encode_tangent_space(src["NORMAL"], src["TANGENT"])
            else:
                packed = meshgen.encode_normal(src["NORMAL"])
            rec["tangent_space"] = s.add_buffer(packed, abi.FORMAT_R10G10B10A2_UNORM)
        rec["sources"] = src
        targets.append(rec)
    sk["targets"] = targets
    return targets


def pick_morph_targets(current_weights, max_targets=abi.MAX_SIMULTANEOUS_MORPH_TARGETS):
    """Renderer::PerformSkinning's choice (Renderer.cpp:425-443): walk the node's weights in order, keep those > 0; once four are
    held, a later weight replaces the FIRST smallest held one if it is larger.  Returns [(target index, weight)] in slot order."""
    w, which = [], []
    for j, cw in enumerate(current_weights):
        cw = float(np.float32(cw))
        if not cw > 0.0:
            continue
        if len(w) < max_targets:
            w.append(cw); which.append(j)
        else:
            mi = int(np.argmin(w))                       # std::min_element: the first of equal minima
            if w[mi] < cw:
                w[mi] = cw; which[mi] = j
    return list(zip(which, w))


class SkinBinding:
    """The per-frame dynamic-mesh step of Renderer::DrawFrame for one skinned instance (Renderer.cpp:399-457,
    Pathtracer.cpp:235-240): output streams, GpuSkin::Run parameters, and the instance table re-pointed at them.
    morph = [(target index, weight)] (<= 4, e.g. from pick_morph_targets) binds morph targets made by add_morph_targets."""

    def __init__(self, backend, scene, handles, skin_index=0, use_mfma=1, morph=None):
        sk = scene.skins[skin_index]
        mesh = sk["mesh"]
        self.backend, self.skin = backend, sk
        self.out_position = backend.buffer_create(None, abi.FORMAT_R32G32B32_FLOAT, mesh.num_vertices * 12)
        self.out_tangent_space = backend.buffer_create(None, abi.FORMAT_R10G10B10A2_UNORM, mesh.num_vertices * 4)
        p = abi.PtSkinParams()
        p.num_of_vertices = mesh.num_vertices
        p.input_mesh_flags = abi.MESH_FLAG_INDEX | abi.MESH_FLAG_TANGENT_SPACE | abi.MESH_FLAG_TEXCOORD_0 | abi.MESH_FLAG_JOINT_WEIGHT
        p.output_mesh_flags = abi.DYNAMIC_MESH_FLAG_POSITION | abi.DYNAMIC_MESH_FLAG_TANGENT_SPACE
        p.input_position = handles["buffers"][sk["input_position"]]
        p.input_tangent_space = handles["buffers"][sk["input_tangent_space"]]
        p.input_joint_weight = handles["buffers"][sk["joint_weight"]]
        p.output_position, p.output_tangent_space = self.out_position, self.out_tangent_space
        p.num_of_morph_targets = 0
        for i in range(4):
            p.morph_position[i] = -1
            p.morph_tangent_space[i] = -1
        for i, (ti, w) in enumerate(morph or []):
            t = sk["targets"][ti]
            p.morph_weights[i] = w
            p.morph_position[i] = handles["buffers"][t["position"]] if t["position"] != -1 else -1
            p.morph_tangent_space[i] = handles["buffers"][t["tangent_space"]] if t["tangent_space"] != -1 else -1
            p.num_of_morph_targets = i + 1
        p.use_mfma = int(use_mfma)
        self.params = p
        inst = [abi.PtInstanceDesc.from_buffer_copy(bytes(d)) for d in handles["instances"]]
        inst[sk["instance"]].gpu.position_descriptor = self.out_position
        inst[sk["instance"]].gpu.tangent_space_descriptor = self.out_tangent_space
        self.instances = inst
        backend.set_instances(inst)

    def pose(self, t, bones=True):
        """Skin the mesh to the walk-cycle pose at time t (seconds); the caller rebuilds the acceleration structure.
        bones=False: a morphed, unskinned node -- GpuSkin::Run is handed no bone buffer (Renderer.cpp:449, quirk q19)."""
        self.backend.skin_run(self.params, bones_for_pose(self.skin, np.eye(4), skinned_figure_pose(t)) if bones else None)
