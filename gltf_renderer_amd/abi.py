"""ctypes mirror of include/mipt.h (the C-ABI data contract).

Every structure here is byte-identical to the POD in include/mipt.h, which in turn is byte-identical
to what the reference uploads to its GPU:
  PtSettings      <- Pathtracer::Settings        Source/Pathtracer.h:70-85
  PtLight         <- Renderer::GpuLight          Source/Renderer.h:53-68
  PtTextureSample <- Renderer::TextureSample     Source/Renderer.h:70-86
  PtMaterial      <- Renderer::GpuMaterial       Source/Renderer.h:88-171
  PtMeshInstance  <- Pathtracer::GpuMeshInstance Source/Pathtracer.h:131-140
  PtBone          <- GpuSkin::Bone               Source/GpuSkin.h:12-15
Sizes are asserted at import.
"""
import ctypes as C

# ---- enums -----------------------------------------------------------------------------------
(DEBUG_OUTPUT_NONE, DEBUG_OUTPUT_HIT_KIND, DEBUG_OUTPUT_VERTEX_COLOR, DEBUG_OUTPUT_VERTEX_ALPHA,
 DEBUG_OUTPUT_VERTEX_NORMAL, DEBUG_OUTPUT_VERTEX_TANGENT, DEBUG_OUTPUT_VERTEX_BITANGENT,
 DEBUG_OUTPUT_TEXCOORD_0, DEBUG_OUTPUT_TEXCOORD_1, DEBUG_OUTPUT_COLOR, DEBUG_OUTPUT_ALPHA,
 DEBUG_OUTPUT_SHADING_NORMAL, DEBUG_OUTPUT_SHADING_TANGENT, DEBUG_OUTPUT_SHADING_BITANGENT,
 DEBUG_OUTPUT_METALNESS, DEBUG_OUTPUT_ROUGHNESS, DEBUG_OUTPUT_SPECULAR, DEBUG_OUTPUT_SPECULAR_COLOR,
 DEBUG_OUTPUT_CLEARCOAT, DEBUG_OUTPUT_CLEARCOAT_ROUGHNESS, DEBUG_OUTPUT_CLEARCOAT_NORMAL,
 DEBUG_OUTPUT_TRANSMISSIVE, DEBUG_OUTPUT_BOUNCE_DIRECTION, DEBUG_OUTPUT_BOUNCE_BSDF,
 DEBUG_OUTPUT_BOUNCE_PDF, DEBUG_OUTPUT_BOUNCE_WEIGHT, DEBUG_BOUNCE_IS_TRANSMISSION,
 DEBUG_OUTPUT_HEMISPHERE_VIEW_SIDE, DEBUG_OUTPUT_COUNT) = range(29)

DEBUG_OUTPUT_NAMES = [
    "none", "hit_kind", "vertex_color", "vertex_alpha", "vertex_normal", "vertex_tangent",
    "vertex_bitangent", "texcoord_0", "texcoord_1", "color", "alpha", "shading_normal",
    "shading_tangent", "shading_bitangent", "metalness", "roughness", "specular", "specular_color",
    "clearcoat", "clearcoat_roughness", "clearcoat_normal", "transmissive", "bounce_direction",
    "bounce_bsdf", "bounce_pdf", "bounce_weight", "bounce_is_transmission", "hemisphere_view_side"]

FLAG_NONE = 1 << 0
FLAG_CULL_BACKFACE = 1 << 1
FLAG_ACCUMULATE = 1 << 2
FLAG_LUMINANCE_CLAMP = 1 << 3
FLAG_INDIRECT_ENVIRONMENT_ONLY = 1 << 4
FLAG_POINT_LIGHTS = 1 << 5
FLAG_SHADOW_RAYS = 1 << 6
FLAG_ALPHA_SHADOWS = 1 << 7
FLAG_ENVIRONMENT_MAP = 1 << 8
FLAG_ENVIRONMENT_MIS = 1 << 9
FLAG_MATERIAL_DIFFUSE_WHITE = 1 << 10
FLAG_MATERIAL_USE_GEOMETRIC_NORMALS = 1 << 11
FLAG_MATERIAL_MIS = 1 << 12
FLAG_SHOW_NAN = 1 << 13
FLAG_SHOW_INF = 1 << 14
FLAG_SHADING_NORMAL_ADAPTATION = 1 << 15
# the application's default path-tracer flags (Source/Main.cpp:462-469)
APP_DEFAULT_FLAGS = (FLAG_ACCUMULATE | FLAG_POINT_LIGHTS | FLAG_SHADOW_RAYS | FLAG_ENVIRONMENT_MAP |
                     FLAG_ENVIRONMENT_MIS | FLAG_MATERIAL_MIS | FLAG_SHADING_NORMAL_ADAPTATION)

REFERENCE_MAX_BOUNCES = 5
MAX_TLAS_INSTANCES = 1000
TILE = 16

LIGHT_POINT, LIGHT_SPOT, LIGHT_DIRECTIONAL = 0, 1, 2
MATERIAL_FLAG_DOUBLE_SIDED = 1
ALPHA_MODE_OPAQUE, ALPHA_MODE_MASK, ALPHA_MODE_BLEND = 0, 1, 2
INSTANCE_FLAG_NONE, INSTANCE_FLAG_TRIANGLE_CULL_DISABLE, INSTANCE_FLAG_FORCE_NON_OPAQUE = 0, 0x1, 0x8
MASK_NONE, MASK_ALPHA_BLEND = 1, 2
(FORMAT_R16_UINT, FORMAT_R32_UINT, FORMAT_R32G32B32_FLOAT, FORMAT_R10G10B10A2_UNORM,
 FORMAT_R32G32_FLOAT, FORMAT_R16G16B16A16_UNORM, FORMAT_JOINT_WEIGHT) = range(1, 8)
ADDRESS_WRAP, ADDRESS_MIRROR, ADDRESS_CLAMP = 0, 1, 2
FILTER_POINT, FILTER_LINEAR = 0, 1
MESH_FLAG_INDEX, MESH_FLAG_TANGENT_SPACE, MESH_FLAG_TEXCOORD_0 = 1, 2, 4
MESH_FLAG_TEXCOORD_1, MESH_FLAG_COLOR, MESH_FLAG_JOINT_WEIGHT = 8, 16, 32
DYNAMIC_MESH_FLAG_POSITION, DYNAMIC_MESH_FLAG_TANGENT_SPACE = 1, 2
MAX_SIMULTANEOUS_MORPH_TARGETS = 4          # Source/Config.h:21
TONEMAPPER_NONE, TONEMAPPER_AGX = 0, 1
MODE_WAVEFRONT, MODE_MEGAKERNEL = 0, 1


class PtSettings(C.Structure):
    _fields_ = [("min_bounces", C.c_int32), ("max_bounces", C.c_int32),
                ("reset", C.c_uint8), ("_pad0", C.c_uint8 * 3),
                ("debug_output", C.c_int32), ("flags", C.c_uint32),
                ("environment_color", C.c_float * 3), ("environment_intensity", C.c_float),
                ("use_frame_as_seed", C.c_uint8), ("_pad1", C.c_uint8 * 3), ("seed", C.c_uint32),
                ("luminance_clamp", C.c_float),
                ("min_russian_roulette_continue_prob", C.c_float),
                ("max_russian_roulette_continue_prob", C.c_float),
                ("max_accumulated_frames", C.c_int32), ("max_ray_length", C.c_float)]

    @classmethod
    def defaults(cls):
        """Pathtracer::Settings in-class defaults (Source/Pathtracer.h:71-84); environment_color
        value-initialised to 0 as the app does (quirk q28)."""
        s = cls()
        s.min_bounces, s.max_bounces = 2, 2
        s.reset = 0
        s.debug_output = DEBUG_OUTPUT_NONE
        s.flags = FLAG_ACCUMULATE | FLAG_POINT_LIGHTS | FLAG_ENVIRONMENT_MAP
        s.environment_intensity = 1.0
        s.use_frame_as_seed = 1
        s.seed = 0
        s.luminance_clamp = 1000.0
        s.min_russian_roulette_continue_prob = 0.1
        s.max_russian_roulette_continue_prob = 0.9
        s.max_accumulated_frames = 65536
        s.max_ray_length = 1000.0
        return s

    @classmethod
    def app_defaults(cls):
        """What Main.cpp sets before the first frame (Source/Main.cpp:462-474)."""
        s = cls.defaults()
        s.flags = APP_DEFAULT_FLAGS
        s.luminance_clamp = 20.0
        s.max_accumulated_frames = 8196
        return s


class PtLight(C.Structure):
    _fields_ = [("type", C.c_int32), ("position", C.c_float * 3), ("cutoff", C.c_float),
                ("direction", C.c_float * 3), ("intensity", C.c_float), ("color", C.c_float * 3),
                ("inner_angle", C.c_float), ("outer_angle", C.c_float), ("pad", C.c_uint8 * 8)]


class PtTextureSample(C.Structure):
    _fields_ = [("descriptor", C.c_int32), ("sampler", C.c_int32), ("tex_coord", C.c_int32),
                ("rotation", C.c_float), ("offset", C.c_float * 2), ("scale", C.c_float * 2)]

    def __init__(self, descriptor=-1, sampler=0, tex_coord=0, rotation=0.0, offset=(0.0, 0.0), scale=(1.0, 1.0)):
        super().__init__()
        self.descriptor, self.sampler, self.tex_coord, self.rotation = descriptor, sampler, tex_coord, rotation
        self.offset[:] = offset
        self.scale[:] = scale


_TS = PtTextureSample


class PtMaterial(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("alpha_mode", C.c_int32), ("metalness_factor", C.c_float),
                ("roughness_factor", C.c_float), ("base_color_factor", C.c_float * 4),
                ("occlusion_factor", C.c_float), ("emissive_factor", C.c_float * 3),
                ("alpha_cutoff", C.c_float), ("ior", C.c_float), ("normal_scale", C.c_float), ("pad_0", C.c_float),
                ("normal", _TS), ("albedo", _TS), ("metallic_roughness", _TS), ("occlusion", _TS), ("emissive", _TS),
                ("specular_factor", C.c_float), ("specular_color_factor", C.c_float * 3),
                ("specular", _TS), ("specular_color", _TS),
                ("clearcoat_factor", C.c_float), ("clearcoat_roughness_factor", C.c_float),
                ("clearcoat_normal_scale", C.c_float), ("pad_1", C.c_float),
                ("clearcoat", _TS), ("clearcoat_roughness", _TS), ("clearcoat_normal", _TS),
                ("anisotropy_strength", C.c_float), ("anisotropy_rotation", C.c_float), ("pad_2", C.c_float * 2),
                ("anisotropy", _TS),
                ("sheen_color_factor", C.c_float * 3), ("sheen_roughness_factor", C.c_float),
                ("sheen_color", _TS), ("sheen_roughness", _TS),
                ("transmission_factor", C.c_float), ("thickness_factor", C.c_float), ("pad_3", C.c_float * 2),
                ("transmission", _TS),
                ("attenuation_distance", C.c_float), ("attenuation_color", C.c_float * 3),
                ("thickness", _TS)]

    TEXTURE_SLOTS = ["normal", "albedo", "metallic_roughness", "occlusion", "emissive", "specular",
                     "specular_color", "clearcoat", "clearcoat_roughness", "clearcoat_normal", "anisotropy",
                     "sheen_color", "sheen_roughness", "transmission", "thickness"]

    @classmethod
    def default(cls):
        """GpuMaterial(Gltf::Material{}) - the default material at index 0 (Source/Gltf.cpp:470-475,
        defaults Source/Gltf.h:109-175)."""
        m = cls()
        m.flags = 0
        m.alpha_mode = ALPHA_MODE_OPAQUE
        m.metalness_factor = 1.0
        m.roughness_factor = 1.0
        m.base_color_factor[:] = (1, 1, 1, 1)
        m.occlusion_factor = 1.0
        m.emissive_factor[:] = (0, 0, 0)
        m.alpha_cutoff = 0.0
        m.ior = 1.5
        m.normal_scale = 1.0
        m.specular_factor = 1.0
        m.specular_color_factor[:] = (1, 1, 1)
        m.clearcoat_factor = 0.0
        m.clearcoat_roughness_factor = 0.0
        m.clearcoat_normal_scale = 1.0
        m.anisotropy_strength = 0.0
        m.anisotropy_rotation = 0.0
        m.sheen_color_factor[:] = (0, 0, 0)
        m.sheen_roughness_factor = 0.0
        m.transmission_factor = 0.0
        m.thickness_factor = 0.0
        m.attenuation_distance = 0.0
        m.attenuation_color[:] = (1, 1, 1)
        for slot in cls.TEXTURE_SLOTS:
            setattr(m, slot, PtTextureSample())
        return m


class PtMeshInstance(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("normal_transform", C.c_float * 16),
                ("index_descriptor", C.c_int32), ("position_descriptor", C.c_int32),
                ("tangent_space_descriptor", C.c_int32), ("texcoord_descriptors", C.c_int32 * 2),
                ("color_descriptor", C.c_int32), ("material_id", C.c_int32)]


class PtInstanceDesc(C.Structure):
    _fields_ = [("gpu", PtMeshInstance), ("instance_mask", C.c_uint32), ("instance_flags", C.c_uint32),
                ("num_of_vertices", C.c_uint32), ("num_of_indices", C.c_uint32), ("dynamic", C.c_int32)]


class PtSamplerDesc(C.Structure):
    _fields_ = [("address_u", C.c_int32), ("address_v", C.c_int32), ("min_filter", C.c_int32), ("mag_filter", C.c_int32)]


class PtExecuteParams(C.Structure):
    _fields_ = [("world_to_view", C.c_float * 16), ("view_to_clip", C.c_float * 16),
                ("width", C.c_uint32), ("height", C.c_uint32), ("frame", C.c_uint64),
                ("light_count", C.c_int32), ("environment_map", C.c_int32), ("output", C.c_void_p),
                ("tile_rank", C.c_uint32), ("tile_rank_count", C.c_uint32)]


class PtBone(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("inverse_transpose", C.c_float * 16)]


class PtSkinParams(C.Structure):
    _fields_ = [("num_of_vertices", C.c_uint32), ("input_mesh_flags", C.c_uint32), ("output_mesh_flags", C.c_uint32),
                ("input_position", C.c_int32), ("input_tangent_space", C.c_int32), ("input_joint_weight", C.c_int32),
                ("output_position", C.c_int32), ("output_tangent_space", C.c_int32),
                ("num_of_morph_targets", C.c_int32), ("morph_weights", C.c_float * 4),
                ("morph_position", C.c_int32 * 4), ("morph_tangent_space", C.c_int32 * 4), ("use_mfma", C.c_int32)]


class PtTonemapConfig(C.Structure):
    _fields_ = [("tonemapper", C.c_int32), ("exposure", C.c_float), ("frame", C.c_int32), ("dither", C.c_int32)]

    @classmethod
    def default(cls):
        return cls(TONEMAPPER_AGX, 1.0, 0, 0)


class PtStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("rays_primary", C.c_uint64), ("rays_bounce", C.c_uint64),
                ("rays_shadow", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
                ("closest_hits", C.c_uint64), ("texture_taps", C.c_uint64),
                ("trace_ms", C.c_float), ("accel_ms", C.c_float), ("skin_ms", C.c_float),
                ("accumulated_frames", C.c_int32), ("bvh_nodes", C.c_uint32), ("bvh_triangles", C.c_uint32),
                ("nodes_visited_shadow", C.c_uint64), ("tris_tested_shadow", C.c_uint64), ("stage_ms", C.c_float * 5),
                ("bvh_stack_need", C.c_uint32), ("accel_builds", C.c_uint32), ("accel_refits", C.c_uint32),
                ("bvh_stack_capacity", C.c_uint32), ("accel_builder_fallbacks", C.c_uint32), ("deep_stack_pushes", C.c_uint64)]


STAGE_NAMES = ("generate", "trace", "shade", "shadow", "resolve")
EXCHANGE_GATHER, EXCHANGE_REDUCE = 0, 1
BUILDER_LBVH, BUILDER_PLOC, BUILDER_PLOC_REINSERT = 0, 1, 2
EXCHANGE_ID_BYTES = 128


assert C.sizeof(PtSettings) == 64
assert C.sizeof(PtLight) == 64
assert C.sizeof(PtTextureSample) == 32
assert C.sizeof(PtMaterial) == 640
assert C.sizeof(PtMeshInstance) == 156
assert C.sizeof(PtInstanceDesc) == 176
assert C.sizeof(PtExecuteParams) == 168
assert C.sizeof(PtBone) == 128
assert C.sizeof(PtStats) == 152
