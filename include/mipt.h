/*
 * mipt.h -- C-ABI of the MI355X-native glTF path tracer (libmipt.so).
 *
 * This is the drop-in boundary for ONE hot path of l-johnson-code/glTF-Renderer: the
 * Pathtracer.cpp render loop (+ GpuSkin, the acceleration structure and the environment map
 * prerequisites it consumes).  The reference has no FFI; its seam is two C++ classes called from
 * Renderer::DrawFrame.  Every entry point below names the reference interface it replaces
 * (file:line relative to the reference root).  D3D12 types cannot cross the boundary, so:
 *   - "descriptors" (ints into ResourceDescriptorHeap[]) become ints into a context-owned
 *     resource table (pt_buffer_create / pt_texture_create / pt_sampler_create);  -1 = absent,
 *     exactly as in GpuMeshInstance (Source/Pathtracer.h:131-140);
 *   - GPU virtual addresses of the per-frame material / light arrays become host pointers that
 *     pt_scene_set_materials / pt_scene_set_lights upload (Source/Renderer.cpp:459-500);
 *   - the output UAV becomes a caller-owned device pointer to W*H RGBA32F texels
 *     (Source/Renderer.cpp:384, Source/Pathtracer.h:98-99).
 * All structs are plain data, byte-identical to the layouts the reference uploads to the GPU
 * (SURVEY.md section 8(a) A1-A6), so scene data drops in unchanged.
 *
 * Conventions: every function returns PT_OK (0) or a negative pt_status; pt_last_error() gives
 * the message.  No exceptions or aborts cross the boundary.  One caller thread per context
 * (as the reference: Source/Renderer.cpp:215-227).  All device work is enqueued on the HIP stream
 * given to pt_create and is asynchronous unless stated; pt_readback / pt_tonemap / pt_get_stats
 * synchronise that stream.
 */
#ifndef MIPT_H
#define MIPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIPT_ABI_VERSION 2

typedef enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID_ARGUMENT = -1,
    PT_ERR_OUT_OF_MEMORY = -2,
    PT_ERR_DEVICE = -3,        /* a HIP call failed; message holds hipGetErrorString */
    PT_ERR_BAD_HANDLE = -4,
    PT_ERR_CAPACITY = -5,      /* > PT_MAX_TLAS_INSTANCES etc. (reference logs and skips) */
    PT_ERR_NOT_READY = -6
} pt_status;

/* ---- Pathtracer::DebugOutput (Source/Pathtracer.h:19-49) -------------------------------- */
enum {
    PT_DEBUG_OUTPUT_NONE = 0,
    PT_DEBUG_OUTPUT_HIT_KIND,
    PT_DEBUG_OUTPUT_VERTEX_COLOR,
    PT_DEBUG_OUTPUT_VERTEX_ALPHA,
    PT_DEBUG_OUTPUT_VERTEX_NORMAL,
    PT_DEBUG_OUTPUT_VERTEX_TANGENT,
    PT_DEBUG_OUTPUT_VERTEX_BITANGENT,
    PT_DEBUG_OUTPUT_TEXCOORD_0,
    PT_DEBUG_OUTPUT_TEXCOORD_1,
    PT_DEBUG_OUTPUT_COLOR,
    PT_DEBUG_OUTPUT_ALPHA,
    PT_DEBUG_OUTPUT_SHADING_NORMAL,
    PT_DEBUG_OUTPUT_SHADING_TANGENT,
    PT_DEBUG_OUTPUT_SHADING_BITANGENT,
    PT_DEBUG_OUTPUT_METALNESS,
    PT_DEBUG_OUTPUT_ROUGHNESS,
    PT_DEBUG_OUTPUT_SPECULAR,
    PT_DEBUG_OUTPUT_SPECULAR_COLOR,
    PT_DEBUG_OUTPUT_CLEARCOAT,
    PT_DEBUG_OUTPUT_CLEARCOAT_ROUGHNESS,
    PT_DEBUG_OUTPUT_CLEARCOAT_NORMAL,
    PT_DEBUG_OUTPUT_TRANSMISSIVE,
    PT_DEBUG_OUTPUT_BOUNCE_DIRECTION,
    PT_DEBUG_OUTPUT_BOUNCE_BSDF,
    PT_DEBUG_OUTPUT_BOUNCE_PDF,
    PT_DEBUG_OUTPUT_BOUNCE_WEIGHT,
    PT_DEBUG_BOUNCE_IS_TRANSMISSION,
    PT_DEBUG_OUTPUT_HEMISPHERE_VIEW_SIDE,
    PT_DEBUG_OUTPUT_COUNT
};

/* ---- Pathtracer::Flags (Source/Pathtracer.h:51-68). FLAG_NONE really is bit 0. ------------ */
enum {
    PT_FLAG_NONE                            = 1 << 0,
    PT_FLAG_CULL_BACKFACE                   = 1 << 1,
    PT_FLAG_ACCUMULATE                      = 1 << 2,
    PT_FLAG_LUMINANCE_CLAMP                 = 1 << 3,
    PT_FLAG_INDIRECT_ENVIRONMENT_ONLY       = 1 << 4,
    PT_FLAG_POINT_LIGHTS                    = 1 << 5,
    PT_FLAG_SHADOW_RAYS                     = 1 << 6,
    PT_FLAG_ALPHA_SHADOWS                   = 1 << 7,
    PT_FLAG_ENVIRONMENT_MAP                 = 1 << 8,
    PT_FLAG_ENVIRONMENT_MIS                 = 1 << 9,
    PT_FLAG_MATERIAL_DIFFUSE_WHITE          = 1 << 10,
    PT_FLAG_MATERIAL_USE_GEOMETRIC_NORMALS  = 1 << 11,
    PT_FLAG_MATERIAL_MIS                    = 1 << 12,
    PT_FLAG_SHOW_NAN                        = 1 << 13,
    PT_FLAG_SHOW_INF                        = 1 << 14,
    PT_FLAG_SHADING_NORMAL_ADAPTATION       = 1 << 15
};

/* Pathtracer::MAX_BOUNCES (Source/Pathtracer.h:102).  pt_trace clamps min/max bounces to
 * [0, bounce_limit]; bounce_limit defaults to this and is raised with pt_set_bounce_limit
 * (the iterative kernel has no recursion-depth limit; BASELINE.json configs use 8 and 16). */
#define PT_REFERENCE_MAX_BOUNCES 5
#define PT_MAX_TLAS_INSTANCES 1000        /* Config::MAX_TLAS_INSTANCES, Source/Config.h:24 */
#define PT_MAX_SIMULTANEOUS_MORPH_TARGETS 4 /* Source/Config.h:21 */

/* ---- Pathtracer::Settings (Source/Pathtracer.h:70-85), field for field, 64 bytes ---------- */
typedef struct pt_settings {
    int32_t  min_bounces;                        /* default 2 */
    int32_t  max_bounces;                        /* default 2 */
    uint8_t  reset;  uint8_t _pad0[3];           /* bool reset */
    int32_t  debug_output;                       /* PT_DEBUG_OUTPUT_* */
    uint32_t flags;                              /* PT_FLAG_* */
    float    environment_color[3];               /* no initialiser upstream (quirk q28) */
    float    environment_intensity;              /* default 1 */
    uint8_t  use_frame_as_seed; uint8_t _pad1[3];/* default true */
    uint32_t seed;
    float    luminance_clamp;                    /* default 1000 */
    float    min_russian_roulette_continue_prob; /* default 0.1 */
    float    max_russian_roulette_continue_prob; /* default 0.9 */
    int32_t  max_accumulated_frames;             /* default 65536 */
    float    max_ray_length;                     /* ignored, as upstream: the host sends 1000
                                                    (Source/Pathtracer.cpp:322) */
} pt_settings;

/* ---- Renderer::GpuLight (Source/Renderer.h:53-68) == Light (Shaders/Lights.hlsli:9-19), 64 B */
enum { PT_LIGHT_POINT = 0, PT_LIGHT_SPOT = 1, PT_LIGHT_DIRECTIONAL = 2 };
typedef struct pt_light {
    int32_t type;
    float   position[3];
    float   cutoff;
    float   direction[3];
    float   intensity;
    float   color[3];
    float   inner_angle;
    float   outer_angle;
    uint8_t pad[8];
} pt_light;

/* ---- Renderer::TextureSample (Source/Renderer.h:70-86) == TextureAddress
 *      (Shaders/Material.hlsli:14-21), 32 B --------------------------------------------------- */
typedef struct pt_texture_sample {
    int32_t descriptor;   /* texture handle from pt_texture_create, -1 = none */
    int32_t sampler;      /* sampler handle, 0 = default linear/wrap (GpuResources.cpp:47-59) */
    int32_t tex_coord;    /* 0 or 1 */
    float   rotation;
    float   offset[2];
    float   scale[2];
} pt_texture_sample;

/* ---- Renderer::GpuMaterial (Source/Renderer.h:88-171) == Material
 *      (Shaders/Material.hlsli:23-66), 640 B -------------------------------------------------- */
enum { PT_MATERIAL_FLAG_DOUBLE_SIDED = 1 << 0 };
enum { PT_ALPHA_MODE_OPAQUE = 0, PT_ALPHA_MODE_MASK = 1, PT_ALPHA_MODE_BLEND = 2 };
typedef struct pt_material {
    uint32_t flags;
    int32_t  alpha_mode;
    float    metalness_factor;
    float    roughness_factor;
    float    base_color_factor[4];
    float    occlusion_factor;
    float    emissive_factor[3];           /* already multiplied by emissive_strength */
    float    alpha_cutoff;                 /* 0 unless MASK (Renderer.h:145) */
    float    ior;
    float    normal_scale;
    float    pad_0;
    pt_texture_sample normal;
    pt_texture_sample albedo;
    pt_texture_sample metallic_roughness;
    pt_texture_sample occlusion;
    pt_texture_sample emissive;
    float    specular_factor;
    float    specular_color_factor[3];
    pt_texture_sample specular;
    pt_texture_sample specular_color;
    float    clearcoat_factor;
    float    clearcoat_roughness_factor;
    float    clearcoat_normal_scale;
    float    pad_1;
    pt_texture_sample clearcoat;
    pt_texture_sample clearcoat_roughness;
    pt_texture_sample clearcoat_normal;
    float    anisotropy_strength;
    float    anisotropy_rotation;
    float    pad_2[2];
    pt_texture_sample anisotropy;
    float    sheen_color_factor[3];
    float    sheen_roughness_factor;
    pt_texture_sample sheen_color;
    pt_texture_sample sheen_roughness;
    float    transmission_factor;
    float    thickness_factor;
    float    pad_3[2];
    pt_texture_sample transmission;
    float    attenuation_distance;
    float    attenuation_color[3];
    pt_texture_sample thickness;
} pt_material;

/* ---- Pathtracer::GpuMeshInstance (Source/Pathtracer.h:131-140) == Instance
 *      (Shaders/PathTracer.lib.hlsl:32-41), 156 B.  Matrices are glm column-major. ----------- */
typedef struct pt_mesh_instance {
    float   transform[16];
    float   normal_transform[16];        /* inverseTranspose(transform), Pathtracer.cpp:205 */
    int32_t index_descriptor;            /* buffer handles; -1 = absent */
    int32_t position_descriptor;
    int32_t tangent_space_descriptor;
    int32_t texcoord_descriptors[2];
    int32_t color_descriptor;
    int32_t material_id;
} pt_mesh_instance;

/* D3D12_RAYTRACING_INSTANCE_FLAG_* values the reference sets (Source/Pathtracer.cpp:216-222) */
enum {
    PT_INSTANCE_FLAG_NONE                  = 0,
    PT_INSTANCE_FLAG_TRIANGLE_CULL_DISABLE = 0x1,
    PT_INSTANCE_FLAG_FORCE_NON_OPAQUE      = 0x8
};
/* InstanceMask (Source/Pathtracer.cpp:191-194) */
enum { PT_MASK_NONE = 1 << 0, PT_MASK_ALPHA_BLEND = 1 << 1 };

/* One TLAS instance: what Pathtracer::BuildTlas hands to AddTlasInstance plus the table row
 * (Source/Pathtracer.cpp:185-257, Source/RayTracingAccelerationStructure.cpp:292-317). */
typedef struct pt_instance_desc {
    pt_mesh_instance gpu;
    uint32_t instance_mask;      /* PT_MASK_* */
    uint32_t instance_flags;     /* PT_INSTANCE_FLAG_* */
    uint32_t num_of_vertices;
    uint32_t num_of_indices;     /* = 3 * triangles; with index_descriptor -1: vertex count */
    int32_t  dynamic;            /* non-zero: positions are rewritten by pt_skin_run each frame
                                    (DynamicBlas, ALLOW_UPDATE) */
} pt_instance_desc;

/* ---- resource formats (Source/Mesh.cpp:124-132) ------------------------------------------- */
typedef enum pt_format {
    PT_FORMAT_R16_UINT = 1,            /* index */
    PT_FORMAT_R32_UINT = 2,            /* index */
    PT_FORMAT_R32G32B32_FLOAT = 3,     /* position, morph position */
    PT_FORMAT_R10G10B10A2_UNORM = 4,   /* tangent space */
    PT_FORMAT_R32G32_FLOAT = 5,        /* texcoord */
    PT_FORMAT_R16G16B16A16_UNORM = 6,  /* color */
    PT_FORMAT_JOINT_WEIGHT = 7         /* 16 B {u16 x4 joints, unorm16 x4 weights} */
} pt_format;

/* glTF sampler (Source/TinyGltfTools.h:16-43) */
enum { PT_ADDRESS_WRAP = 0, PT_ADDRESS_MIRROR = 1, PT_ADDRESS_CLAMP = 2 };
enum { PT_FILTER_POINT = 0, PT_FILTER_LINEAR = 1 };
typedef struct pt_sampler_desc {
    int32_t address_u, address_v;
    int32_t min_filter, mag_filter;   /* one mip only: mag filter decides (LOD 0) */
} pt_sampler_desc;

/* ---- Pathtracer::ExecuteParams (Source/Pathtracer.h:87-100) -------------------------------- */
typedef struct pt_execute_params {
    float    world_to_view[16];   /* Camera::GetWorldToView, glm column-major */
    float    view_to_clip[16];    /* Camera::GetViewToClip (reversed-Z, Camera.h:80-92) */
    uint32_t width, height;
    uint64_t frame;               /* renderer's global frame counter (Renderer.h:193) */
    int32_t  light_count;
    int32_t  environment_map;     /* handle from pt_env_create, -1 = none */
    void*    output;              /* device pointer, width*height float4, caller-owned */
    /* Multi-GPU pixel-tile sharding (new capability).  Tile t (PT_TILE x PT_TILE
     * pixels, row-major) is rendered iff t % tile_rank_count == tile_rank.  {0,1} = whole frame. */
    uint32_t tile_rank, tile_rank_count;
} pt_execute_params;
#define PT_TILE 16

/* ---- GpuSkin (Source/GpuSkin.h:17-19, Shaders/Skin.cs.hlsl) -------------------------------- */
typedef struct pt_bone {          /* GpuSkin::Bone, 128 B */
    float transform[16];
    float inverse_transpose[16];
} pt_bone;
enum {                             /* Mesh::Flags as the shader sees them (Skin.cs.hlsl:4-11) */
    PT_MESH_FLAG_INDEX = 1 << 0, PT_MESH_FLAG_TANGENT_SPACE = 1 << 1, PT_MESH_FLAG_TEXCOORD_0 = 1 << 2,
    PT_MESH_FLAG_TEXCOORD_1 = 1 << 3, PT_MESH_FLAG_COLOR = 1 << 4, PT_MESH_FLAG_JOINT_WEIGHT = 1 << 5
};
enum { PT_DYNAMIC_MESH_FLAG_POSITION = 1 << 0, PT_DYNAMIC_MESH_FLAG_TANGENT_SPACE = 1 << 1 };
typedef struct pt_skin_params {
    uint32_t num_of_vertices;
    uint32_t input_mesh_flags;         /* PT_MESH_FLAG_* */
    uint32_t output_mesh_flags;        /* PT_DYNAMIC_MESH_FLAG_* */
    int32_t  input_position;           /* buffer handles */
    int32_t  input_tangent_space;
    int32_t  input_joint_weight;
    int32_t  output_position;
    int32_t  output_tangent_space;
    int32_t  num_of_morph_targets;     /* clamped to 4 */
    float    morph_weights[PT_MAX_SIMULTANEOUS_MORPH_TARGETS];
    int32_t  morph_position[PT_MAX_SIMULTANEOUS_MORPH_TARGETS];       /* handles, -1 = absent */
    int32_t  morph_tangent_space[PT_MAX_SIMULTANEOUS_MORPH_TARGETS];
    int32_t  use_mfma;                 /* 0: per-vertex VALU blend; 1: v_mfma_f32_16x16x4_f32 blend */
} pt_skin_params;

/* ---- ToneMapper::Config (Source/ToneMapper.h:11-20) ---------------------------------------- */
enum { PT_TONEMAPPER_NONE = 0, PT_TONEMAPPER_AGX = 1 };
typedef struct pt_tonemap_config {
    int32_t tonemapper;   /* default AGX */
    float   exposure;     /* default 1 */
    int32_t frame;        /* dither seed; upstream never sets it (quirk q21) */
    int32_t dither;       /* 0 = off (parity metric is taken before dither) */
} pt_tonemap_config;

/* ---- counters ------------------------------------------------------------------------------ */
typedef struct pt_stats {
    uint64_t rays;              /* traversals started since pt_reset_stats (all kinds) */
    uint64_t rays_primary, rays_bounce, rays_shadow;
    uint64_t nodes_visited;     /* 128-B 4-wide BVH nodes fetched (0 unless counters enabled) */
    uint64_t tris_tested;       /* 48-B triangle packets fetched (0 unless counters enabled) */
    uint64_t closest_hits;      /* shading invocations */
    uint64_t texture_taps;      /* bilinear footprints fetched in shading */
    float    trace_ms;          /* hipEvent time of the last pt_trace's kernels */
    float    accel_ms;          /* last pt_build_accel */
    float    skin_ms;           /* last pt_skin_run */
    int32_t  accumulated_frames;
    uint32_t bvh_nodes, bvh_triangles;
    /* ABI 2 */
    uint64_t nodes_visited_shadow;   /* the occlusion stage's share of nodes_visited / tris_tested (wavefront mode, counters on) */
    uint64_t tris_tested_shadow;
    float    stage_ms[5];       /* last pt_trace with pt_enable_stage_timing(1): generate, closest-hit trace, shade, shadow trace,
                                   resolve -- summed over the bounces of the launch */
    uint32_t bvh_stack_need;    /* most traversal-stack entries any ray can hold in the current tree (build-time bound) */
    uint32_t accel_builds;      /* full builds / refits since pt_create */
    uint32_t accel_refits;
    uint32_t bvh_stack_capacity;       /* entries a ray's traversal stack holds for the current tree: 64 on chip, more (in memory) when
                                          bvh_stack_need asks for it -- a deep tree is never refused and never drops geometry */
    uint32_t accel_builder_fallbacks;  /* builds in which the clustering builder gave up (or made a tree too deep) and the radix
                                          tree over the same Morton order took over */
    uint64_t deep_stack_pushes;        /* stack entries rays wrote beyond the 64 on-chip ones since pt_reset_stats (0 for ordinary scenes) */
} pt_stats;

typedef struct pt_ctx pt_ctx;

/* Pathtracer::Init + GpuSkin::Create + EnvironmentMap::Init + GpuResources::LoadLookupTables
 * (Source/Pathtracer.h:104, GpuSkin.h:17, Renderer.cpp:161-170).  `device` = HIP device ordinal,
 * `hip_stream` = hipStream_t all work is enqueued on (NULL = the device's default stream).
 * `sheen_e_16x16` = the 256-float Sheen_E table (row = alpha, column = cos_theta). */
int pt_create(int device, void* hip_stream, const float* sheen_e_16x16, pt_ctx** out);
/* Pathtracer::Shutdown (Source/Pathtracer.h:106) */
void pt_destroy(pt_ctx* ctx);
const char* pt_last_error(const pt_ctx* ctx);
int pt_abi_version(void);

/* Vertex / index streams: Mesh::Create sub-allocations + descriptors (Source/Mesh.cpp:103-190).
 * host may be NULL (DynamicMesh outputs, Source/Mesh.cpp:236-290). */
int pt_buffer_create(pt_ctx* ctx, const void* host, size_t bytes, int format, int* handle_out);
int pt_buffer_update(pt_ctx* ctx, int handle, const void* host, size_t bytes);
int pt_buffer_read(pt_ctx* ctx, int handle, void* host, size_t bytes);
/* Gltf::Unload (Source/Gltf.cpp:123-157; called before the next scene loads, Source/Main.cpp:43-54): release one stream / texture.
 * Fails with PT_ERR_NOT_READY while the current instance table (buffers) or material table (textures) still refers to it --
 * replace those tables first, as the reference's per-frame tables are rebuilt from the new scene.  Handles are reused. */
int pt_buffer_destroy(pt_ctx* ctx, int handle);
int pt_texture_destroy(pt_ctx* ctx, int handle);
/* Gltf::LoadTexture (Source/Gltf.cpp:1047-1078): RGBA8, one mip, optional sRGB view. */
int pt_texture_create(pt_ctx* ctx, const uint8_t* rgba8, int width, int height, int srgb, int* handle_out);
/* Gltf::LoadSamplers (Source/Gltf.cpp:935, TinyGltfTools.h:16-43). Handle 0 pre-exists. */
int pt_sampler_create(pt_ctx* ctx, const pt_sampler_desc* desc, int* handle_out);

/* Renderer::GatherMaterials / GatherLights outputs (Source/Renderer.cpp:459-500) */
int pt_scene_set_materials(pt_ctx* ctx, const pt_material* materials, int count);
int pt_scene_set_lights(pt_ctx* ctx, const pt_light* lights, int count);
/* Pathtracer::BuildTlas instance list (Source/Pathtracer.cpp:185-257); marks the accel dirty. */
int pt_scene_set_instances(pt_ctx* ctx, const pt_instance_desc* instances, int count);

/* EnvironmentMap::CreateEnvironmentMap (Source/EnvironmentMap.cpp:84-130) minus the raster-only
 * GGX / diffuse cubes: equirect RGB32F -> RGBA16F cube (+mips) -> 1024^2 importance pyramid. */
int pt_env_create(pt_ctx* ctx, const float* equirect_rgb32f, int width, int height, int* env_out);
/* Test hook: copy out the preprocessed maps.  cube_rgba16f: 6*N*N*4 halfs of mip 0 (may be NULL);
 * importance: the whole pyramid, level 0 first (may be NULL).  Sizes via the out params. */
int pt_env_read(pt_ctx* ctx, int env, int* cube_size_out, uint16_t* cube_rgba16f, float* importance_pyramid);
/* EnvironmentMap::Destroy (the reference replaces its one environment in place, Source/EnvironmentMap.cpp:84-130). */
int pt_env_destroy(pt_ctx* ctx, int env);

/* BuildAllBlas / UpdateAllBlas / BuildTlas (Source/Pathtracer.cpp:138-257).  Called implicitly
 * by pt_trace when the scene is dirty; exposed so it can be timed on its own.
 *   - a new set of triangles (instance count, index counts) -> full build: Morton sort, LBVH, 4-wide collapse;
 *   - only vertices moved (pt_skin_run, pt_buffer_update) or instance rows changed (transform, flags) -> REFIT, like
 *     UpdateDynamicBlas (Source/RayTracingAccelerationStructure.cpp:110-158): topology and triangle order stay, the touched
 *     instances' packets are rewritten and every box re-derived.  Hits equal a full rebuild's; the tree's quality follows the
 *     pose it was built for, as upstream's refitted BLAS does (upstream never rebuilds one);
 *   - an instance table identical to the current one -> nothing.
 * pt_accel_request_rebuild forces the next call to build from scratch (e.g. after a pose has drifted far from the built one). */
int pt_build_accel(pt_ctx* ctx);
int pt_accel_request_rebuild(pt_ctx* ctx);
/* Which builder a full build uses (all on-device, all over the Morton order of the triangles' centroids):
 *   PT_BUILDER_LBVH  the radix tree of the sorted codes (Karras 2012): the fastest build (0.6 ms at 257 k triangles);
 *   PT_BUILDER_PLOC  parallel locally-ordered clustering (Meister & Bittner 2018) over the same order: neighbours
 *                    merge by joint surface area, bottom-up -- a better tree (7-35 % fewer node visits per ray on the BASELINE
 *                    scenes) for four times the build time (2.3 ms): the counterpart of the reference's PREFER_FAST_TRACE
 *                    static BLAS (RayTracingAccelerationStructure.cpp:228-290);
 *   PT_BUILDER_PLOC_REINSERT  (default) the PLOC tree improved by passes of parallel reinsertion (Meister & Bittner 2018): every
 *                    node looks for the place in the tree where it costs the least surface area and the moves that do not
 *                    touch each other are carried out, eight times over.
 * Same hits whichever builds.  Changing it makes the next pt_build_accel a full build.  The refit works on all.  The environment
 * variable MIPT_ACCEL_BUILDER = "lbvh" | "ploc" | "reinsert" sets the builder new contexts start with. */
enum { PT_BUILDER_LBVH = 0, PT_BUILDER_PLOC = 1, PT_BUILDER_PLOC_REINSERT = 2 };
int pt_set_accel_builder(pt_ctx* ctx, int builder);

/* GpuSkin::Run (Source/GpuSkin.cpp:57-118).  bones may be NULL (no skinning, quirk q19 kept). */
int pt_skin_run(pt_ctx* ctx, const pt_skin_params* params, const pt_bone* bones, int bone_count);

/* Pathtracer::PathtraceScene (Source/Pathtracer.cpp:259-367): one sample per pixel blended into
 * params->output with weight 1/(n+1); no-op once accumulated_frames >= max_accumulated_frames;
 * accumulation resets when world_to_clip changes or settings->reset. */
int pt_trace(pt_ctx* ctx, const pt_settings* settings, const pt_execute_params* params);
int pt_set_bounce_limit(pt_ctx* ctx, int limit);      /* default PT_REFERENCE_MAX_BOUNCES */
/* Sample batch: with samples > 1 one pt_trace stands for `samples` consecutive PathtraceScene calls (frames frame ..
 * frame + samples - 1, camera unchanged) carried by ONE set of kernel launches, so a small image or a small tile shard
 * still fills the GPU.  The output is bit-identical to issuing the calls one by one: sample k draws with the seed of frame
 * frame + k (or settings->seed when use_frame_as_seed is 0) and is blended with weight 1 / (accumulated_frames + k + 1)
 * in order; accumulated_frames advances by the batch and never passes max_accumulated_frames.  Without FLAG_ACCUMULATE,
 * or with a debug output, the batch is ignored: the call renders the one sample of `frame`, as always.  Default 1. */
#define PT_MAX_SAMPLES_PER_TRACE 64
int pt_set_samples_per_trace(pt_ctx* ctx, int samples);
/* Null shadow rays.  The reference traces every NEE shadow ray before it evaluates the BSDF (PathTracer.lib.hlsl:932, 948), also
 * when the sample then contributes nothing (light behind the surface, black texel, light out of range).  With culling enabled a
 * shadow ray whose weighted contribution is exactly (0,0,0) is not traced: the image is unchanged (T * 0 adds nothing), the ray
 * counts drop.  Default 0, so that rays-per-frame and Mrays/s mean what they mean for the reference. */
int pt_set_null_shadow_culling(pt_ctx* ctx, int enable);
int pt_enable_counters(pt_ctx* ctx, int enable);      /* node / triangle / tap counters (slower) */
int pt_enable_stage_timing(pt_ctx* ctx, int enable);  /* pt_stats.stage_ms: an event after every stage launch (diagnostic) */
/* Kernel arrangement (same per-vertex code, same results up to fp32 accumulation order): PT_MODE_WAVEFRONT (default) =
 * staged trace / shade / shadow kernels over SoA ray queues in HBM with ballot compaction; PT_MODE_MEGAKERNEL = one
 * lane per pixel-sample for the whole path.  stage_blocks: workgroups per wavefront stage launch (<= 0 keeps the current setting; the
 * initial setting sizes it by the launch: 1536 for a full 1080p frame, fewer for a small tile shard). */
enum { PT_MODE_WAVEFRONT = 0, PT_MODE_MEGAKERNEL = 1 };
int pt_set_kernel_mode(pt_ctx* ctx, int mode, int stage_blocks);
/* Counters accumulate over pt_trace calls since the last pt_reset_stats; the *_ms fields are the
 * hipEvent times of the most recent call of each kind. */
int pt_get_stats(pt_ctx* ctx, pt_stats* out);
int pt_reset_stats(pt_ctx* ctx);

/* Absent upstream (the reference only presents to a swapchain): offline output. */
int pt_readback(pt_ctx* ctx, const void* device_rgba32f, uint32_t width, uint32_t height, float* host_rgba32f);
/* ToneMapper::Run (Source/ToneMapper.cpp:60-91, Shaders/ToneMapper.ps.hlsl:83-101).  Writes
 * float RGB (pre-quantisation, what the parity metric uses) and/or RGBA8; either may be NULL. */
int pt_tonemap(pt_ctx* ctx, const pt_tonemap_config* config, const void* device_rgba32f,
               uint32_t width, uint32_t height, float* host_rgb32f, uint8_t* host_rgba8);

/* ---- multi-GPU: the one exchange per output frame (new capability; the reference is single-GPU) -------------
 * One process per GPU, scene replicated, rank r renders tiles t % world == r (pt_execute_params.tile_rank / tile_rank_count)
 * into ITS OWN full-size accumulation image; per reported frame pt_exchange_frame assembles the image on rank dst_rank with one
 * RCCL exchange on the context's stream (asynchronous).  RCCL is bound at run time (librccl.so.1); PT_ERR_NOT_READY if absent.
 *   PT_EXCHANGE_GATHER  own tiles packed (1/N of the image), grouped ncclSend / ncclRecv to the root over direct xGMI links,
 *                       unpacked into `frame`: bit-identical to a 1-rank frame.
 *   PT_EXCHANGE_REDUCE  ncclReduce(sum) of a zero-masked copy (own tiles, zeros elsewhere) into `frame`.
 * `local_image` is this rank's accumulation image and is only read, so accumulation composes with the exchange; `frame` is
 * written on the root only (NULL or == local_image: assemble in place over the other ranks' tiles, which the root never renders).
 * pt_exchange_unique_id: ncclGetUniqueId on one rank; the host hands the PT_EXCHANGE_ID_BYTES to every rank (file, socket, MPI,
 * torch.distributed ...).  world == 1 with unique_id NULL creates no communicator (the frame is the local image). */
#define PT_EXCHANGE_ID_BYTES 128
enum { PT_EXCHANGE_GATHER = 0, PT_EXCHANGE_REDUCE = 1 };
int pt_exchange_unique_id(void* id_out);
/* PT_OK if RCCL can be loaded in this process, PT_ERR_NOT_READY otherwise.  No GPU call, no communicator: a job lets every rank
 * probe and agree on the outcome BEFORE any rank enters pt_exchange_unique_id / pt_exchange_create (ncclCommInitRank is collective:
 * a rank that cannot load RCCL would leave the others waiting in it). */
int pt_exchange_probe(void);
int pt_exchange_create(pt_ctx* ctx, int rank, int world, const void* unique_id);
/* The same exchange for N contexts of ONE process (on one GPU or several): no RCCL, the transfers are stream-ordered
 * device-to-device copies between the contexts that joined the same `group` (any host-chosen number).  Transfers are posted, not
 * blocking: call pt_exchange_frame on the root AFTER the other ranks, every frame (PT_ERR_NOT_READY otherwise).  This is also how
 * the N > 1 logic of the exchange is tested on a one-GPU box. */
int pt_exchange_create_loopback(pt_ctx* ctx, int rank, int world, uint64_t group);
int pt_exchange_frame(pt_ctx* ctx, const void* local_image, void* frame, uint32_t width, uint32_t height, int mode, int dst_rank);
int pt_exchange_destroy(pt_ctx* ctx);
/* The transport-free halves of the gather, for hosts with their own transport (and the tests): a rank's tiles in slot order,
 * 256 float4 per 16x16 tile (pixels outside a ragged edge tile pack as zeros). */
size_t pt_tiles_packed_bytes(uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_rank_count);
int pt_tiles_pack(pt_ctx* ctx, const void* image, uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_rank_count, void* packed_device);
int pt_tiles_unpack(pt_ctx* ctx, const void* packed_device, uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_rank_count, void* image);

#ifdef __cplusplus
}
#endif
#endif /* MIPT_H */
