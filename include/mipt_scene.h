/* mipt_scene.h -- C-ABI of the scene side of libmipt.so: everything upstream of the
 * path-tracing hot path that turns files into the exact data model include/mipt.h consumes.
 *
 *   gs_*  mirrors class Gltf            (Source/Gltf.h:16-232, Source/Gltf.cpp:105-1077, Source/TinyGltfTools.h:45-389)
 *         + Animation / AnimationPlayer (Source/Animation.cpp:9-123, Source/AnimationPlayer.cpp:3-22)
 *         + the per-frame host walk of Renderer::DrawFrame: PerformSkinning, GatherLights, GatherMaterials
 *           (Source/Renderer.cpp:399-500) and Pathtracer::BuildTlas' instance table (Source/Pathtracer.cpp:185-257)
 *   img_* mirrors the image loaders: LoadEnvironmentMapImageHdr / LoadEnvironmentMapImageExr
 *         (Source/EnvironmentMap.cpp:148-289) and tinygltf's image callback (RGBA8).
 *
 * Conventions as in mipt.h: int status (PT_OK = 0), no exceptions across the boundary, plain pointers and sizes.
 * Pointers returned by gs_get_* stay valid until gs_free; the scene is single-threaded like the reference's.
 */
#ifndef MIPT_SCENE_H
#define MIPT_SCENE_H

#include "mipt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gs_scene gs_scene;

/* Gltf::LoadFromGltf (Gltf.cpp:883-947): ".glb" -> binary container, ".gltf" -> JSON + external / data-URI buffers and images.
 * Fails (like the reference) on an extensionsRequired entry outside the whitelist of Gltf.cpp:920-933. */
int gs_load_file(const char* path, gs_scene** out);
void gs_free(gs_scene* s);
const char* gs_last_error(void);                       /* message of the last failed gs_ / img_ call on this thread */

typedef struct gs_counts {
    int meshes, primitives, materials /* incl. the default material 0 */, nodes, scenes, skins, animations, lights,
        textures /* = images */, samplers, cameras, dynamic_meshes;
} gs_counts;
int gs_get_counts(const gs_scene* s, gs_counts* out);

/* Mesh streams exactly as Mesh.cpp:124-132 lays them out (formats = pt_format). flat index = primitives in mesh order. */
typedef struct gs_primitive_info {
    int mesh, index_in_mesh;
    int flags;                 /* PT_MESH_FLAG_* */
    int topology;              /* glTF mode (4 = TRIANGLES) */
    int num_vertices, num_indices;
    int index_format;          /* PT_FORMAT_R16_UINT / PT_FORMAT_R32_UINT, 0 if not indexed */
    int material_id;           /* glTF material + 1; 0 = default material (Gltf.cpp:311-312) */
    int num_targets;
    const void* index;
    const float* position;             /* 3 x f32 */
    const uint32_t* tangent_space;     /* 10-10-10-2 (Gltf.cpp:65-104) */
    const float* texcoord[2];          /* 2 x f32 */
    const uint16_t* color;             /* 4 x unorm16 */
    const void* joint_weight;          /* 16 B: 4 x u16 joints, 4 x unorm16 weights */
} gs_primitive_info;
int gs_get_primitive(const gs_scene* s, int flat_index, gs_primitive_info* out);
int gs_get_morph_target(const gs_scene* s, int flat_index, int target, int* flags_out, const float** position_out, const uint32_t** tangent_space_out);

/* GpuMaterial(Gltf::Material) (Renderer.h:125-170).  Before gs_upload, texture `descriptor` = image index (or -1) and
 * `sampler` = glTF sampler index + 1 (0 = the default sampler); after gs_upload both are the context's handles. */
int gs_get_material(const gs_scene* s, int material, pt_material* out);
/* image `i`: decoded lazily on first material reference, with the sRGB flag of that first reference (Gltf.cpp:404-418). */
int gs_get_texture(const gs_scene* s, int i, int* width, int* height, int* srgb, int* loaded, const uint8_t** rgba8);
int gs_get_sampler(const gs_scene* s, int i, pt_sampler_desc* out);

/* Gltf::LoadCameras (Source/Gltf.cpp:642-655) -> class Camera (Source/Camera.h).  The fields are the file's (glTF 2.0 `perspective` /
 * `orthographic` objects); `view_to_clip` is Camera::GetViewToClip's reversed-Z matrix for them (Camera.h:80-92: perspectiveRH_ZO with near and
 * far swapped, far == 0 -> 100000; orthoRH_ZO(-1/xmag, 1/xmag, -1/ymag, 1/ymag, far, near)), column-major, ready for pt_execute_params.
 * Two things upstream does differently, neither observable there because the application never renders through a file's camera (it uses its
 * orbit / free controllers, Main.cpp:515): it compares the type with "Perspective" / "Orthographic" -- the specification's values are
 * lower-case, so a conformant file leaves its Camera objects unset (`upstream_type_matches` = 0) -- and it hands (aspectRatio, yfov, zfar, znear)
 * to Perspective(aspect, y_fov, z_near, z_far), i.e. near and far swapped.  Neither is reproduced: there is no behaviour to match. */
typedef struct gs_camera_info {
    int type;                          /* 0 perspective, 1 orthographic, -1 neither */
    float aspect_ratio, y_fov;         /* perspective (aspect_ratio 0 when the file leaves it to the viewport) */
    float x_mag, y_mag;                /* orthographic */
    float z_near, z_far;               /* z_far 0: infinite (perspective) */
    int upstream_type_matches;
    float view_to_clip[16];
} gs_camera_info;
int gs_get_camera(const gs_scene* s, int i, gs_camera_info* out);

typedef struct gs_node_info {
    int child, sibling, mesh, skin, dynamic_mesh, camera, light;
    float rest_translation[3], rest_rotation[4] /* x y z w */, rest_scale[3];
    float local_translation[3], local_rotation[4], local_scale[3];
    float global_transform[16];        /* column-major, valid after gs_calculate_global_transforms */
    int num_current_weights;
} gs_node_info;
int gs_get_node(const gs_scene* s, int node, gs_node_info* out);
int gs_get_node_weights(const gs_scene* s, int node, float* out, int capacity);      /* current_weights; returns the count */
int gs_get_scene_nodes(const gs_scene* s, int scene, int* out, int capacity);        /* root nodes; returns the count */
int gs_get_skin(const gs_scene* s, int skin, int* num_joints, const uint32_t** joints, const float** inverse_bind_poses);

typedef struct gs_channel_info {
    int node, path /* 0 translation, 1 rotation, 2 scale, 3 weights */, interpolation /* 0 STEP, 1 LINEAR, 2 CUBICSPLINE */;
    int format /* Animation.h:22-28 */, width, num_times, num_transform_bytes;
    const float* times;
    const uint8_t* transforms;
} gs_channel_info;
int gs_get_animation(const gs_scene* s, int animation, float* length, int* num_channels);
int gs_get_channel(const gs_scene* s, int animation, int channel, gs_channel_info* out);
/* Animation::Channel::GetTransform (Animation.cpp:73-122) of one channel at `time` into out[width].  CUBICSPLINE reproduces
 * the reference's behaviour (value and tangents all read from keyframe*3, its own TODO) unless fix_cubic_spline != 0. */
int gs_sample_channel(const gs_scene* s, int animation, int channel, float time, int fix_cubic_spline, float* out);

/* Gltf::ApplyRestTransforms / Animate / CalculateGlobalTransforms (Gltf.cpp:976-1041). */
int gs_apply_rest_transforms(gs_scene* s);
int gs_animate(gs_scene* s, int animation, float time);
int gs_calculate_global_transforms(gs_scene* s, int scene);

/* AnimationPlayer (AnimationPlayer.h, AnimationPlayer.cpp:3-22). */
typedef struct gs_player { int animation; float playhead; int playing; int loop; } gs_player;
int gs_player_tick(gs_scene* s, gs_player* player, float delta_time);

/* Renderer::GatherLights (Renderer.cpp:459-492): scene traversal order.  Returns the count (<= capacity) or < 0. */
int gs_gather_lights(const gs_scene* s, int scene, pt_light* out, int capacity);
/* Renderer::PerformSkinning's bone matrices for one skinned node (Renderer.cpp:408-417).  Returns the count. */
int gs_gather_bones(const gs_scene* s, int node, pt_bone* out, int capacity);

/* Create every stream, texture, sampler and dynamic-mesh output of the scene in the path-tracing context. */
int gs_upload(gs_scene* s, pt_ctx* ctx);
/* Gltf::Unload (Gltf.cpp:123-157; Main.cpp:43-54 calls it before the next scene is loaded): empties the context's instance and
 * material tables and destroys every stream, dynamic-mesh output and texture gs_upload created.  The scene can be uploaded again. */
int gs_unload(gs_scene* s, pt_ctx* ctx);
/* One frame of host work (Renderer.cpp:293-330): PerformSkinning (pt_skin_run per dynamic primitive), GatherLights ->
 * pt_scene_set_lights, GatherMaterials -> pt_scene_set_materials, BuildTlas' instance walk -> pt_scene_set_instances.
 * Global transforms must be current.  light_count_out feeds pt_execute_params.light_count. */
int gs_frame(gs_scene* s, pt_ctx* ctx, int scene, int* light_count_out);

/* ---- image files (N2 + the glTF image callback) */
int img_load_rgba8(const char* path, int* width, int* height, uint8_t** rgba8_out);            /* PNG / JPEG */
int img_decode_rgba8(const void* data, size_t bytes, int* width, int* height, uint8_t** rgba8_out);
/* LoadEnvironmentMapImageHdr / Exr: RGB32F, top row first; half_source_out = 1 for EXR HALF channels.
 * is_exr: 0 = Radiance .hdr, 1 = EXR with R, G, B channels (environment maps), 2 = EXR with exactly one HALF channel,
 * replicated into r, g, b (GpuResources::LoadLookupTables, GpuResources.cpp:72-132: Sheen_E.exr). */
int img_load_rgb32f(const char* path, int* width, int* height, int* half_source_out, float** rgb_out);
int img_decode_rgb32f(const void* data, size_t bytes, int is_exr, int* width, int* height, int* half_source_out, float** rgb_out);
void img_free(void* p);
/* Image writers for offline output (the reference only presents to a swap chain): 8-bit PNG of the tone-mapped
 * frame (pt_tonemap's RGBA8; channels = 3 drops alpha), and the linear RGB32F radiance as PFM or uncompressed scan-line
 * OpenEXR (half != 0: HALF channels). */
int img_write_png(const char* path, const uint8_t* rgba8, int width, int height, int channels);
int img_write_pfm(const char* path, const float* rgb32f, int width, int height);
int img_write_exr(const char* path, const float* rgb32f, int width, int height, int half);

#ifdef __cplusplus
}
#endif
#endif
