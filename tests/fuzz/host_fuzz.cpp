// host_fuzz.cpp -- mutation fuzzer for the plain-C++ scene side (image decoders, glTF/GLB loader, animation sampler).
// Test infrastructure: built with -fsanitize=address,undefined by tests/test_host_fuzz.py (CPU only; the GPU entry points the
// loader would call on upload are stubbed out here).  Every input is a seed file with a few random byte edits / truncations;
// the decoders must return an error or a result, never touch memory they do not own.
//   usage: host_fuzz <seed-dir> <iterations> <rng-seed> [only seeds whose name contains this]
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <unistd.h>
#include <vector>

#include "../../include/mipt.h"
#include "../../include/mipt_scene.h"

// ---- stubs for the device-side C-ABI (never reached: the fuzzer does not upload)
extern "C" {
int pt_buffer_create(pt_ctx*, const void*, size_t, int, int*) { return -1; }
int pt_texture_create(pt_ctx*, const uint8_t*, int, int, int, int*) { return -1; }
int pt_sampler_create(pt_ctx*, const pt_sampler_desc*, int*) { return -1; }
int pt_scene_set_materials(pt_ctx*, const pt_material*, int) { return -1; }
int pt_scene_set_lights(pt_ctx*, const pt_light*, int) { return -1; }
int pt_scene_set_instances(pt_ctx*, const pt_instance_desc*, int) { return -1; }
int pt_skin_run(pt_ctx*, const pt_skin_params*, const pt_bone*, int) { return -1; }
const char* pt_last_error(const pt_ctx*) { return "stub"; }
int pt_buffer_destroy(pt_ctx*, int) { return -1; }
int pt_texture_destroy(pt_ctx*, int) { return -1; }
}

static uint64_t rng_state = 1;
static uint32_t rnd() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 33); }

static std::vector<uint8_t> read_file(const std::string& p) {
    std::vector<uint8_t> d;
    FILE* f = fopen(p.c_str(), "rb");
    if (!f) return d;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    d.resize(n > 0 ? (size_t)n : 0);
    if (n > 0 && fread(d.data(), 1, (size_t)n, f) != (size_t)n) d.clear();
    fclose(f);
    return d;
}

static void mutate(std::vector<uint8_t>& d) {
    if (d.empty()) return;
    const uint32_t kind = rnd() % 8;
    if (kind == 0) { d.resize(rnd() % d.size()); return; }                                   // truncate
    if (kind == 1 && d.size() > 16) { size_t a = rnd() % (d.size() - 8); memset(&d[a], rnd() & 1 ? 0xff : 0x00, 1 + rnd() % 8); return; }
    if (kind == 2 && d.size() > 16) {                                                        // splice a 32-bit extreme value
        static const uint32_t ext[] = {0u, 1u, 0x7fffffffu, 0x80000000u, 0xffffffffu, 0x00010000u, 0xfffffff0u};
        uint32_t v = ext[rnd() % 7]; memcpy(&d[rnd() % (d.size() - 4)], &v, 4); return;
    }
    const uint32_t edits = 1 + rnd() % 6;                                                    // byte flips
    for (uint32_t k = 0; k < edits; k++) d[rnd() % d.size()] ^= (uint8_t)(1u << (rnd() % 8)) | (uint8_t)(rnd() & (rnd() & 0xff));
}

// glTF JSON stays parseable if a number is overwritten by another number of the same length (also keeps GLB chunk sizes): this
// is what reaches the loader's index / count / offset validation instead of dying in the JSON parser.
static void mutate_json_number(std::vector<uint8_t>& d) {
    if (d.size() < 64) return;
    const size_t json_end = (d[0] == 'g' && d[1] == 'l' && d[2] == 'T' && d[3] == 'F') ? std::min(d.size(), (size_t)20 + (d[12] | d[13] << 8 | d[14] << 16)) : d.size();
    for (int attempt = 0; attempt < 64; attempt++) {
        size_t p = rnd() % json_end;
        while (p < json_end && !(d[p] >= '0' && d[p] <= '9')) p++;
        if (p >= json_end || p == 0) continue;
        const uint8_t before = d[p - 1];
        if (!(before == ':' || before == ',' || before == '[' || before == ' ' || before == '-')) continue;       // not inside a name / URI
        size_t e = p;
        while (e < json_end && d[e] >= '0' && d[e] <= '9') e++;
        const uint32_t how = rnd() % 4;
        for (size_t k = p; k < e; k++) d[k] = how == 0 ? '9' : (how == 1 ? '0' : (how == 2 ? (uint8_t)('0' + rnd() % 10) : (k == p ? (uint8_t)('1' + rnd() % 9) : d[k])));
        return;
    }
}

static bool ends_with(const std::string& s, const char* e) { size_t n = strlen(e); return s.size() >= n && s.compare(s.size() - n, n, e) == 0; }

static bool exercise_scene(const char* path) {
    gs_scene* sc = nullptr;
    if (gs_load_file(path, &sc) != 0 || !sc) return false;
    gs_counts c;
    if (gs_get_counts(sc, &c) == 0) {
        // touch what a caller would touch: the accessors, every channel sampled inside and outside its range, the hierarchy posed
        const int n_anim = c.animations < 8 ? c.animations : 8;
        for (int a = 0; a < n_anim; a++) {
            float length = 0; int channels = 0;
            if (gs_get_animation(sc, a, &length, &channels) != 0) continue;
            for (int ch = 0; ch < channels && ch < 32; ch++) {
                gs_channel_info ci;
                if (gs_get_channel(sc, a, ch, &ci) != 0) continue;
                if (ci.width < 0 || ci.width > (1 << 20)) continue;
                std::vector<float> out((size_t)ci.width + 8);
                for (float t : {-1.0f, 0.0f, 0.37f * length, length, 2.0f * length + 1.0f}) { gs_sample_channel(sc, a, ch, t, 0, out.data()); gs_sample_channel(sc, a, ch, t, 1, out.data()); }
            }
            gs_animate(sc, a, 0.37f * length);
        }
        gs_apply_rest_transforms(sc);
        gs_calculate_global_transforms(sc, 0);
        pt_light lights[16];
        gs_gather_lights(sc, 0, lights, 16);
        for (int m = 0; m < c.materials && m < 64; m++) { pt_material mat; gs_get_material(sc, m, &mat); }
        for (int p = 0; p < c.primitives && p < 256; p++) { gs_primitive_info pi; gs_get_primitive(sc, p, &pi); }
        for (int n = 0; n < c.nodes && n < 256; n++) { gs_node_info ni; gs_get_node(sc, n, &ni); pt_bone bones[64]; gs_gather_bones(sc, n, bones, 64); }
    }
    gs_free(sc);
    return true;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: host_fuzz <seed-dir> <iterations> <rng-seed>\n"); return 2; }
    const std::string dir = argv[1];
    const long iters = atol(argv[2]);
    rng_state = strtoull(argv[3], nullptr, 10) * 2654435761ull + 1;
    std::vector<std::string> names;
    if (DIR* d = opendir(dir.c_str())) {
        while (dirent* e = readdir(d)) if (e->d_name[0] != '.' && (argc < 5 || strstr(e->d_name, argv[4]))) names.push_back(e->d_name);
        closedir(d);
    }
    if (names.empty()) { fprintf(stderr, "no seeds in %s\n", dir.c_str()); return 2; }
    std::vector<std::vector<uint8_t>> seeds;
    for (auto& n : names) seeds.push_back(read_file(dir + "/" + n));
    const std::string tmp_base = dir + "/.fuzz_case";
    long decoded = 0, rejected = 0, scenes_ok = 0, scenes_bad = 0;
    for (long it = 0; it < iters; it++) {
        const size_t s = rnd() % seeds.size();
        std::vector<uint8_t> d = seeds[s];
        const uint32_t rounds = it < (long)seeds.size() ? 0 : 1 + rnd() % 3;                  // first pass: the seeds unchanged
        const std::string& name = names[s];
        const bool is_gltf = ends_with(name, ".glb") || ends_with(name, ".gltf");
        for (uint32_t r = 0; r < rounds; r++) { if (is_gltf && rnd() % 4 != 0) mutate_json_number(d); else mutate(d); }
        int w = 0, h = 0, half = 0;
        if (ends_with(name, ".png") || ends_with(name, ".jpg")) {
            uint8_t* out = nullptr;
            if (img_decode_rgba8(d.data(), d.size(), &w, &h, &out) == 0 && out) { volatile uint8_t sink = out[(size_t)w * h * 4 - 1]; (void)sink; img_free(out); decoded++; } else rejected++;
        } else if (ends_with(name, ".hdr") || ends_with(name, ".exr")) {
            float* out = nullptr;
            const int is_exr = ends_with(name, ".exr") ? 1 + (int)(rnd() % 2) : 0;
            if (img_decode_rgb32f(d.data(), d.size(), is_exr, &w, &h, &half, &out) == 0 && out) { volatile float sink = out[0]; (void)sink; img_free(out); decoded++; } else rejected++;
        } else {                                                                             // .glb / .gltf: the loader reads files
            const std::string path = tmp_base + (ends_with(name, ".glb") ? ".glb" : ".gltf");
            FILE* f = fopen(path.c_str(), "wb");
            if (!f) continue;
            if (!d.empty()) fwrite(d.data(), 1, d.size(), f);
            fclose(f);
            if (exercise_scene(path.c_str())) { decoded++; scenes_ok++; } else { rejected++; scenes_bad++; }
            unlink(path.c_str());
        }
    }
    printf("fuzz: %ld cases, %ld decoded, %ld rejected (scenes: %ld loaded, %ld refused)\n", iters, decoded, rejected, scenes_ok, scenes_bad);
    return 0;
}
