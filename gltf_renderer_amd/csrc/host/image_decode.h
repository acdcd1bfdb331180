// image_decode.h -- the file decoders the reference reaches through third-party libraries that are absent here:
//   PNG / JPEG -> RGBA8     (tinygltf -> stb_image, `Gltf.cpp:1047-1077` consumes `image.image` as 4 x 8 bit)
//   Radiance .hdr -> RGB32F (stb_image `stbi_loadf(.., 3)`, `EnvironmentMap.cpp:253-289`)
//   OpenEXR .exr -> RGB32F  (tinyexr scan-line images, `EnvironmentMap.cpp:148-251`, `GpuResources.cpp:82-98`)
// Written from the published formats (RFC 1950/1951 zlib+deflate, PNG 1.2, ITU T.81 JPEG, Radiance RGBE, OpenEXR 2 file
// layout).  Parity with stb/tinyexr is unpinned (neither library exists in this image): PNG, HDR and EXR are lossless
// formats with one correct answer; JPEG follows stb's integer IDCT / chroma filters / fixed-point colour transform as published.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace hostimg {

struct Image8 {            // RGBA8, row-major, top row first
    int width = 0, height = 0;
    int source_bits = 8;   // 16 when a 16-bit PNG was reduced (the reference asserts bits == 8, Gltf.cpp:1053)
    std::vector<uint8_t> rgba;
};
struct ImageF {            // RGB32F, row-major, top row first
    int width = 0, height = 0;
    bool half_source = false;   // EXR HALF channels (the reference uploads those as R16G16B16A16_FLOAT)
    std::vector<float> rgb;
};

bool zlib_inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out, std::string& err, size_t size_hint = 0);
bool decode_png(const uint8_t* data, size_t n, Image8& out, std::string& err);
bool decode_jpeg(const uint8_t* data, size_t n, Image8& out, std::string& err);
bool decode_image8(const uint8_t* data, size_t n, Image8& out, std::string& err);     // sniffs PNG / JPEG
bool decode_hdr(const uint8_t* data, size_t n, ImageF& out, std::string& err);
// single_channel: the lookup-table loader's rules (GpuResources.cpp:72-132: one HALF channel, replicated into r, g, b)
bool decode_exr(const uint8_t* data, size_t n, ImageF& out, std::string& err, bool single_channel = false);
bool read_file(const std::string& path, std::vector<uint8_t>& out, std::string& err);
// writers: PNG from RGBA8 rows (3 or 4 channels kept; own deflate: LZ77 + fixed Huffman), PFM and uncompressed
// scan-line OpenEXR (HALF or FLOAT, channels B G R) from RGB32F
bool encode_png(const uint8_t* rgba, int w, int h, int channels, std::vector<uint8_t>& out, std::string& err);
bool write_png(const std::string& path, const uint8_t* rgba, int w, int h, int channels, std::string& err);
bool write_pfm(const std::string& path, const float* rgb, int w, int h, std::string& err);
bool write_exr(const std::string& path, const float* rgb, int w, int h, bool half, std::string& err);
float half_to_float(uint16_t h);

}  // namespace hostimg
