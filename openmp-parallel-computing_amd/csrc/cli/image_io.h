// image_io.h -- image file decode / encode for the drop-in CLIs (host side, no GPU involved).
//
// Stands in for the two calls the reference drivers make into their vendored third-party codec:
//   monolithic/src/main.c:21   stbi_load(path, &w, &h, &channels, 0)      -> gh_image_load
//   monolithic/src/main.c:41   stbi_write_png(path, w, h, channels, data, w*channels) -> gh_image_write_png
// Same contract: native channel count is kept (grey=1, grey+alpha=2, RGB=3, RGBA=4; palette images
// expand to RGB/RGBA; 16-bit samples are reduced to 8 bits), rows are tightly packed.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

struct GhImage {
    int width = 0, height = 0, channels = 0;
    std::vector<uint8_t> pixels;   // height * width * channels, row-major, no padding
};

// Decodes PNG (all colour types / bit depths, Adam7), baseline + progressive JPEG, binary PGM/PPM.
// Returns false and fills `err` on failure.
bool gh_image_load(const std::string &path, GhImage &out, std::string &err);
bool gh_image_load_memory(const uint8_t *data, size_t size, GhImage &out, std::string &err);

// 8-bit PNG, colour type chosen from `channels` (1,2,3,4), adaptive row filters, zlib deflate.
bool gh_image_write_png(const std::string &path, int width, int height, int channels, const uint8_t *pixels,
                        std::string &err);
