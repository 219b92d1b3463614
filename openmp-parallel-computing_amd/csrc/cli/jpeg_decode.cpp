// jpeg_decode.cpp -- placeholder until the stb-compatible JPEG decoder lands.
#include "image_io.h"

bool gh_jpeg_decode(const uint8_t *, size_t, GhImage &, std::string &err)
{
    err = "JPEG input is not supported yet (use PNG or binary PPM/PGM)";
    return false;
}
