// imgtool.cpp -- exercises the CLIs' image codec without a GPU (used by tests/test_cli_image_io.py).
//   gh_imgtool decode <in.(png|jpg|ppm|pgm)> <out.raw>      -> prints "w h channels"
//   gh_imgtool encode <w> <h> <channels> <in.raw> <out.png>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "image_io.h"

int main(int argc, char **argv)
{
    std::string err;
    if (argc == 4 && !strcmp(argv[1], "decode")) {
        GhImage img;
        if (!gh_image_load(argv[2], img, err)) { fprintf(stderr, "decode failed: %s\n", err.c_str()); return 1; }
        FILE *f = fopen(argv[3], "wb");
        if (!f) return 1;
        fwrite(img.pixels.data(), 1, img.pixels.size(), f);
        fclose(f);
        printf("%d %d %d\n", img.width, img.height, img.channels);
        return 0;
    }
    if (argc == 7 && !strcmp(argv[1], "encode")) {
        const int w = atoi(argv[2]), h = atoi(argv[3]), c = atoi(argv[4]);
        std::vector<uint8_t> px((size_t)w * h * c);
        FILE *f = fopen(argv[5], "rb");
        if (!f || fread(px.data(), 1, px.size(), f) != px.size()) { fprintf(stderr, "short read\n"); return 1; }
        fclose(f);
        if (!gh_image_write_png(argv[6], w, h, c, px.data(), err)) { fprintf(stderr, "encode failed: %s\n", err.c_str()); return 1; }
        return 0;
    }
    fprintf(stderr, "usage: %s decode <in> <out.raw> | encode <w> <h> <c> <in.raw> <out.png>\n", argv[0]);
    return 2;
}
