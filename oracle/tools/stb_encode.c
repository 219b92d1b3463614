/* stb_encode.c -- fixture tool (build container only): the REFERENCE's own PNG writer (the stb_image_write.h it vendors,
 * included where it lies under /root/reference) applied to raw pixels, exactly as monolithic/src/main.c:41 calls it.
 *   stb_encode <w> <h> <channels> <in.raw> <out.png>
 * Built by oracle/Makefile into oracle/_ref/; tests/test_cli_image_io.py compares this repo's PNG files with its output
 * byte for byte. */
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include "stb_image_write.h"
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv)
{
    if (argc != 6) { fprintf(stderr, "usage: %s <w> <h> <channels> <in.raw> <out.png>\n", argv[0]); return 2; }
    int w = atoi(argv[1]), h = atoi(argv[2]), c = atoi(argv[3]);
    size_t n = (size_t)w * h * c;
    unsigned char *px = (unsigned char *)malloc(n ? n : 1);
    FILE *f = fopen(argv[4], "rb");
    if (!f || fread(px, 1, n, f) != n) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    return stbi_write_png(argv[5], w, h, c, px, w * c) ? 0 : 1;
}
