/* stb_decode.c -- fixture-generation helper (test infrastructure, build container only).
 * Dumps the pixels of an image exactly as the reference CLI sees them, i.e. as decoded by the
 * stb_image.h the reference vendors (monolithic/src/main.c:21 calls stbi_load(path,&w,&h,&c,0)).
 * The header is included in place from /root/reference via -I; nothing is copied.
 * usage: stb_decode <in.jpg|png> <out.raw>   -> prints "w h channels" on stdout */
#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"
#include <stdio.h>

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <in> <out.raw>\n", argv[0]); return 2; }
    int w, h, c;
    unsigned char *px = stbi_load(argv[1], &w, &h, &c, 0);
    if (!px) { fprintf(stderr, "decode failed: %s\n", stbi_failure_reason()); return 1; }
    FILE *f = fopen(argv[2], "wb");
    if (!f) return 1;
    fwrite(px, 1, (size_t)w * h * c, f);
    fclose(f);
    printf("%d %d %d\n", w, h, c);
    return 0;
}
