// main_gray.cpp -- drop-in for the reference's grayscale CLI (monolithic/src/main.c:14-49):
//   grayscale <input_img> <output_img.png> [passes]
// Same argv, same stdout line, same exit codes and stderr texts; the kernel passes run on MI355X
// through libgrayhip's C ABI (one upload, `passes` launches, one download).  stderr stays empty on
// success: the reference's bench script parses the child's stderr together with /usr/bin/time's line.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>

#include "grayhip.h"
#include "image_io.h"

int main(int argc, char *argv[])
{
    if (argc < 3) {
        fprintf(stderr, "Uso: %s <input_img> <output_img.png> [passaggi_kernel]\n", argv[0]);
        return 1;
    }
    // HIP start-up (~0.1 s: runtime, context, code object) overlaps the image decode; a failure is reported by the
    // first real call below, which finds the library already (un)initialised
    std::thread warm([] { const char *e = getenv("GRAYHIP_DEVICE"); (void)gh_init(e ? atoi(e) : 0); });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{warm};
    GhImage img;
    std::string err;
    if (!gh_image_load(argv[1], img, err)) {
        fprintf(stderr, "Errore caricando immagine\n");
        if (getenv("GRAYHIP_VERBOSE")) fprintf(stderr, "grayhip: %s\n", err.c_str());
        return 1;
    }
    int passes = (argc >= 4) ? atoi(argv[3]) : 1;
    if (passes < 1) passes = 1;

    double secs = 0.0;
    if (gh_gray_passes_host(img.pixels.data(), img.width, img.height, img.channels, passes, gh_default_mode(), &secs) != GH_OK) {
        fprintf(stderr, "grayhip: %s\n", gh_last_error());   // no CPU fallback: a GPU failure is an error exit
        return 1;
    }
    printf("Compute kernel ×%d: %.4f s\n", passes, secs);

    if (!gh_image_write_png(argv[2], img.width, img.height, img.channels, img.pixels.data(), err)) {
        fprintf(stderr, "Errore nel salvataggio\n");
        return 1;
    }
    return 0;
}
