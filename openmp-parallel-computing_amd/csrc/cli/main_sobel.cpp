// main_sobel.cpp -- drop-in for the reference's grayscale+Sobel CLI (monolithic/src/main_with_sobel.c:16-92):
//   grayscale_sobel <input_img> <output_img.png> [passes]
// Per pass: luma -> 3x3 Sobel -> edge replicated into R,G,B (alpha untouched), all on MI355X.
// As in the reference, a failed PNG save only prints a message and the exit code stays 0.
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
        fprintf(stderr, "Errore caricando immagine \"%s\"\n", argv[1]);
        if (getenv("GRAYHIP_VERBOSE")) fprintf(stderr, "grayhip: %s\n", err.c_str());
        return 1;
    }
    int passes = (argc >= 4) ? atoi(argv[3]) : 1;
    if (passes < 1) passes = 1;

    double secs = 0.0;
    if (gh_pipeline_gray_sobel_host(img.pixels.data(), img.width, img.height, img.channels, passes, gh_default_mode(), &secs) != GH_OK) {
        fprintf(stderr, "grayhip: %s\n", gh_last_error());
        return 1;
    }
    printf("Compute kernel (grayscale + sobel) ×%d: %.4f s\n", passes, secs);

    if (!gh_image_write_png(argv[2], img.width, img.height, img.channels, img.pixels.data(), err))
        fprintf(stderr, "Errore nel salvataggio di \"%s\"\n", argv[2]);
    return 0;
}
