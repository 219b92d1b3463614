// main_sobel.cpp -- drop-in for the reference's grayscale+Sobel CLI (monolithic/src/main_with_sobel.c:16-92):
//   grayscale_sobel <input_img> <output_img.png> [passes]
// Per pass: luma -> 3x3 Sobel -> edge replicated into R,G,B (alpha untouched), all on MI355X.
// As in the reference, a failed PNG save only prints a message and the exit code stays 0.
// GRAYHIP_SERVER=<socket>: served by a resident bin/grayhipd when one listens there (see main_gray.cpp).
#include <cstdio>

#include "cli_core.h"
#include "grayhip.h"

int main(int argc, char *argv[])
{
    if (argc < 3) {
        fprintf(stderr, "Uso: %s <input_img> <output_img.png> [passaggi_kernel]\n", argv[0]);
        return 1;
    }
    const char *passes = argc >= 4 ? argv[3] : nullptr;
    GhCliResult r;
    if (!gh_cli_try_server(GH_CLI_SOBEL, argv[1], argv[2], passes, r)) r = gh_cli_run(GH_CLI_SOBEL, argv[0], argv[1], argv[2], passes, gh_default_mode());
    fputs(r.out.c_str(), stdout);
    fputs(r.err.c_str(), stderr);
    return r.exit_code;
}
