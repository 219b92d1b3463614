// main_gray.cpp -- drop-in for the reference's grayscale CLI (monolithic/src/main.c:14-49):
//   grayscale <input_img> <output_img.png> [passes]
// Same argv, same stdout line, same exit codes and stderr texts; the kernel passes run on MI355X
// through libgrayhip's C ABI (one upload, all passes in one launch, one download).  stderr stays empty on
// success: the reference's bench script parses the child's stderr together with /usr/bin/time's line.
// With GRAYHIP_SERVER=<socket> and a bin/grayhipd listening there, the request is served by that warm
// process instead (no HIP start-up per call); without one, everything happens here.
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
    if (!gh_cli_try_server(GH_CLI_GRAY, argv[1], argv[2], passes, r)) r = gh_cli_run(GH_CLI_GRAY, argv[0], argv[1], argv[2], passes, gh_default_mode());
    fputs(r.out.c_str(), stdout);
    fputs(r.err.c_str(), stderr);
    return r.exit_code;
}
