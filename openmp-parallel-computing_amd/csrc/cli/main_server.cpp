// main_server.cpp -- bin/grayhipd: the drop-in CLIs' work in ONE resident process.
//   grayhipd <unix socket path>
// The reference's wrappers (microservices/grayscale/app.py:33, the RabbitMQ worker) start `bin/grayscale` once per job;
// with a GPU behind it every start pays HIP start-up (~0.1 s, more than decode + kernels + encode of a 1080p frame).
// Start this server once per host / container, export GRAYHIP_SERVER=<socket> for the wrappers, and their unchanged
// `bin/grayscale in out passes` calls become thin clients: same stdout, stderr and exit codes, same output file.
// One thread per connection: decode and PNG encode of different requests overlap, the GPU section is serialised inside
// libgrayhip.  The per-request arithmetic mode travels with the request (GRAYHIP_MODE of the client).
#include <signal.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include "cli_core.h"
#include "grayhip.h"

namespace {

std::atomic<bool> g_stop{false};
int g_listen_fd = -1;
std::string g_path;

void on_signal(int) { g_stop = true; if (g_listen_fd >= 0) shutdown(g_listen_fd, SHUT_RDWR); }

void serve(int fd)
{
    std::string head, in, out, passes, mode;
    GhCliResult r;
    r.exit_code = 1;
    char kind_s[16] = "";
    int nf = 0;
    if (gh_wire_recv_line(fd, head, 64) && sscanf(head.c_str(), "GRAYHIP1 %15s %d", kind_s, &nf) == 2 && nf == 4 &&
        gh_wire_recv_field(fd, in, 1 << 16) && gh_wire_recv_field(fd, out, 1 << 16) && gh_wire_recv_field(fd, passes, 32) &&
        gh_wire_recv_field(fd, mode, 16)) {
        const GhCliKind kind = strcmp(kind_s, "sobel") == 0 ? GH_CLI_SOBEL : GH_CLI_GRAY;
        // the client's GRAYHIP_MODE travels with the request; requests run concurrently (decode / encode in parallel,
        // the GPU section is serialised inside libgrayhip)
        const int m = mode.empty() ? gh_default_mode() : ((mode == "fma" || mode == "FMA") ? GH_MODE_FMA : GH_MODE_IEEE);
        r = gh_cli_run(kind, "grayhipd", in.c_str(), out.c_str(), passes.c_str(), m);
    } else {
        r.err = "grayhipd: malformed request\n";
    }
    const std::string code = std::to_string(r.exit_code) + "\n";
    (void)(gh_wire_send_all(fd, code.data(), code.size()) && gh_wire_send_field(fd, r.out) && gh_wire_send_field(fd, r.err));
    close(fd);
}

}  // namespace

int main(int argc, char *argv[])
{
    if (argc != 2) {
        fprintf(stderr, "usage: %s <unix socket path>\n", argv[0]);
        return 2;
    }
    g_path = argv[1];
    const char *dev = getenv("GRAYHIP_DEVICE");
    if (gh_init(dev ? atoi(dev) : 0) != GH_OK) {        // no CPU fallback: without a GPU there is nothing to serve
        fprintf(stderr, "grayhipd: %s\n", gh_last_error());
        return 1;
    }
    sockaddr_un addr;
    memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    if (g_path.size() >= sizeof addr.sun_path) { fprintf(stderr, "grayhipd: socket path too long\n"); return 2; }
    strcpy(addr.sun_path, g_path.c_str());
    unlink(g_path.c_str());
    g_listen_fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (g_listen_fd < 0 || bind(g_listen_fd, reinterpret_cast<sockaddr *>(&addr), sizeof addr) != 0 || listen(g_listen_fd, 128) != 0) {
        perror("grayhipd: socket");
        return 1;
    }
    chmod(g_path.c_str(), 0600);
    signal(SIGINT, on_signal);
    signal(SIGTERM, on_signal);
    signal(SIGPIPE, SIG_IGN);
    printf("grayhipd: serving on %s\n", g_path.c_str());
    fflush(stdout);
    while (!g_stop) {
        const int fd = accept(g_listen_fd, nullptr, nullptr);
        if (fd < 0) {
            if (g_stop) break;
            continue;
        }
        std::thread(serve, fd).detach();
    }
    close(g_listen_fd);
    unlink(g_path.c_str());
    return 0;
}
