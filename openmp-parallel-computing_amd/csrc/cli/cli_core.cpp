// cli_core.cpp -- see cli_core.h
#include "cli_core.h"

#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "grayhip.h"
#include "image_io.h"

namespace {

std::string fmt(const char *f, ...) __attribute__((format(printf, 1, 2)));
std::string fmt(const char *f, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}

}  // namespace

GhCliResult gh_cli_run(GhCliKind kind, const char *prog, const char *in_path, const char *out_path, const char *passes_arg, int mode)
{
    GhCliResult r;
    // HIP start-up (~0.1 s: runtime, context, code object) overlaps the image decode; a failure is reported by the first
    // real call below, which finds the library already (un)initialised.  In the resident server this is a no-op.
    std::thread warm([] { const char *e = getenv("GRAYHIP_DEVICE"); (void)gh_init(e ? atoi(e) : 0); });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{warm};
    GhImage img;
    std::string err;
    if (!gh_image_load(in_path, img, err)) {
        r.err = kind == GH_CLI_GRAY ? "Errore caricando immagine\n" : fmt("Errore caricando immagine \"%s\"\n", in_path);
        if (getenv("GRAYHIP_VERBOSE")) r.err += fmt("grayhip: %s\n", err.c_str());
        r.exit_code = 1;
        return r;
    }
    int passes = passes_arg ? atoi(passes_arg) : 1;
    if (passes < 1) passes = 1;
    double secs = 0.0;
    const int rc = kind == GH_CLI_GRAY
                       ? gh_gray_passes_host(img.pixels.data(), img.width, img.height, img.channels, passes, mode, &secs)
                       : gh_pipeline_gray_sobel_host(img.pixels.data(), img.width, img.height, img.channels, passes, mode, &secs);
    if (rc != GH_OK) {
        r.err = fmt("grayhip: %s\n", gh_last_error());   // no CPU fallback: a GPU failure is an error exit
        r.exit_code = 1;
        return r;
    }
    r.out = kind == GH_CLI_GRAY ? fmt("Compute kernel ×%d: %.4f s\n", passes, secs)
                                : fmt("Compute kernel (grayscale + sobel) ×%d: %.4f s\n", passes, secs);
    if (!gh_image_write_png(out_path, img.width, img.height, img.channels, img.pixels.data(), err)) {
        if (kind == GH_CLI_GRAY) {
            r.err = "Errore nel salvataggio\n";
            r.exit_code = 1;
        } else {
            r.err = fmt("Errore nel salvataggio di \"%s\"\n", out_path);     // main_with_sobel.c: message only, exit code stays 0
        }
    }
    (void)prog;
    return r;
}

// ---- wire helpers ---------------------------------------------------------------------------------------------------
bool gh_wire_send_all(int fd, const void *p, size_t n)
{
    const char *c = static_cast<const char *>(p);
    while (n) {
        const ssize_t w = send(fd, c, n, MSG_NOSIGNAL);
        if (w <= 0) return false;
        c += w;
        n -= (size_t)w;
    }
    return true;
}

bool gh_wire_send_field(int fd, const std::string &s)
{
    const std::string head = std::to_string(s.size()) + "\n";
    return gh_wire_send_all(fd, head.data(), head.size()) && gh_wire_send_all(fd, s.data(), s.size());
}

bool gh_wire_recv_line(int fd, std::string &line, size_t max_len)
{
    line.clear();
    char c;
    while (line.size() <= max_len) {
        const ssize_t r = recv(fd, &c, 1, 0);
        if (r <= 0) return false;
        if (c == '\n') return true;
        line.push_back(c);
    }
    return false;
}

bool gh_wire_recv_field(int fd, std::string &s, size_t max_len)
{
    std::string head;
    if (!gh_wire_recv_line(fd, head, 20) || head.empty()) return false;
    for (char c : head)
        if (c < '0' || c > '9') return false;
    const unsigned long long n = strtoull(head.c_str(), nullptr, 10);
    if (n > max_len) return false;
    s.resize((size_t)n);
    size_t got = 0;
    while (got < n) {
        const ssize_t r = recv(fd, &s[got], (size_t)n - got, 0);
        if (r <= 0) return false;
        got += (size_t)r;
    }
    return true;
}

bool gh_cli_try_server(GhCliKind kind, const char *in_path, const char *out_path, const char *passes_arg, GhCliResult &res)
{
    const char *sock = getenv("GRAYHIP_SERVER");
    if (!sock || !*sock) return false;
    sockaddr_un addr;
    memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    if (strlen(sock) >= sizeof addr.sun_path) return false;
    strcpy(addr.sun_path, sock);
    const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (fd < 0) return false;
    if (connect(fd, reinterpret_cast<sockaddr *>(&addr), sizeof addr) != 0) { close(fd); return false; }   // no server: run in-process
    // the server resolves nothing relative to ITS working directory: send absolute paths
    auto absolute = [](const char *p) -> std::string {
        if (p[0] == '/') return p;
        char cwd[PATH_MAX];
        if (!getcwd(cwd, sizeof cwd)) return p;
        return std::string(cwd) + "/" + p;
    };
    const char *mode = getenv("GRAYHIP_MODE");
    bool ok = gh_wire_send_all(fd, kind == GH_CLI_GRAY ? "GRAYHIP1 gray 4\n" : "GRAYHIP1 sobel 4\n", kind == GH_CLI_GRAY ? 16 : 17) &&
              gh_wire_send_field(fd, absolute(in_path)) && gh_wire_send_field(fd, absolute(out_path)) &&
              gh_wire_send_field(fd, passes_arg ? passes_arg : "1") && gh_wire_send_field(fd, mode ? mode : "");
    std::string code;
    ok = ok && gh_wire_recv_line(fd, code, 8) && gh_wire_recv_field(fd, res.out, 1 << 20) && gh_wire_recv_field(fd, res.err, 1 << 20);
    close(fd);
    if (!ok) return false;      // a broken conversation: fall back to doing the work here
    res.exit_code = atoi(code.c_str());
    return true;
}
