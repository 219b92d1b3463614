// cli_core.h -- what the two drop-in CLIs do, as functions: used in-process by bin/grayscale / bin/grayscale_sobel and by
// the resident server bin/grayhipd, which runs the same code for many requests in one warm process (the
// reference's service wrappers start one process per job -- microservices/grayscale/app.py:33 -- and would pay HIP
// start-up, ~0.1 s, on every call).
#ifndef GH_CLI_CORE_H
#define GH_CLI_CORE_H

#include <string>

enum GhCliKind { GH_CLI_GRAY = 0, GH_CLI_SOBEL = 1 };

struct GhCliResult {
    int exit_code = 0;
    std::string out;   // what the CLI prints on stdout
    std::string err;   // ... and on stderr (empty on success: the reference's bench script parses it)
};

// monolithic/src/main.c:14-49 (gray) / main_with_sobel.c:16-92 (sobel) with the kernels on the GPU.
// prog = argv[0] for the usage text; paths as given on the command line.
// mode: GH_MODE_IEEE / GH_MODE_FMA (the CLIs pass gh_default_mode(), i.e. GRAYHIP_MODE; the server passes its client's).
GhCliResult gh_cli_run(GhCliKind kind, const char *prog, const char *in_path, const char *out_path, const char *passes_arg, int mode);

// Client side of the resident server: if the environment names a server socket (GRAYHIP_SERVER) and a server answers
// there, the request is served by it and true is returned with `res` filled; otherwise false (run in-process).
bool gh_cli_try_server(GhCliKind kind, const char *in_path, const char *out_path, const char *passes_arg, GhCliResult &res);

// Wire format shared by client and server (one request per connection): a header line "GRAYHIP1 <kind> <n fields>\n" then
// fields as "<decimal length>\n<bytes>"; the reply is "<exit code>\n" followed by two such fields (stdout, stderr).
bool gh_wire_send_field(int fd, const std::string &s);
bool gh_wire_recv_field(int fd, std::string &s, size_t max_len);
bool gh_wire_recv_line(int fd, std::string &line, size_t max_len);
bool gh_wire_send_all(int fd, const void *p, size_t n);

#endif
