// Prints the host-side launch-rule helpers of stream_core.h for a list of cases (tests/test_launch_rules.py compares them with
// a Python restatement and with the figures DESIGN.md quotes).  Host code only: built with hipcc, runs without a GPU.
#include <cstdio>
#include <utility>
#include "stream_core.h"
int main()
{
    const long long pitches[] = {16384, 32768, 16368, 16370, 12288, 10240, 20480, 24576, 8192, 15360, 7680, 4096, 49152, 36864, 30720, 49104};
    for (long long p : pitches)
        for (int r : {16, 32, 64})
            printf("phases %lld %d %d\n", p, r, stream_core::band_start_phases((long long)r * p));
    for (long long p : pitches) {
        printf("prefer %lld 64 16 %d\n", p, (int)stream_core::prefer_short_bands(p, 64, 16));
        printf("prefer %lld 64 32 %d\n", p, (int)stream_core::prefer_short_bands(p, 64, 32));
        printf("prefer %lld 32 16 %d\n", p, (int)stream_core::prefer_short_bands(p, 32, 16));
    }
    const long long nbs[] = {1, 255, 256, 257, 384, 480, 500, 512, 513, 768, 1024, 1536};
    for (long long nb : nbs) printf("balance %lld %.6f\n", nb, stream_core::wg_balance(nb, 256));
    const int shapes[][3] = {{2, 8200, 64}, {2, 16390, 64}, {2, 8192, 64}, {1, 4100, 16}, {2, 4320, 16}, {1, 1080, 16}, {2, 3000, 16}, {3, 8200, 16}, {4, 4100, 16}, {4, 100000, 16}};
    for (const auto &s : shapes) printf("overflow %d %d %d %d\n", s[0], s[1], s[2], (int)stream_core::shifted_band_overflows(s[0], s[1], s[2], 2, 256));
    // band heights the launchers pick (stream_core::fused_band_rows / sobel_band_rows) for a device with `cus` compute units:
    // "rows_fused cus width rows R" (RGB8: pitch 3 w, a workgroup spans 4096 columns), "rows_sobel cus width rows R"
    const int frames[][2] = {{16384, 16384}, {16384, 8192}, {16384, 4096}, {16384, 2048}, {12288, 8192}, {10240, 16384}, {4096, 2048}, {4096, 4096},
                             {2048, 2048}, {1920, 1080}, {8192, 1024}, {16384, 512}};
    for (int cus : {256, 32, 304})
        for (const auto &f : frames) {
            const int gx = (f[0] + 4095) / 4096;
            printf("rows_fused %d %d %d %d\n", cus, f[0], f[1], stream_core::fused_band_rows(gx, f[1], 3ll * f[0], cus, true));
            printf("rows_sobel %d %d %d %d\n", cus, f[0], f[1], stream_core::sobel_band_rows(gx, f[1], f[0], cus, true, true, true));
            printf("rows_fused_shard %d %d %d %d\n", cus, f[0], f[1], stream_core::fused_band_rows(gx, f[1], 3ll * f[0], cus, true, true));
        }
    // workgroup -> band mapping requests as sanitised for a device with `xcds` XCDs: "order xcds bands groups_x requested granted grid"
    for (int xcds : {8, 1, 4})
        for (int req : {0, 6, 8, 4, 2, 2006, 108, 16})
            for (const auto &bg : {std::pair<int, int>{256, 4}, {32, 4}, {33, 2}, {1, 4}}) {
                stream_core::BandPlan p = stream_core::make_band_plan(bg.first * 64, 64, -1, 0, -1, 0);
                p.order = stream_core::sanitize_order(req, p.bands, bg.second, xcds);
                printf("order %d %d %d %d %d %lld\n", xcds, bg.first, bg.second, req, p.order, stream_core::plan_grid(p, bg.second));
            }
    return 0;
}
