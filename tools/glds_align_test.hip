// glds_align_test.hip -- developer probe: does an LDS-DMA (global_load_lds_dwordx4) land correctly when its LDS
// destination base (M0) is not 16-byte aligned?  For every shift 0..31 one wave copies 1 KiB from a 16-byte aligned
// global address to LDS base + shift, then the whole LDS window is written back byte by byte for the host to check.
//   hipcc --offload-arch=gfx950 -O2 tools/glds_align_test.hip -o build_probes/glds_align_test && build_probes/glds_align_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(64) void k(const uint8_t *src, uint8_t *out, int shift, int gshift)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) lds[i] = 0xEE;
    __syncthreads();
    const uint32_t d = (uint32_t)(uintptr_t)&lds[256] + (uint32_t)shift;
    const uint32_t dd = __builtin_amdgcn_readfirstlane(d);
    const uint32_t voff = (uint32_t)gshift + 16u * lane;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(voff), "s"(src), "s"(dd) : "memory");
    __syncthreads();
    for (int i = lane; i < 2048; i += 64) out[i] = lds[i];
}

int main()
{
    std::vector<uint8_t> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (uint8_t)((i * 7 + (i >> 8) * 13 + 1) & 0xff) | 1;   // never 0xEE? keep simple: force odd
    for (auto &b : h) if (b == 0xEE) b = 0x11;
    uint8_t *ds, *dout;
    hipMalloc(&ds, 4096); hipMalloc(&dout, 2048);
    hipMemcpy(ds, h.data(), 4096, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int gshift = 0; gshift <= 16; gshift += 16)
        for (int shift = 0; shift < 32; ++shift) {
            k<<<1, 64>>>(ds, dout, shift, gshift);
            std::vector<uint8_t> o(2048);
            hipMemcpy(o.data(), dout, 2048, hipMemcpyDeviceToHost);
            int bad = 0, first = -1;
            for (int i = 0; i < 2048; ++i) {
                const int rel = i - 256 - shift;
                const uint8_t want = (rel >= 0 && rel < 1024) ? h[gshift + rel] : 0xEE;
                if (o[i] != want) { ++bad; if (first < 0) first = i; }
            }
            printf("gshift %2d lds shift %2d: %s (%d bytes differ, first at %d)\n", gshift, shift, bad ? "WRONG" : "ok", bad, first);
            bad_total += bad;
        }
    printf("%s\n", bad_total ? "SOME SHIFTS WRONG" : "ALL SHIFTS OK");
    return 0;
}
