// What does s_memrealtime count on this device?  Two one-thread kernels 200 ms of host time apart (developer probe: the halo-flag form's
// time-outs and standstill reports are in these ticks).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
__global__ void k_stamp(unsigned long long *out) { out[0] = __builtin_amdgcn_s_memrealtime(); out[1] = wall_clock64(); out[2] = clock64(); }
int main()
{
    unsigned long long *d, a[3], b[3];
    hipMalloc(&d, 24);
    k_stamp<<<1, 1>>>(d); hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        k_stamp<<<1, 1>>>(d); hipMemcpy(a, d, 24, hipMemcpyDeviceToHost);
        auto t0 = std::chrono::steady_clock::now();
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
        k_stamp<<<1, 1>>>(d); hipMemcpy(b, d, 24, hipMemcpyDeviceToHost);
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
        printf("s_memrealtime %.2f MHz   wall_clock64 %.2f MHz   clock64 %.1f MHz   (hipDeviceAttributeWallClockRate %d kHz, %.3f s)\n",
               (b[0] - a[0]) / s / 1e6, (b[1] - a[1]) / s / 1e6, (b[2] - a[2]) / s / 1e6, rate, s);
    }
    return 0;
}
