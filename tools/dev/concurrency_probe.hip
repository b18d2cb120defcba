// How many kernels from different HIP streams execute at the same time on this GPU?  K streams, one thin kernel each (G
// workgroups of 64 threads spinning for ~1 ms on the wall clock); wall time of the batch / 1 ms = how many ran one after the
// other.  hipcc --offload-arch=gfx950 -O2 -o concurrency_probe concurrency_probe.hip && GPU_MAX_HW_QUEUES=16 ./concurrency_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long ticks, unsigned *sink)
{
    const unsigned long long t0 = wall_clock64();
    unsigned v = 0;
    while (wall_clock64() - t0 < ticks) v++;
    if (v == 0xFFFFFFFFu) *sink = v;
}
int main()
{
    unsigned *sink; hipMalloc(&sink, 4);
    int rate_khz = 100000; hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const unsigned long long ticks = (unsigned long long)rate_khz;      // 1 ms
    std::vector<hipStream_t> st(32);
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int G : { 1, 64, 1024 })
        for (int K : { 1, 2, 4, 6, 8, 12, 16, 24, 32 }) {
            for (int rep = 0; rep < 2; rep++) {
                hipDeviceSynchronize();
                auto t0 = std::chrono::steady_clock::now();
                for (int k = 0; k < K; k++) hipLaunchKernelGGL(spin, dim3(G), dim3(64), 0, st[k], ticks, sink);
                hipDeviceSynchronize();
                double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (rep) std::printf("workgroups %4d  streams %2d: %.2f ms -> %.1f kernels at a time\n", G, K, ms, K / ms);
            }
        }
    return 0;
}
