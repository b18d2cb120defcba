// How many kernel dispatches per second can the GPU take, summed over K HIP streams, when every kernel is tiny and each stream is
// an in-order chain?  (The receive pipeline is ~12 kernels per block on 13 streams.)
// hipcc --offload-arch=gfx950 -O2 -o dispatch_probe dispatch_probe.hip && GPU_MAX_HW_QUEUES=16 ./dispatch_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void tiny(unsigned *sink, unsigned spin) { unsigned v = 0; for (unsigned i = 0; i < spin; i++) v += i * i; if (v == 0xFFFFFFFFu) *sink = v; }
int main()
{
    unsigned *sink; hipMalloc(&sink, 4);
    std::vector<hipStream_t> st(16);
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 2000;
    for (unsigned spin : { 0u, 20000u }) for (int G : { 1, 256 })
        for (int K : { 1, 2, 4, 8, 13, 16 }) {
            for (int rep = 0; rep < 2; rep++) {
                hipDeviceSynchronize();
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < N; i++) for (int k = 0; k < K; k++) hipLaunchKernelGGL(tiny, dim3(G), dim3(64), 0, st[k], sink, spin);
                auto t1 = std::chrono::steady_clock::now();
                hipDeviceSynchronize();
                double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                double hms = std::chrono::duration<double, std::milli>(t1 - t0).count();
                if (rep) std::printf("spin %5u grid %3d streams %2d: %7.2f us per kernel overall (%.0f k kernels/s), host launch %5.2f us each\n", spin, G, K, ms * 1e3 / (N * K), N * K / ms, hms * 1e3 / (N * K));
            }
        }
    return 0;
}
