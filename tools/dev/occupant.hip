// An idle occupant: workgroups that hold registers (and optionally LDS) on every CU without issuing instructions, while the
// receive pipeline runs beside them -- if throughput falls in proportion to what they hold, the pipeline is bound by register-file
// (or LDS) capacity, i.e. by how many waves of its kernels fit on the chip at once, not by any throughput resource.
// hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o liboccupant.so occupant.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NREG, int LDSF>
__global__ __launch_bounds__(256) void occupant(const volatile int *stop, float *sink, float seed)
{
    __shared__ float lds[LDSF > 0 ? LDSF : 1];
    float r[NREG];
#pragma unroll
    for (int i = 0; i < NREG; i++) r[i] = seed * (float)(i + 1) + (float)threadIdx.x;
    if (LDSF > 0) lds[threadIdx.x % (LDSF > 0 ? LDSF : 1)] = seed;
    while (*stop == 0) __builtin_amdgcn_s_sleep(127);
    float s = LDSF > 0 ? lds[0] : 0.0f;
#pragma unroll
    for (int i = 0; i < NREG; i++) s += r[i];
    if (s == 12345.678f) *sink = s;
}
static int *h_stop = nullptr, *h_one = nullptr; static float *d_sink = nullptr; static hipStream_t st = nullptr, st2 = nullptr;   // (h_stop: DEVICE memory -- polling a host flag from a thousand waves floods the PCIe link)
extern "C" int occupant_start(int regs, int lds_kb, int wgs)
{
    if (!h_stop) { hipMalloc((void **)&h_stop, 4); hipHostMalloc((void **)&h_one, 8, hipHostMallocDefault); h_one[0] = 0; h_one[1] = 1; hipMalloc(&d_sink, 4);
                   hipStreamCreateWithFlags(&st, hipStreamNonBlocking); hipStreamCreateWithFlags(&st2, hipStreamNonBlocking); }
    hipMemcpyAsync(h_stop, h_one, 4, hipMemcpyHostToDevice, st2); hipStreamSynchronize(st2);
    if (regs <= 64) { if (lds_kb) hipLaunchKernelGGL((occupant<40, 8192>), dim3(wgs), dim3(256), 0, st, h_stop, d_sink, 1.0f); else hipLaunchKernelGGL((occupant<40, 0>), dim3(wgs), dim3(256), 0, st, h_stop, d_sink, 1.0f); }
    else if (regs <= 128) { if (lds_kb) hipLaunchKernelGGL((occupant<100, 8192>), dim3(wgs), dim3(256), 0, st, h_stop, d_sink, 1.0f); else hipLaunchKernelGGL((occupant<100, 0>), dim3(wgs), dim3(256), 0, st, h_stop, d_sink, 1.0f); }
    else { if (lds_kb) hipLaunchKernelGGL((occupant<220, 8192>), dim3(wgs), dim3(256), 0, st, h_stop, d_sink, 1.0f); else hipLaunchKernelGGL((occupant<220, 0>), dim3(wgs), dim3(256), 0, st, h_stop, d_sink, 1.0f); }
    return (int)hipGetLastError();
}
extern "C" int occupant_stop(void) { if (!h_stop) return 0; hipMemcpyAsync(h_stop, h_one + 1, 4, hipMemcpyHostToDevice, st2); hipStreamSynchronize(st2); return (int)hipStreamSynchronize(st); }
