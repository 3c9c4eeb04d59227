// Dev tool: do kernels from two HIP streams overlap on this box?  (a) two long kernels, (b) two dependent
// chains of short kernels, with small and with chain-like (512 threads, big LDS) workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void spin(long ticks, float* sink) {      // s_memrealtime ticks at 100 MHz
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long t = wall_clock64();
    const unsigned long long end = t + ticks;
    while (wall_clock64() < end) { }
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = (float)(__builtin_readcyclecounter() - t0) + lds[0];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t a, b; hipStreamCreate(&a); hipStreamCreate(&b);
    float* sink; hipMalloc(&sink, 4096 * 4);
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    struct { int grid, block, lds; const char* name; } shapes[] = {
        {64, 64, 0, "64 WG x 64 thr"}, {192, 512, 0, "192 WG x 512 thr"}, {192, 512, 40 * 1024, "192 WG x 512 thr x 40 KB LDS"},
        {192, 512, 100 * 1024, "192 WG x 512 thr x 100 KB LDS"}, {256, 512, 40 * 1024, "256 WG x 512 thr x 40 KB LDS"}};
    for (auto& s : shapes) {
        for (int warm = 0; warm < 2; ++warm) { hipLaunchKernelGGL(spin, s.grid, s.block, s.lds, a, 100, sink); hipLaunchKernelGGL(spin, s.grid, s.block, s.lds, b, 100, sink); }
        hipDeviceSynchronize();
        // (a) one long kernel per stream (500 us each)
        double t0 = now();
        hipLaunchKernelGGL(spin, s.grid, s.block, s.lds, a, 50000, sink);
        hipDeviceSynchronize();
        double one = now() - t0;
        t0 = now();
        hipLaunchKernelGGL(spin, s.grid, s.block, s.lds, a, 50000, sink);
        hipLaunchKernelGGL(spin, s.grid, s.block, s.lds, b, 50000, sink);
        hipDeviceSynchronize();
        double two = now() - t0;
        // (b) chains of 500 x 8 us kernels
        t0 = now();
        for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(spin, s.grid, s.block, s.lds, a, 800, sink);
        hipDeviceSynchronize();
        double c1 = now() - t0;
        t0 = now();
        for (int i = 0; i < 500; ++i) { hipLaunchKernelGGL(spin, s.grid, s.block, s.lds, a, 800, sink); hipLaunchKernelGGL(spin, s.grid, s.block, s.lds, b, 800, sink); }
        hipDeviceSynchronize();
        double c2 = now() - t0;
        printf("%-34s long: one %.0f us, two streams %.0f us | chain of 500x8us: one %.2f ms (%.1f us/launch), two streams %.2f ms\n",
               s.name, one * 1e6, two * 1e6, c1 * 1e3, c1 * 1e6 / 500, c2 * 1e3);
    }
    return 0;
}
