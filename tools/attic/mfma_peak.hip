// Dev tool: sustained f32 MFMA rate of the whole chip (all CUs, 2 waves per SIMD, independent accumulators) and the shader
// clock it runs at (s_memtime ticks per 100 MHz s_memrealtime tick): what the "157 TF/s" peak is worth under load.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(int n, float* sink, unsigned long long* out) {
    float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if (MODE == 0) {
        f32x16 a[4];
        for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) a[q][r] = 0.f;
        for (int i = 0; i < n; ++i)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[q], 0, 0, 0);
        s = a[0][0] + a[1][0] + a[2][0] + a[3][0];
    } else {
        f32x4 a[8];
        for (int q = 0; q < 8; ++q) a[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < n; ++i)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 8; ++q) a[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[q], 0, 0, 0);
        for (int q = 0; q < 8; ++q) s += a[q][0];
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = r1 - r0; }
    if (s == 12345.f) sink[0] = 1;
}
int main() {
    float* sink; hipMalloc(&sink, 4);
    unsigned long long* out; hipMalloc(&out, 16 * 1024);
    unsigned long long h[2048];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < (mode == 0 ? 150 : 6); ++rep) {
            const int n = 20000, wgs = 256;
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(512), 0, 0, n, sink, out);
            else hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(512), 0, 0, n, sink, out);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, out, wgs * 16, hipMemcpyDeviceToHost);
            double flops = mode == 0 ? (double)wgs * 8 * n * 16 * 4096.0 : (double)wgs * 8 * n * 32 * 2048.0;
            double mhz = 0; for (int i = 0; i < wgs; ++i) mhz += 100.0 * h[2 * i] / h[2 * i + 1]; mhz /= wgs;
            if (rep < 3 || rep % 25 == 0) printf("%s rep %d: %.3f ms  %.1f TF/s  shader clock %.0f MHz (s_memtime / s_memrealtime)\n", mode == 0 ? "32x32x2 " : "16x16x4 ", rep, ms, flops / ms / 1e9, mhz);
        }
    return 0;
}
