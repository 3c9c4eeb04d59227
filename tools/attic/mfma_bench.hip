// Dev tool: MFMA 16x16x4 f32 issue patterns, cycles measured with s_memtime inside the kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(int n, float* sink, unsigned long long* cyc) {
    f32x4 a[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {          // 4 dependent MFMAs per accumulator, accumulators in sequence (current kernel order)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[q], 0, 0, 0);
                a[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a[q], 0, 0, 0);
                a[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a[q], 0, 0, 0);
                a[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a[q], 0, 0, 0);
            }
        } else {                  // round-robin over the 4 accumulators
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(r & 1 ? x : y, r & 2 ? x : y, a[q], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a[0][0] + a[1][0] + a[2][0] + a[3][0] == 12345.f) sink[0] = 1;
}
int main() {
    float* sink; hipMalloc(&sink, 4);
    unsigned long long* cyc; hipMalloc(&cyc, 8 * 4096);
    unsigned long long h[4096];
    for (int waves : {4, 8, 16}) for (int mode : {0, 1}) {
        int n = 8;   // 8 iterations x 16 MFMAs = 128 MFMAs per wave
        for (int rep = 0; rep < 3; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(192), dim3(waves * 64), 0, 0, n, sink, cyc);
            else hipLaunchKernelGGL(k<1>, dim3(192), dim3(waves * 64), 0, 0, n, sink, cyc);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, 8 * 192 * waves, hipMemcpyDeviceToHost);
        unsigned long long mx = 0, mn = ~0ull, sum = 0;
        for (int i = 0; i < 192 * waves; ++i) { mx = h[i] > mx ? h[i] : mx; mn = h[i] < mn ? h[i] : mn; sum += h[i]; }
        printf("waves/WG %2d mode %d: 128 MFMAs per wave: cycles min %llu avg %llu max %llu  (ideal %d)\n", waves, mode, mn,
               sum / (192 * waves), mx, 128 * 32 * waves / 4);
    }
    return 0;
}
