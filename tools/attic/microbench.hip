// Micro-benchmarks that decide the LSTM step-kernel structure on MI355X (dev tool, not shipped):
//  1. dependent kernel-boundary cost for a 192..384-workgroup grid
//  2. does a per-XCD L2 keep a read-only 3 MB/XCD weight slice across kernel launches?
//  3. achieved fetch rate of the "each workgroup streams its own 128 KB slice" pattern, cold vs warm
//  4. effective MFMA clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

// each workgroup streams `bytes_per_wg` of its own slice (float4, 256 threads), `reps` times in-kernel
__global__ __launch_bounds__(256) void slice_read(const float4* __restrict__ w, size_t f4_per_wg, int reps, float* sink) {
    const float4* p = w + (size_t)blockIdx.x * f4_per_wg;
    float acc = 0.f;
    for (int r = 0; r < reps; ++r)
        for (size_t i = threadIdx.x; i < f4_per_wg; i += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (i + q * 256 < f4_per_wg) ? p[i + q * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q].x + v[q].y + v[q].z + v[q].w;
        }
    if (acc == 12345.678f) sink[0] = acc;
}

// every workgroup reads the SAME `n` float4 (the h all-gather pattern)
__global__ __launch_bounds__(256) void shared_read(const float4* __restrict__ a, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < n; i += 256 * 8) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (i + q * 256 < n) ? a[i + q * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q].x + v[q].y + v[q].z + v[q].w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// same, but every workgroup starts at a different offset (rotation) so the CUs do not walk the same
// L2 channel in lockstep
__global__ __launch_bounds__(256) void shared_read_rot(const float4* __restrict__ a, size_t n, int rot_f4, float* sink) {
    float acc = 0.f;
    const size_t start = ((size_t)blockIdx.x * rot_f4) % n;
    for (size_t i0 = 0; i0 < n; i0 += 256 * 8) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { size_t i = (start + i0 + threadIdx.x + q * 256) % n; v[q] = a[i]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q].x + v[q].y + v[q].z + v[q].w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// MFMA A-fragment pattern: [32 rows][K] row-major, lane (i = l&15, kq = l>>4) reads float4 at row i, k = kb*16 + 4*kq;
// 4 waves split K.  rot != 0 rotates the K-block order per workgroup.
__global__ __launch_bounds__(256) void frag_read(const float* __restrict__ a, int K, int rot, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nkb = K / 16, per = nkb / 4;
    const int kb0 = wave * per;
    const int r0 = rot ? (blockIdx.x * rot) % per : 0;
    float acc = 0.f;
    for (int i0 = 0; i0 < per; i0 += 8) {
        float4 v[16];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int kb = kb0 + (i0 + q + r0) % per;
            const float* p = a + (size_t)(lane & 15) * K + kb * 16 + 4 * (lane >> 4);
            v[2 * q] = *reinterpret_cast<const float4*>(p);
            v[2 * q + 1] = *reinterpret_cast<const float4*>(p + (size_t)16 * K);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += v[q].x + v[q].y + v[q].z + v[q].w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void mfma_chain(int n, float* sink) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    for (int i = 0; i < n; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
    }
    if (a0[0] + a1[0] == 12345.678f) sink[0] = a0[0];
}

template <typename F>
static float time_ms(hipStream_t s, int iters, F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float* sink; CK(hipMalloc(&sink, 4));
    const size_t WBYTES = 24u << 20;
    float4* w; CK(hipMalloc(&w, WBYTES)); CK(hipMemset(w, 0, WBYTES));
    float4* a; CK(hipMalloc(&a, 1 << 20)); CK(hipMemset(a, 0, 1 << 20));
    for (int wgs : {64, 192, 256, 384, 768}) {
        float ms = time_ms(s, 2000, [&] { hipLaunchKernelGGL(empty_kernel, dim3(wgs), dim3(256), 0, s, (int*)nullptr); });
        printf("empty kernel, %4d WGs: %.2f us per dependent launch\n", wgs, ms * 1e3);
    }
    for (int wgs : {192, 256, 384}) {
        size_t f4 = WBYTES / 16 / wgs;
        float ms1 = time_ms(s, 1000, [&] { hipLaunchKernelGGL(slice_read, dim3(wgs), dim3(256), 0, s, w, f4, 1, sink); });
        float ms4 = time_ms(s, 300, [&] { hipLaunchKernelGGL(slice_read, dim3(wgs), dim3(256), 0, s, w, f4, 4, sink); });
        float ms16 = time_ms(s, 100, [&] { hipLaunchKernelGGL(slice_read, dim3(wgs), dim3(256), 0, s, w, f4, 16, sink); });
        printf("slice_read 24 MB over %3d WGs: 1 pass/launch %.2f us (%.2f TB/s) | 4 passes %.2f us | 16 passes %.2f us -> warm pass %.2f us (%.2f TB/s)\n",
               wgs, ms1 * 1e3, WBYTES / ms1 / 1e9, ms4 * 1e3, ms16 * 1e3, (ms16 - ms4) / 12 * 1e3, WBYTES / ((ms16 - ms4) / 12) / 1e9);
    }
    for (size_t wb : {(size_t)6 << 20, (size_t)12 << 20}) {
        int wgs = 256; size_t f4 = wb / 16 / wgs;
        float ms1 = time_ms(s, 1000, [&] { hipLaunchKernelGGL(slice_read, dim3(wgs), dim3(256), 0, s, w, f4, 1, sink); });
        printf("slice_read %zu MB over 256 WGs: %.2f us per launch (%.2f TB/s)\n", wb >> 20, ms1 * 1e3, wb / ms1 / 1e9);
    }
    for (size_t kb : {64, 128, 384}) {
        size_t n = kb * 1024 / 16;
        float ms = time_ms(s, 1000, [&] { hipLaunchKernelGGL(shared_read, dim3(256), dim3(256), 0, s, a, n, sink); });
        printf("shared_read: 256 WGs each read the same %zu KB: %.2f us per launch\n", kb, ms * 1e3);
    }
    for (int rot : {0, 64, 512, 1031}) {
        size_t n = 128 * 1024 / 16;
        float ms = time_ms(s, 1000, [&] { hipLaunchKernelGGL(shared_read_rot, dim3(256), dim3(256), 0, s, a, n, rot, sink); });
        printf("shared_read_rot (128 KB, rot %4d float4 per WG): %.2f us per launch\n", rot, ms * 1e3);
    }
    for (int wgs : {64, 192, 256}) for (int rot : {0, 1, 3}) {
        float ms = time_ms(s, 1000, [&] { hipLaunchKernelGGL(frag_read, dim3(wgs), dim3(256), 0, s, (const float*)a, 1024, rot, sink); });
        printf("frag_read [32 x 1024] f32 = 128 KB by %3d WGs, rot %d: %.2f us per launch\n", wgs, rot, ms * 1e3);
    }
    {
        const int n = 20000;
        float ms = time_ms(s, 20, [&] { hipLaunchKernelGGL(mfma_chain, dim3(256), dim3(256), 0, s, n, sink); });
        // per wave: 2n MFMAs of 32 cycles, one wave per SIMD
        printf("mfma_chain: %.1f us for %d MFMAs/wave -> %.2f GHz effective (32 cyc each)\n", ms * 1e3, 2 * n, 2.0 * n * 32 / (ms * 1e-3) / 1e9);
    }
    return 0;
}
