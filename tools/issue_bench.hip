// Dev tool: how a wave that streams MFMAs shares its SIMD with a co-resident wave's VALU work (gfx950).
// One workgroup of 8 waves per CU (2 per SIMD): waves 4-7 run N MFMAs (16x16x4 f32) with a configurable gap
// after each; waves 0-3 run M dependent-free VALU instructions (or nothing).  Prints time per MFMA / per VALU op.
//   issue_bench <gap mode> <valu on/off>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int GAP, int VALU>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int n, float* buf) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned long long t0 = wall_clock64();
    if (wave >= 4) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        const float fa = 1.0f + lane, fb = 0.5f;
        if (GAP == 6) {                      // one accumulator: every MFMA depends on the previous one
            for (int i = 0; i < n; ++i) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, a0, 0, 0, 0);
        } else if (GAP == 7) {               // 4 dependent MFMAs per accumulator, 4 accumulators in turn (lstm_*_flow before the fix)
            for (int i = 0; i < n; i += 16) {
#define DEP4(x) x = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, x, 0, 0, 0); x = __builtin_amdgcn_mfma_f32_16x16x4f32(fb, fa, x, 0, 0, 0); x = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, x, 0, 0, 0); x = __builtin_amdgcn_mfma_f32_16x16x4f32(fb, fa, x, 0, 0, 0);
                DEP4(a0) DEP4(a1) DEP4(a2) DEP4(a3)
            }
        } else
        for (int i = 0; i < n; i += 4) {
#define GAPI() do { if (GAP == 1) asm volatile("s_nop 7"); else if (GAP == 2) asm volatile("s_nop 15"); else if (GAP == 3) asm volatile("s_nop 15\n\ts_nop 7"); else if (GAP == 4) asm volatile("s_sleep 1"); else if (GAP == 5) asm volatile("s_nop 3"); } while (0)
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, a0, 0, 0, 0); GAPI();
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, a1, 0, 0, 0); GAPI();
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, a2, 0, 0, 0); GAPI();
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, a3, 0, 0, 0); GAPI();
        }
        a0 += a1 + a2 + a3;
        if (a0[0] == 123.456f) sink[0] = a0[1];
        if (lane == 0) out[blockIdx.x * 8 + wave] = wall_clock64() - t0;
    } else if (VALU == 3) {
        // a "clock wave": back-to-back wall_clock64() (s_memrealtime + s_waitcnt lgkmcnt(0)), as in a time-limit check
        unsigned long long acc = 0;
        for (int i = 0; i < n; ++i) { acc += wall_clock64(); asm volatile("" : "+s"(acc)); }
        if (acc == 12345ull) sink[1] = 1.0f;
        if (lane == 0) out[blockIdx.x * 8 + wave] = wall_clock64() - t0;
    } else if (VALU == 4) {
        // s_sleep loop
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
        if (lane == 0) out[blockIdx.x * 8 + wave] = wall_clock64() - t0;
    } else if (VALU == 2) {
        // a "load wave": 8 x 16 B per lane from an L2-resident buffer, wait, repeat (the polling pattern of lstm_*_flow)
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const auto r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1u << 20, 0x00020000);
        unsigned sum = 0;
        for (int i = 0; i < n / 16; ++i) {
            u32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)((wave * 512 + q * 64 + lane) * 16), (unsigned)((i & 7) * 32768), 2);
#pragma unroll
            for (int q = 0; q < 8; ++q) sum += v[q][0];
            asm volatile("" : "+v"(sum));
        }
        if (sum == 12345u) sink[1] = 1.0f;
        if (lane == 0) out[blockIdx.x * 8 + wave] = wall_clock64() - t0;
    } else if (VALU) {
        float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3;
        for (int i = 0; i < n; ++i) {          // 8 independent-ish VALU ops per iteration
            asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t"
                         "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        }
        if (v0 + v1 + v2 + v3 == 123.456f) sink[1] = v0;
        if (lane == 0) out[blockIdx.x * 8 + wave] = wall_clock64() - t0;
    }
}
int main(int argc, char** argv) {
    const int gap = argc > 1 ? atoi(argv[1]) : 0, valu = argc > 2 ? atoi(argv[2]) : 1, n = 4096;
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 8 * 8); hipMalloc(&sink, 64); float* buf; hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20); hipMemset(out, 0, 256 * 8 * 8);
    for (int rep = 0; rep < 2; ++rep) {
#define L(G) do { if (valu == 2) hipLaunchKernelGGL((k<G, 2>), dim3(256), dim3(512), 0, 0, out, sink, n, buf); else if (valu == 3) hipLaunchKernelGGL((k<G, 3>), dim3(256), dim3(512), 0, 0, out, sink, n, buf); else if (valu == 4) hipLaunchKernelGGL((k<G, 4>), dim3(256), dim3(512), 0, 0, out, sink, n, buf); else if (valu) hipLaunchKernelGGL((k<G, 1>), dim3(256), dim3(512), 0, 0, out, sink, n, buf); else hipLaunchKernelGGL((k<G, 0>), dim3(256), dim3(512), 0, 0, out, sink, n, buf); } while (0)
        if (gap == 0) L(0); else if (gap == 1) L(1); else if (gap == 2) L(2); else if (gap == 3) L(3); else if (gap == 4) L(4); else if (gap == 6) L(6); else if (gap == 7) L(7); else L(5);
        hipDeviceSynchronize();
    }
    unsigned long long h[8]; hipMemcpy(h, out + 8 * 17, sizeof(h), hipMemcpyDeviceToHost);
    if (valu == 3 || valu == 4) { printf("gap mode %d, %s wave: MFMA wave %.1f ns per MFMA (%.0f us total); %.1f ns per call (%.0f us total)\n", gap, valu == 3 ? "clock" : "sleep", h[4] * 10.0 / n, h[4] / 100.0, h[0] * 10.0 / n, h[0] / 100.0); return 0; }
    if (valu == 2) { printf("gap mode %d, load wave: MFMA wave %.1f ns per MFMA (%.0f us total); load wave %.0f ns per round of 8 x 1 KB (%.0f us total)\n", gap, h[4] * 10.0 / n, h[4] / 100.0, h[0] * 10.0 / (n / 16), h[0] / 100.0); return 0; }
    printf("gap mode %d, VALU wave %s: MFMA wave %.1f ns per MFMA; VALU wave %.2f ns per v_fma (alone: 4 cycles = 1.8 ns)\n", gap, valu ? "on" : "off",
           h[4] * 10.0 / n, valu ? h[0] * 10.0 / (n * 8.0) : 0.0);
    return 0;
}
