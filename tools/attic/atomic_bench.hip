// Dev probe: what does it cost a recurrence group (32 workgroups on ONE XCD) to exchange partial tiles per step
//   mode 0: b128 stores of 32 x 1 KiB tiles per workgroup + b128 gathers of 32 tiles (the Q ring of lstm_bwd_flow2)
//   mode 1: f32 atomic adds (no return) of the same 32 KiB into 32 accumulation tiles + a 1 KiB read
//   mode 2: stores only      mode 3: gathers only
// 256 workgroups x 512 threads; a workgroup joins the group of its XCD by ticket.  Prints us per step.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void probe(float* ring, float* acc, unsigned* tickets, int iters, int mode, float* sink, int slots) {
    __shared__ unsigned s_ticket;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xF;
    if (threadIdx.x == 0) s_ticket = atomicAdd(tickets + xcc, 1u);
    __syncthreads();
    const int ub = s_ticket;          // 0..31
    if (ub >= 32) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NU = 32, NTW = 4;
    float* myring = ring + (size_t)xcc * slots * NU * NU * 256;
    float* myacc = acc + (size_t)xcc * slots * NU * 256;
    f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    f32x4 total = {0.f, 0.f, 0.f, 0.f};
    int slot = 0;
    for (int it = 0; it < iters; ++it) {
        if (mode == 0 || mode == 2) {
#pragma unroll
            for (int n = 0; n < NTW; ++n) {          // consumer wave*NTW + n, producer ub
                float* p = myring + ((size_t)slot * NU * NU + (size_t)(wave * NTW + n) * NU + ub) * 256 + lane * 4;
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
            }
        }
        if (mode == 1) {
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                float* p = myacc + ((size_t)slot * NU + (wave * NTW + n)) * 256 + lane;
#pragma unroll
                for (int j = 0; j < 4; ++j) unsafeAtomicAdd(p + j * 64, v[j]);
            }
            if (wave == 0) total += *reinterpret_cast<const f32x4*>(myacc + ((size_t)((slot + 1) % slots) * NU + ub) * 256 + lane * 4);
        }
        if (mode == 0 || mode == 3) {
#pragma unroll
            for (int q = 0; q < NTW; ++q) {          // consumer ub, producer wave*NTW + q
                const float* p = myring + ((size_t)((slot + 1) % slots) * NU * NU + (size_t)ub * NU + wave * NTW + q) * 256 + lane * 4;
                total += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
            }
        }
        v += total * 1e-30f;
        // some MFMA-like gap: none.  barrier per step as in the kernel
        __syncthreads();
        if (++slot == slots) slot = 0;
    }
    if (total[0] == 12345.f) sink[0] = total[1];
}

int main(int argc, char** argv) {
    const int slots = argc > 1 ? atoi(argv[1]) : 3;
    float *ring, *acc, *sink; unsigned* tickets;
    hipMalloc(&ring, (size_t)8 * slots * 32 * 32 * 1024);
    hipMalloc(&acc, (size_t)8 * slots * 32 * 1024);
    hipMalloc(&sink, 64); hipMalloc(&tickets, 64);
    hipMemset(ring, 0, (size_t)8 * slots * 32 * 32 * 1024); hipMemset(acc, 0, (size_t)8 * slots * 32 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 4; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(tickets, 0, 64);
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, ring, acc, tickets, iters, mode, sink, slots);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("slots %d mode %d: %.3f us per step\n", slots, mode, ms * 1e3 / iters);
        }
    return 0;
}
