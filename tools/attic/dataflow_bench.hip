// Dev tool: floor of a persistent, dataflow-synchronised LSTM forward -- NO counters, NO flags: every slot
// of the packed panel history is written exactly once per sequence and pre-filled with a NaN sentinel; a
// consumer simply (re)loads the float4s it needs with agent-coherent (sc1) loads until none carries the
// sentinel.  3 layers x 64 workgroups (one per CU, 8 waves, K split over the waves as in lstm_fwd_step).
// Per step a workgroup: [x half: load x_t panel slice, MFMAs] [h half: load h_{t-1} slice (the loop-carried
// dependency), MFMAs] [barrier] [write its 8 units x 32 rows into h_t and (next layer's) x_t, write-through].
//   dataflow_bench <T> <mfma_per_wave_per_half>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int L = 3, NWG = 64, H = 512, B = 32, PANEL = B * H;            // floats per panel (64 KB)
constexpr unsigned SENT = 0x7FC0DEADu;

struct Args { float* xp; float* hp; unsigned* err; int T; int nmfma; int mode; float* sink; unsigned long long xp_bytes, hp_bytes; };
// xp[l][t][PANEL] (l = 0 pre-filled input), hp[l][t+1][PANEL] (slot 0 = initial state, pre-filled)

__device__ __forceinline__ bool has_sentinel(const u32x4 v) {
    return v[0] == SENT || v[1] == SENT || v[2] == SENT || v[3] == SENT;
}

template <int AUX, int SYS_STORE>
__global__ __launch_bounds__(512) void k(Args a) {
    __shared__ float red[8][64];
    const int l = blockIdx.y, ub = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(a.xp, 0, (unsigned)a.xp_bytes, 0x00020000);
    const auto rh = __builtin_amdgcn_make_buffer_rsrc(a.hp, 0, (unsigned)a.hp_bytes, 0x00020000);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float fa = 1.0f + lane, fb = 0.5f;
    const unsigned long long t_start = wall_clock64();
    for (int t = 0; t < a.T; ++t) {
        // wave w owns the K slice [w*64, w*64+64) of each 512-wide half: 32 rows x 64 k = 512 float4 = 8 per lane
        auto poll = [&](decltype(rx) rsrc, unsigned base_bytes) -> float {
            u32x4 v[8];
            unsigned pending = 0xFFu;
            float s = 0.f;
            int spins = 0;
            while (true) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (pending & (1u << q))
                        v[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)((wave * 512 + q * 64 + lane) * 16), base_bytes, AUX);
                unsigned still = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if ((pending & (1u << q)) && has_sentinel(v[q])) still |= 1u << q;
                pending = still;
                if (__all(pending == 0)) break;
                if (++spins > 200000 || wall_clock64() - t_start > 400000000ull) { if (lane == 0) *a.err = 1; return 0.f; }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) s += __uint_as_float(v[q][0]);
            return s;
        };
        const unsigned xo = (unsigned)(((size_t)l * a.T + t) * PANEL * 4);
        const unsigned ho = (unsigned)(((size_t)l * (a.T + 1) + t) * PANEL * 4);
        float s = 0.f;
        if (a.mode & 256) {
            // mode 256: the h loads go out FIRST (speculatively), the x-half MFMAs run under them, then the poll
            // loop re-loads only if a sentinel is still there
            u32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                v[q] = __builtin_amdgcn_raw_buffer_load_b128(rh, (unsigned)((wave * 512 + q * 64 + lane) * 16), ho, AUX);
            __builtin_amdgcn_sched_barrier(0);
            for (int i = 0; i < a.nmfma; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            int spins = 0;
            while (true) {
                bool again = false;
#pragma unroll
                for (int q = 0; q < 8; ++q) again = again || has_sentinel(v[q]);
                if (!__any(again)) break;
                if (++spins > 200000 || wall_clock64() - t_start > 400000000ull) { if (lane == 0) *a.err = 1; break; }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    v[q] = __builtin_amdgcn_raw_buffer_load_b128(rh, (unsigned)((wave * 512 + q * 64 + lane) * 16), ho, AUX);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) s += __uint_as_float(v[q][0]);
        } else {
            s = (a.mode & 6) ? 0.f : poll(rx, xo);           // mode 2: no x panel at all (half the read traffic); 4: MFMA only
            for (int i = 0; i < a.nmfma; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc, 0, 0, 0);
            if (!(a.mode & 4)) s += poll(rh, ho);
        }
        for (int i = 0; i < a.nmfma; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc, 0, 0, 0);
        red[wave][lane] = s + acc[0] * 1e-30f;
        __syncthreads();
        if ((a.mode & 8) && !(a.mode & 4)) {
            // mode 8: the same 8 units x 32 rows as 64 float4 stores (4 consecutive units of one row each)
            if (tid < 64) {
                const int row = tid >> 1, ug = ub * 2 + (tid & 1);            // unit group of 4
                const size_t po = ((size_t)ug * 32 + row) * 4;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) v += red[w][tid & 63];
                v = v * 1e-6f + 1.0f;
                const f32x4 v4 = {v, v, v, v};
                f32x4* ph = reinterpret_cast<f32x4*>(a.hp + ((size_t)l * (a.T + 1) + t + 1) * PANEL + po);
                asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(ph), "v"(v4) : "memory");
                if (l + 1 < L) {
                    f32x4* px = reinterpret_cast<f32x4*>(a.xp + ((size_t)(l + 1) * a.T + t) * PANEL + po);
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(px), "v"(v4) : "memory");
                }
            }
        } else
        if (tid < 256 && !(a.mode & 4)) {
            // this workgroup's 8 units x 32 rows: 256 floats, one per thread, fragment-major position
            const int row = tid >> 3, u = ub * 8 + (tid & 7);
            const size_t po = ((size_t)(u >> 2) * 32 + row) * 4 + (u & 3);
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red[w][tid & 63];
            v = v * 1e-6f + 1.0f;                                   // finite, never the sentinel
            __hip_atomic_store(a.hp + ((size_t)l * (a.T + 1) + t + 1) * PANEL + po, v, __ATOMIC_RELAXED, SYS_STORE ? __HIP_MEMORY_SCOPE_SYSTEM : __HIP_MEMORY_SCOPE_AGENT);
            if (l + 1 < L)
                __hip_atomic_store(a.xp + ((size_t)(l + 1) * a.T + t) * PANEL + po, v, __ATOMIC_RELAXED, SYS_STORE ? __HIP_MEMORY_SCOPE_SYSTEM : __HIP_MEMORY_SCOPE_AGENT);
        }
        if (a.mode & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // expose the store acknowledgement latency
        __syncthreads();
    }
    if (acc[1] == 12345.678f) a.sink[0] = acc[1];
}

int main(int argc, char** argv) {
    Args a;
    a.T = argc > 1 ? atoi(argv[1]) : 1000;
    a.nmfma = argc > 2 ? atoi(argv[2]) : 0;
    a.mode = argc > 3 ? atoi(argv[3]) : 0;
    const size_t xn = (size_t)L * a.T * PANEL, hn = (size_t)L * (a.T + 1) * PANEL;
    a.xp_bytes = xn * 4; a.hp_bytes = hn * 4;
    if (a.xp_bytes >= (1ull << 32) || a.hp_bytes >= (1ull << 32)) { printf("T too large for one buffer descriptor\n"); return 1; }
    CK(hipMalloc(&a.xp, xn * 4)); CK(hipMalloc(&a.hp, hn * 4)); CK(hipMalloc(&a.err, 4)); CK(hipMalloc(&a.sink, 4));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(a.xp), SENT, xn));
        CK(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(a.hp), SENT, hn));
        CK(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(a.xp), 0x3F800000u, (size_t)a.T * PANEL));          // layer-0 inputs
        for (int l = 0; l < L; ++l)
            CK(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(a.hp + (size_t)l * (a.T + 1) * PANEL), 0u, PANEL));   // h_{-1}
        CK(hipMemset(a.err, 0, 4));
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        const int pol = (a.mode >> 4) & 7;
        if (pol == 0) hipLaunchKernelGGL((k<16, 0>), dim3(NWG, L), dim3(512), 0, 0, a);
        else if (pol == 1) hipLaunchKernelGGL((k<17, 0>), dim3(NWG, L), dim3(512), 0, 0, a);
        else if (pol == 2) hipLaunchKernelGGL((k<16, 1>), dim3(NWG, L), dim3(512), 0, 0, a);
        else if (pol == 3) hipLaunchKernelGGL((k<17, 1>), dim3(NWG, L), dim3(512), 0, 0, a);
        else if (pol == 4) hipLaunchKernelGGL((k<18, 0>), dim3(NWG, L), dim3(512), 0, 0, a);
        else hipLaunchKernelGGL((k<1, 0>), dim3(NWG, L), dim3(512), 0, 0, a);
        hipEventRecord(e1, 0);
        CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned err; CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
        printf("mode %d T=%d mfma/half/wave=%d: %.2f us per step (%.2f ms total), err=%u\n", a.mode, a.T, a.nmfma, ms * 1e3 / (a.T + L - 1), ms, err);
    }
    return 0;
}
