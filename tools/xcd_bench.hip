// Dev tool: hand-off floor when the producers and consumers of a recurrence group share ONE XCD, so the
// loop-carried data only has to reach that XCD's L2 (coherent for all of its CUs) instead of memory.
// 256 workgroups (one per CU); each reads its XCC_ID and takes a ticket inside that XCD -> 8 groups of 32.
// Per step every workgroup of a group (re)loads the group's whole 32 KB panel (16 rows x 512 units) with sc0
// loads (L1 bypass, served by the local L2) until no sentinel is left, then writes its own 16 units x 16 rows
// with plain stores (write-through L1 -> L2).   xcd_bench <T> <mfma_per_wave> <mode: 0 = L2-local, 1 = sc1 (memory)>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int H = 512, ROWS = 16, PANEL = ROWS * H, NG = 8, GW = 32;        // floats per panel; groups; WGs per group
constexpr unsigned SENT = 0x7FC0DEADu;

struct Args { unsigned long long* tl; unsigned* stats; float* hp; unsigned* tickets; unsigned* err; unsigned* layout; int T; int nmfma; int mode; float* sink; unsigned long long bytes; };
// hp[g][t+1][PANEL]

__device__ __forceinline__ bool has_sentinel(const u32x4 v) { return v[0] == SENT || v[1] == SENT || v[2] == SENT || v[3] == SENT; }

template <int AUX, int ST = 0, int SPEC = 0>
__global__ __launch_bounds__(512) void k(Args a) {
    __shared__ float red[8][64];
    __shared__ unsigned s_ticket;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xF;
    if (tid == 0) s_ticket = atomicAdd(a.tickets + xcc, 1u);
    __syncthreads();
    const int g = (int)xcc, ub = (int)s_ticket;
    if (tid == 0) a.layout[blockIdx.x] = (xcc << 8) | s_ticket;
    if (g >= NG || ub >= GW) { if (tid == 0) atomicOr(a.err, 2u); return; }
    const auto rh = __builtin_amdgcn_make_buffer_rsrc(a.hp, 0, (unsigned)a.bytes, 0x00020000);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float fa = 1.0f + lane, fb = 0.5f;
    const unsigned long long t_start = wall_clock64();
    if (SPEC) {
        // wave-specialised variant (as lstm_fwd_flow): waves 0-3 poll the whole panel (8 float4 per lane), run their
        // MFMAs and store; waves 4-7 only run MFMAs, starting right after the barrier (i.e. under the others' poll)
        const int uw = __builtin_amdgcn_readfirstlane(wave);
        unsigned nspin = 0, tfirst = 0, tretry = 0, trest = 0; unsigned long long cend = wall_clock64();
        for (int t = 0; t < a.T; ++t) {
            float s = 0.f;
            if (uw >= 4) {
                if (SPEC == 13) {           // x waves idle: is the poll slow without any partner MFMAs?
                } else
                if (SPEC == 10) {           // VALU work of the same duration instead of MFMAs
                    float v0 = lane, v1 = lane + 1;
                    for (int i = 0; i < a.nmfma * 4; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1" : "+v"(v0), "+v"(v1));
                    s += (v0 + v1) * 1e-30f;
                } else if (SPEC == 11 || SPEC == 12) {    // the MFMA burst starts 0.6 / 1.2 us after the barrier
                    for (int i = 0; i < (SPEC == 11 ? 20 : 40); ++i) __builtin_amdgcn_s_sleep(1);
                    for (int i = 0; i < a.nmfma; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc, 0, 0, 0);
                } else
                if (SPEC >= 7) {
                    // leave the SIMD's issue port to the polling wave between MFMAs: a wave whose next MFMA waits for the
                    // pipe blocks the other wave's VALU instructions
                    f32x4 acc2 = acc, acc3 = acc, acc4 = acc;
                    for (int i = 0; i < a.nmfma; i += 4) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc, 0, 0, 0);
                        if (SPEC == 7) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); else if (SPEC == 8) asm volatile("s_nop 15" ::: "memory"); else if (SPEC == 9) asm volatile("s_sleep 1" ::: "memory");
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc2, 0, 0, 0);
                        if (SPEC == 7) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); else if (SPEC == 8) asm volatile("s_nop 15" ::: "memory"); else if (SPEC == 9) asm volatile("s_sleep 1" ::: "memory");
                        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc3, 0, 0, 0);
                        if (SPEC == 7) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); else if (SPEC == 8) asm volatile("s_nop 15" ::: "memory"); else if (SPEC == 9) asm volatile("s_sleep 1" ::: "memory");
                        acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc4, 0, 0, 0);
                        if (SPEC == 7) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); else if (SPEC == 8) asm volatile("s_nop 15" ::: "memory"); else if (SPEC == 9) asm volatile("s_sleep 1" ::: "memory");
                    }
                    acc += acc2 + acc3 + acc4;
                } else
                for (int i = 0; i < a.nmfma; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc, 0, 0, 0);
                if (SPEC == 2) {
                    // plus the x waves' memory traffic: 8 sc1 loads per lane from a far slot, 9 scattered stores
                    const unsigned fb2 = (unsigned)(((size_t)((g + 4) % NG) * (a.T + 1) + (t + 7) % a.T) * PANEL * 4);
                    u32x4 w[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) w[q] = __builtin_amdgcn_raw_buffer_load_b128(rh, (unsigned)(((wave - 4) * 512 + q * 64 + lane) * 16), fb2, 16);
#pragma unroll
                    for (int q = 0; q < 8; ++q) s += __uint_as_float(w[q][1]) * 1e-30f;
                }
            } else {
                const unsigned base = (unsigned)(((size_t)g * (a.T + 1) + t) * PANEL * 4);
                trest += (unsigned)(wall_clock64() - cend);
                if (t == 500 && tid == 0) a.tl[blockIdx.x * 8 + 0] = wall_clock64();      // poll start
                u32x4 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rh, (unsigned)((wave * 512 + q * 64 + lane) * 16), base, AUX);
                unsigned long long c0 = wall_clock64();
                if (t == 500 && tid == 0) a.tl[blockIdx.x * 8 + 7] = c0;                         // all 8 loads issued
                bool first = true;
                while (true) {
                    bool again = false;
#pragma unroll
                    for (int q = 0; q < 8; ++q) again = again || has_sentinel(v[q]);
                    const unsigned long long c1 = wall_clock64();
                    if (first) tfirst += (unsigned)(c1 - c0); else tretry += (unsigned)(c1 - c0);
                    if (first && t == 500 && tid == 0) a.tl[blockIdx.x * 8 + 1] = c1;            // first round back
                    first = false; c0 = c1;
                    if (!__any(again)) break;
                    ++nspin;
                    if (wall_clock64() - t_start > 100000000ull) { if (lane == 0) atomicOr(a.err, 1u); break; }
                    if (SPEC == 4) asm volatile("buffer_inv sc0" ::: "memory");
                    if (SPEC == 5) asm volatile("buffer_inv sc1" ::: "memory");
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (SPEC == 3 || SPEC >= 7 || (SPEC == 6 ? __any(has_sentinel(v[q])) : has_sentinel(v[q]))) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rh, (unsigned)((wave * 512 + q * 64 + lane) * 16), base, AUX);
                }
                cend = wall_clock64();
                if (t == 500 && tid == 0) a.tl[blockIdx.x * 8 + 2] = cend;                        // wave 0 has everything
#pragma unroll
                for (int q = 0; q < 8; ++q) s += __uint_as_float(v[q][0]);
                for (int i = 0; i < a.nmfma; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc, 0, 0, 0);
            }
            red[wave][lane] = s + acc[0] * 1e-30f;
            __syncthreads();
            if (tid < 256) {
                const int row = tid >> 4, u = ub * 16 + (tid & 15);
                const size_t po = ((size_t)(u >> 2) * 16 + row) * 4 + (u & 3);
                float val = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) val += red[w][tid & 63];
                val = val * 1e-6f + 1.0f;
                float* dst = a.hp + ((size_t)g * (a.T + 1) + t + 1) * PANEL + po;
                if (t == 500 && tid == 0) a.tl[blockIdx.x * 8 + 3] = wall_clock64();              // all 4 waves through the barrier: store issue
                if (t == 499 && tid == 0) a.tl[blockIdx.x * 8 + 5] = wall_clock64();              // the store this step's poll waits for
                if (ST == 0) __hip_atomic_store(dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (ST == 1) __hip_atomic_store(dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (ST == 2) asm volatile("global_store_dword %0, %1, off nt" :: "v"(dst), "v"(val) : "memory");
                else if (ST == 3) asm volatile("global_store_dword %0, %1, off sc0" :: "v"(dst), "v"(val) : "memory");
                else if (ST == 4) asm volatile("global_atomic_swap %0, %1, off" :: "v"(dst), "v"(val) : "memory");
                else if (ST == 5) { __hip_atomic_store(dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_wbl2 sc0" ::: "memory"); }
                else if (ST == 6) asm volatile("global_store_dword %0, %1, off sc0 nt" :: "v"(dst), "v"(val) : "memory");
            }
            __syncthreads();
            if (t == 500 && tid == 0) a.tl[blockIdx.x * 8 + 4] = wall_clock64();                  // store acknowledged + barrier
            if (t == 499 && tid == 0) a.tl[blockIdx.x * 8 + 6] = wall_clock64();
        }
        if (acc[1] == 12345.678f) a.sink[0] = acc[1];
        if (blockIdx.x == 11 && tid == 0) { a.stats[0] = nspin; a.stats[1] = tfirst; a.stats[2] = tretry; a.stats[3] = trest; }
        return;
    }
    for (int t = 0; t < a.T; ++t) {
        // wave w owns 1/8 of the panel: 4 KB = 256 float4 = 4 per lane
        const unsigned base = (unsigned)(((size_t)g * (a.T + 1) + t) * PANEL * 4);
        u32x4 v[4];
        int spins = 0;
        while (true) {
            if (AUX == 100) asm volatile("buffer_inv sc1" ::: "memory");
            if (AUX == 101) asm volatile("buffer_inv sc0" ::: "memory");
            if (AUX == 102) {
                unsigned long long* p = reinterpret_cast<unsigned long long*>(a.hp) + (size_t)base / 8;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const size_t o = ((size_t)(wave * 256 + q * 64 + lane)) * 2;
                    const unsigned long long lo = __hip_atomic_fetch_or(p + o, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const unsigned long long hi = __hip_atomic_fetch_or(p + o + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    v[q][0] = (unsigned)lo; v[q][1] = (unsigned)(lo >> 32); v[q][2] = (unsigned)hi; v[q][3] = (unsigned)(hi >> 32);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    v[q] = __builtin_amdgcn_raw_buffer_load_b128(rh, (unsigned)((wave * 256 + q * 64 + lane) * 16), base, AUX >= 100 ? 0 : AUX);
            }
            bool again = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) again = again || has_sentinel(v[q]);
            if (!__any(again)) break;
            if (++spins > 400000 || wall_clock64() - t_start > 100000000ull) { if (lane == 0) atomicOr(a.err, 1u); break; }
        }
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) s += __uint_as_float(v[q][0]);
        for (int i = 0; i < a.nmfma; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc, 0, 0, 0);
        red[wave][lane] = s + acc[0] * 1e-30f;
        __syncthreads();
        if (tid < 256) {
            // this workgroup's 16 units x 16 rows
            const int row = tid >> 4, u = ub * 16 + (tid & 15);
            const size_t po = ((size_t)(u >> 2) * 16 + row) * 4 + (u & 3);
            float val = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) val += red[w][tid & 63];
            val = val * 1e-6f + 1.0f;
            float* dst = a.hp + ((size_t)g * (a.T + 1) + t + 1) * PANEL + po;
            if (AUX == 16 || ST == 1) __hip_atomic_store(dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
    }
    if (acc[1] == 12345.678f) a.sink[0] = acc[1];
}

int main(int argc, char** argv) {
    Args a;
    a.T = argc > 1 ? atoi(argv[1]) : 1000;
    a.nmfma = argc > 2 ? atoi(argv[2]) : 0;
    a.mode = argc > 3 ? atoi(argv[3]) : 0;
    const size_t n = (size_t)NG * (a.T + 1) * PANEL;
    a.bytes = n * 4;
    CK(hipMalloc(&a.hp, n * 4)); CK(hipMalloc(&a.tickets, 64)); CK(hipMalloc(&a.err, 4)); CK(hipMalloc(&a.sink, 4)); CK(hipMalloc(&a.layout, 256 * 4)); CK(hipMalloc(&a.stats, 64)); CK(hipMalloc(&a.tl, 256 * 8 * 8)); CK(hipMemset(a.tl, 0, 256 * 8 * 8)); CK(hipMemset(a.stats, 0, 64));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(a.hp), SENT, n));
        for (int g = 0; g < NG; ++g)
            CK(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(a.hp + (size_t)g * (a.T + 1) * PANEL), 0u, PANEL));
        CK(hipMemset(a.tickets, 0, 64)); CK(hipMemset(a.err, 0, 4));
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        if (a.mode == 0) hipLaunchKernelGGL((k<1>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 1) hipLaunchKernelGGL((k<16>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 2) hipLaunchKernelGGL((k<100>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 3) hipLaunchKernelGGL((k<101>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 4) hipLaunchKernelGGL((k<102>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 5) hipLaunchKernelGGL((k<2>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 6) hipLaunchKernelGGL((k<3>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 8) hipLaunchKernelGGL((k<2, 0, 1>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 9) hipLaunchKernelGGL((k<2, 0, 2>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 10) hipLaunchKernelGGL((k<2, 0, 3>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 11) hipLaunchKernelGGL((k<2, 0, 4>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 12) hipLaunchKernelGGL((k<3, 0, 1>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 13) hipLaunchKernelGGL((k<1, 0, 1>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 14) hipLaunchKernelGGL((k<0, 0, 4>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 16) hipLaunchKernelGGL((k<16, 0, 1>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 17) hipLaunchKernelGGL((k<17, 0, 1>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 18) hipLaunchKernelGGL((k<18, 0, 1>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 19) hipLaunchKernelGGL((k<19, 0, 1>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 21) hipLaunchKernelGGL((k<2, 1, 1>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 22) hipLaunchKernelGGL((k<2, 2, 1>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 23) hipLaunchKernelGGL((k<2, 3, 1>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 24) hipLaunchKernelGGL((k<2, 4, 1>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 25) hipLaunchKernelGGL((k<2, 5, 1>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 26) hipLaunchKernelGGL((k<2, 6, 1>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 27) hipLaunchKernelGGL((k<2, 0, 6>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 30) hipLaunchKernelGGL((k<2, 0, 7>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 31) hipLaunchKernelGGL((k<2, 0, 8>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 32) hipLaunchKernelGGL((k<2, 0, 9>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 33) hipLaunchKernelGGL((k<2, 0, 10>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 34) hipLaunchKernelGGL((k<2, 0, 11>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 35) hipLaunchKernelGGL((k<2, 0, 12>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 36) hipLaunchKernelGGL((k<2, 0, 13>), dim3(NG * GW), dim3(512), 0, 0, a);
        else if (a.mode == 15) hipLaunchKernelGGL((k<2, 0, 5>), dim3(NG * GW), dim3(512), 0, 0, a);
        else hipLaunchKernelGGL((k<2, 1>), dim3(NG * GW), dim3(512), 0, 0, a);
        hipEventRecord(e1, 0);
        CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned err; CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
        unsigned lay[256]; CK(hipMemcpy(lay, a.layout, sizeof(lay), hipMemcpyDeviceToHost));
        unsigned st[4]; CK(hipMemcpy(st, a.stats, 16, hipMemcpyDeviceToHost));
        printf("   retry rounds per step (one wave): %.2f; first round %.2f us, retries %.2f us, rest of the step %.2f us\n", (double)st[0] / a.T, st[1] / 100.0 / a.T, st[2] / 100.0 / a.T, st[3] / 100.0 / a.T);
        if (rep == 2 && a.mode >= 8 && getenv("TIMELINE")) {
            static unsigned long long tl[256 * 8]; CK(hipMemcpy(tl, a.tl, sizeof(tl), hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull;
            for (int i = 0; i < 256; ++i) if ((lay[i] >> 8) == 0 && tl[i * 8 + 5] && tl[i * 8 + 5] < t0) t0 = tl[i * 8 + 5];
            printf("   group 0, step 500 (us relative to the earliest store issue of step 499): store499 issue, acked+barrier | poll start, first round back, all there | store500 issue, acked\n");
            for (int i = 0; i < 256; ++i) if ((lay[i] >> 8) == 0)
                printf("   wg %3d ub %2u: %6.2f %6.2f | %6.2f (issued %6.2f) %6.2f %6.2f | %6.2f %6.2f\n", i, lay[i] & 255, (tl[i*8+5]-t0)/100.0, (tl[i*8+6]-t0)/100.0, (tl[i*8+0]-t0)/100.0, (tl[i*8+7]-t0)/100.0, (tl[i*8+1]-t0)/100.0, (tl[i*8+2]-t0)/100.0, (tl[i*8+3]-t0)/100.0, (tl[i*8+4]-t0)/100.0);
        }
        int rr = 0;
        for (int i = 0; i < 256; ++i) rr += ((lay[i] >> 8) == (unsigned)(i % 8));
        printf("mode %d (%s) T=%d mfma/wave=%d: %.2f us per step, err=%u, workgroups with xcc == id %% 8: %d/256\n", a.mode,
               a.mode == 0 ? "sc0 loads" : a.mode == 1 ? "sc1 loads / memory" : a.mode == 2 ? "buffer_inv sc1 + plain loads" : a.mode == 3 ? "buffer_inv sc0 + plain loads" : a.mode == 4 ? "returning 64-bit atomics" : a.mode == 5 ? "nt loads" : a.mode == 6 ? "nt sc0 loads" : a.mode == 8 ? "nt loads, wave-specialised" : a.mode == 10 ? "nt, spec, full reload per retry" : a.mode == 11 ? "nt, spec, buffer_inv sc0 per retry" : a.mode == 12 ? "nt sc0, spec" : a.mode == 13 ? "sc0, spec" : a.mode == 14 ? "plain loads, spec, buffer_inv sc0 per retry" : a.mode == 15 ? "nt, spec, buffer_inv sc1 per retry" : a.mode == 36 ? "spec, full reload, x waves idle (h waves: poll, then their MFMAs)" : a.mode == 33 ? "spec, full reload, x waves run VALU instead" : a.mode == 34 ? "spec, full reload, x MFMA burst 0.6 us late" : a.mode == 35 ? "spec, full reload, x MFMA burst 1.2 us late" : a.mode == 30 ? "spec, full reload, x MFMAs spaced by s_nop 24" : a.mode == 31 ? "spec, full reload, x MFMAs spaced by s_nop 16" : a.mode == 32 ? "spec, full reload, x MFMAs spaced by s_sleep 1" : a.mode == 27 ? "spec, nt, retry per fragment (whole wave)" : a.mode == 21 ? "spec, nt loads, sc1 stores" : a.mode == 22 ? "spec, nt loads, nt stores" : a.mode == 23 ? "spec, nt loads, sc0 stores" : a.mode == 24 ? "spec, nt loads, atomic-swap stores" : a.mode == 25 ? "spec, nt loads, store + wbl2 sc0" : a.mode == 26 ? "spec, nt loads, sc0 nt stores" : a.mode == 16 ? "sc1 loads / plain stores, spec" : a.mode == 17 ? "sc0 sc1 loads / plain stores, spec" : a.mode == 18 ? "sc1 nt loads / plain stores, spec" : a.mode == 19 ? "sc0 sc1 nt loads / plain stores, spec" : a.mode == 9 ? "nt loads, wave-specialised + x traffic" : "nt loads, sc1 (write-through) stores", a.T, a.nmfma, ms * 1e3 / a.T, err, rr);
    }
    return 0;
}
