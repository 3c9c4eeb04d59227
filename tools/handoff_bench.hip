// Dev tool: cost of the per-step all-gather hand-off a persistent LSTM kernel would need.
// 3 "layers" x 64 workgroups (one per CU, 512 threads).  Per step and per half-batch pipeline each WG
//  (1) waits for its producers' arrival counters (own layer: step-1, layer below: step),
//  (2) reads the two 32 KB packed panels (x from the layer below, h from its own layer) -- sc1 loads,
//  (3) writes its 8 units x 16 rows into the next h / x panels -- sc1 (write-through) stores,
//  (4) drains and bumps its layer's counter (8 shards).
// No MFMA work: this is the pure synchronisation + data-movement floor, in us per step.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int L = 3, NWG = 64, NH = 2, SH = 8, PANEL_F = 16 * 512;   // floats per half-batch panel (16 rows x 512)

struct Args { float* xp; float* hp; unsigned* cnt; unsigned* err; int T; int mode; float* sink; };
// panels: xp[l][slot][half][PANEL_F], hp likewise.  mode bit0: sc1 loads (else acquire fence + plain loads)
__global__ __launch_bounds__(512) void k(Args a) {
    extern __shared__ float lds[];
    const int l = blockIdx.y, ub = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned shard = ub & (SH - 1), per_shard = NWG / SH;
    auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc(a.xp, 0, L * 2 * NH * PANEL_F * 4, 0x00020000);
    auto rsrc_h = __builtin_amdgcn_make_buffer_rsrc(a.hp, 0, L * 2 * NH * PANEL_F * 4, 0x00020000);
    float acc = 0.f;
    for (int t = 0; t < a.T; ++t) {
        const int d = t + l, slot = d & 1;
        for (int hf = 0; hf < NH; ++hf) {
            if (wave == 0) {      // one wave polls: lanes 0-7 own layer, 8-15 layer below, 16-23 layer above (back-pressure)
                const int which = lane >> 3, sh = lane & 7;
                const unsigned* c = a.cnt; unsigned target = 0; bool need = false;
                if (which == 0) { c = a.cnt + ((l * NH + hf) * SH + sh); target = t * per_shard; need = t > 0; }
                else if (which == 1 && l > 0) { c = a.cnt + (((l - 1) * NH + hf) * SH + sh); target = (t + 1) * per_shard; need = true; }
                else if (which == 2 && l + 1 < L && t >= 2) { c = a.cnt + (((l + 1) * NH + hf) * SH + sh); target = (t - 1) * per_shard; need = true; }
                unsigned spins = 0;
                while (true) {
                    unsigned v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bool ok = !need || lane >= 24 || v >= target;
                    if (__all(ok)) break;
                    if (++spins > 2000000u) { if (lane == 0) *a.err = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!(a.mode & 1)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            // read both panels: 64 KB = 4096 float4 over 512 threads = 8 loads each
            const unsigned xoff = ((l * 2 + slot) * NH + hf) * PANEL_F * 4, hoff = xoff;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned o = (q * 512 + tid) * 16;
                if (a.mode & 1) {
                    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, o, xoff, 16);
                    u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc_h, o, hoff, 16);
                    acc += __uint_as_float(v[0]) + __uint_as_float(w[3]);
                } else {
                    float4 v = *reinterpret_cast<const float4*>(a.xp + xoff / 4 + (q * 512 + tid) * 4);
                    float4 w = *reinterpret_cast<const float4*>(a.hp + hoff / 4 + (q * 512 + tid) * 4);
                    acc += v.x + w.w;
                }
            }
            // write this WG's 16 rows x 8 units into the next-slot panels (128 threads, sc1 stores)
            if (tid < 128) {
                const int row = tid >> 3, u = ub * 8 + (tid & 7);
                const size_t po = ((size_t)(u >> 4) * 64 + (((u >> 2) & 3) * 16 + row)) * 4 + (u & 3);
                __hip_atomic_store(a.hp + ((l * 2 + (slot ^ 1)) * NH + hf) * PANEL_F + po, acc * 1e-9f + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (l + 1 < L)
                    __hip_atomic_store(a.xp + (((l + 1) * 2 + (slot ^ 1)) * NH + hf) * PANEL_F + po, acc * 1e-9f + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(a.cnt + ((l * NH + hf) * SH + shard), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == 12345.678f) a.sink[0] = acc;
}

int main(int argc, char** argv) {
    Args a; a.T = 1000;
    const size_t pf = (size_t)L * 2 * NH * PANEL_F;
    CK(hipMalloc(&a.xp, pf * 4)); CK(hipMalloc(&a.hp, pf * 4)); CK(hipMemset(a.xp, 0, pf * 4)); CK(hipMemset(a.hp, 0, pf * 4));
    CK(hipMalloc(&a.cnt, 4096)); CK(hipMalloc(&a.err, 4)); CK(hipMalloc(&a.sink, 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024));
    for (int mode : {1, 0}) {
        a.mode = mode;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(a.cnt, 0, 4096)); CK(hipMemset(a.err, 0, 4));
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(NWG, L), dim3(512), 130 * 1024, 0, a);
            hipEventRecord(e1, 0);
            CK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned err; CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
            printf("mode %d (%s): %.2f us per step (both half-batch pipelines), err=%u\n", mode,
                   mode & 1 ? "sc1 loads" : "acquire fence + plain loads", ms * 1e3 / a.T, err);
        }
    }
    return 0;
}
