// Does the L2 absorb repeated stores to the same lines, or does every store reach the fabric?  (round 4: the partial-tile rings of the
// backward recurrence kernels are rewritten every two steps, yet rocprofv3's WRITE_SIZE equals the bytes STORED.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/l2_writeback_probe tools/l2_writeback_probe.hip
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out -- tools/l2_writeback_probe
// rewrite<SA>: each of 256 workgroups rewrites ITS OWN 4 KiB `reps` times (store policy SA: 0 plain, 2 nt, 16 sc1).
// ring<LA, KB>: the rings' pattern -- every workgroup stores KB KiB (plain) into slot r & 1 of a [2][256][KB KiB] ring and then reads the
// tile of the workgroup 8 further on (same XCD: workgroups are dealt round-robin to the 8 XCDs) with load policy LA (0 plain, 1 sc0, 2 nt,
// 3 sc0|nt, 16 sc1): does the LOAD policy decide whether the dirty lines survive in the L2 until they are overwritten?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int AUX>
__global__ void rewrite(float* buf, int reps, int pause) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(buf + (size_t)blockIdx.x * 1024, 0, 4096u, 0x00020000);
    for (int r = 0; r < reps; ++r) {
        u4 v = {(unsigned)r, (unsigned)r + 1, (unsigned)r + 2, (unsigned)r + 3};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, threadIdx.x * 16u, 0, AUX);
        for (int i = 0; i < pause; ++i) __builtin_amdgcn_s_sleep(8);
    }
}
template <int LA, int KB>
__global__ void ring(float* buf, int reps, unsigned* sink) {
    const unsigned slot_bytes = 256u * KB * 1024u;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 2u * slot_bytes, 0x00020000);
    const unsigned mine = blockIdx.x * (KB * 1024u), other = ((blockIdx.x + 8) & 255) * (KB * 1024u);
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r) {
        const unsigned so = (r & 1) * slot_bytes;
#pragma unroll
        for (int q = 0; q < KB / 4; ++q) {
            u4 v = {(unsigned)r, (unsigned)r + 1, (unsigned)r + 2, (unsigned)r + 3};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, mine + q * 4096u + threadIdx.x * 16u + so, 0, 0);
        }
        for (int i = 0; i < 6; ++i) __builtin_amdgcn_s_sleep(8);
#pragma unroll
        for (int q = 0; q < KB / 4; ++q) {
            const u4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, other + q * 4096u + threadIdx.x * 16u + so, 0, LA);
            acc += w[0] ^ w[3];
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}
// ptiles<LAYOUT>: the P ring's real pattern.  Group g = blockIdx % 8 (one XCD), member m = blockIdx / 8 of 32; per round every wave w
// stores four 1 KiB tiles for consumers 4w .. 4w+3 and gathers the four tiles producers 4w .. 4w+3 addressed to m (nt loads).
// LAYOUT 0: [slot][consumer][producer][1 KiB] (lstm_bwd_flow2); 1: [slot][producer][consumer][1 KiB] (a producer's 32 tiles contiguous).
// EXTRA: 1 = every round also stores 512 bytes per workgroup WRITE-THROUGH (sc1) into a write-once history (the row-major dG rows / dX
// tiles of the real kernel); 2 = also loads 512 bytes with sc1 from a history; 4 = streams 8 KiB of plain loads per workgroup and round
template <int LAYOUT, int EXTRA = 0>
__global__ void ptiles(float* buf, int reps, unsigned* sink, float* hist = nullptr) {
    const int g = blockIdx.x & 7, m = blockIdx.x >> 3, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned slot_bytes = 32u * 32u * 1024u;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(buf + (size_t)g * 2 * (slot_bytes / 4), 0, 2u * slot_bytes, 0x00020000);
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r) {
        const unsigned so = (r & 1) * slot_bytes;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int c = w * 4 + n;
            const unsigned off = (LAYOUT == 0 ? (unsigned)(c * 32 + m) : (unsigned)(m * 32 + c)) * 1024u + lane * 16u;
            u4 v = {(unsigned)r, (unsigned)r + 1, (unsigned)r + 2, (unsigned)r + 3};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, off + so, 0, 0);
        }
        if (EXTRA & 1) {
            if (w == 4 && lane < 32) __hip_atomic_store(hist + ((size_t)r * 256 + blockIdx.x) * 128 + lane * 4, (float)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (EXTRA & 2) {
            if (w == 5 && lane < 32) acc += (unsigned)__hip_atomic_load(hist + ((size_t)r * 256 + ((blockIdx.x + 3) & 255)) * 128 + lane * 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (EXTRA & 4) {
            const float4 v = *reinterpret_cast<const float4*>(hist + ((size_t)r * 256 + blockIdx.x) * 2048 + threadIdx.x * 4);
            acc += (unsigned)v.x;
        }
        if (EXTRA & 8) {        // the same 8 KiB stream as non-temporal loads
            const auto rh = __builtin_amdgcn_make_buffer_rsrc(hist + ((size_t)r * 256 + blockIdx.x) * 2048, 0, 8192u, 0x00020000);
            const u4 x = __builtin_amdgcn_raw_buffer_load_b128(rh, threadIdx.x * 16u, 0, 2);
            acc += x[0];
        }
        if (EXTRA & 48) {       // 4 KiB per workgroup and round into a write-once history: 16 = write-through (sc1), 32 = sc1 | nt
            const auto rh = __builtin_amdgcn_make_buffer_rsrc(hist + ((size_t)r * 256 + blockIdx.x) * 2048, 0, 8192u, 0x00020000);
            u4 v = {(unsigned)r, 1u, 2u, 3u};
            if (w < 4) {
                if (EXTRA & 16) __builtin_amdgcn_raw_buffer_store_b128(v, rh, (threadIdx.x & 255) * 16u, 0, 16);
                else __builtin_amdgcn_raw_buffer_store_b128(v, rh, (threadIdx.x & 255) * 16u, 0, 18);
            }
        }
        if (EXTRA & 64) {       // ... or PLAINLY (visible to other XCDs only when evicted / at the end of the kernel)
            const auto rh = __builtin_amdgcn_make_buffer_rsrc(hist + ((size_t)r * 256 + blockIdx.x) * 2048, 0, 8192u, 0x00020000);
            u4 v = {(unsigned)r, 1u, 2u, 3u};
            if (w < 4) __builtin_amdgcn_raw_buffer_store_b128(v, rh, (threadIdx.x & 255) * 16u, 0, 0);
        }
        for (int i = 0; i < 6; ++i) __builtin_amdgcn_s_sleep(8);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int q = w * 4 + n;
            const unsigned off = (LAYOUT == 0 ? (unsigned)(m * 32 + q) : (unsigned)(q * 32 + m)) * 1024u + lane * 16u;
            const u4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, off + so, 0, 2);
            acc += x[0] ^ x[3];
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}
int main() {
    float* buf; hipMalloc(&buf, 2 * 256 * 32 * 1024);
    unsigned* sink; hipMalloc(&sink, 64);
    const int reps = 2000;
    hipLaunchKernelGGL(rewrite<0>, dim3(256), dim3(256), 0, 0, buf, reps, 0);
    hipLaunchKernelGGL(rewrite<16>, dim3(256), dim3(256), 0, 0, buf, reps, 0);
    // 32 KiB per workgroup and slot = 2 x 1 MiB per XCD: the P ring of lstm_bwd_flow2 at H = 512
    hipLaunchKernelGGL((ring<0, 32>), dim3(256), dim3(256), 0, 0, buf, reps, sink);
    hipLaunchKernelGGL((ring<1, 32>), dim3(256), dim3(256), 0, 0, buf, reps, sink);
    hipLaunchKernelGGL((ring<2, 32>), dim3(256), dim3(256), 0, 0, buf, reps, sink);
    hipLaunchKernelGGL((ring<3, 32>), dim3(256), dim3(256), 0, 0, buf, reps, sink);
    hipLaunchKernelGGL((ring<16, 32>), dim3(256), dim3(256), 0, 0, buf, reps, sink);
    hipLaunchKernelGGL((ring<2, 4>), dim3(256), dim3(256), 0, 0, buf, reps, sink);
    hipLaunchKernelGGL((ptiles<0>), dim3(256), dim3(512), 0, 0, buf, reps, sink);
    hipLaunchKernelGGL((ptiles<1>), dim3(256), dim3(512), 0, 0, buf, reps, sink);
    float* hist; hipMalloc(&hist, (size_t)reps * 256 * 2048 * 4); hipMemset(hist, 0, (size_t)reps * 256 * 2048 * 4);
    hipLaunchKernelGGL((ptiles<0, 1>), dim3(256), dim3(512), 0, 0, buf, reps, sink, hist);      // + sc1 stores
    hipLaunchKernelGGL((ptiles<0, 2>), dim3(256), dim3(512), 0, 0, buf, reps, sink, hist);      // + sc1 loads
    hipLaunchKernelGGL((ptiles<0, 4>), dim3(256), dim3(512), 0, 0, buf, reps, sink, hist);      // + 8 KiB of streaming plain loads
    hipLaunchKernelGGL((ptiles<0, 7>), dim3(256), dim3(512), 0, 0, buf, reps, sink, hist);      // all three
    hipLaunchKernelGGL((ptiles<0, 8>), dim3(256), dim3(512), 0, 0, buf, reps, sink, hist);      // + 8 KiB of streaming nt loads
    hipLaunchKernelGGL((ptiles<0, 16>), dim3(256), dim3(512), 0, 0, buf, reps, sink, hist);     // + 4 KiB of sc1 stores
    hipLaunchKernelGGL((ptiles<0, 32>), dim3(256), dim3(512), 0, 0, buf, reps, sink, hist);     // + 4 KiB of sc1 | nt stores
    hipLaunchKernelGGL((ptiles<0, 64>), dim3(256), dim3(512), 0, 0, buf, reps, sink, hist);     // + 4 KiB of plain stores (write-once)
    hipDeviceSynchronize();
    printf("rewrite: %.1f MB stored per launch over 1 MiB; ring<.,32>: %.1f MB stored over 16 MiB; ring<.,4>: %.1f MB over 2 MiB\n",
           256.0 * 4096 * reps / 1e6, 256.0 * 32768 * reps / 1e6, 256.0 * 4096 * reps / 1e6);
    return 0;
}
