// Dev tool: can a SIMD run f32 MFMAs (one wave) and f32 VALU FMAs with DPP row rotation (another wave)
// concurrently?  8 waves per workgroup, one workgroup per CU.  Work unit = one 16x16 tile x 16 k
// ("tile-chunk"): 4 v_mfma_f32_16x16x4_f32, or 64 v_fmac_f32 with a DPP row_ror on the B operand.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int R> __device__ __forceinline__ float rot(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + R, 0xF, 0xF, true));
}
template <> __device__ __forceinline__ float rot<0>(float v) { return v; }

// acc[R] += row_ror_R(b) * a  as ONE v_fmac_f32 with a DPP source (hipcc does not fold the rotation itself)
#define FMAC_DPP(ACC, B, A, R) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:" #R " row_mask:0xf bank_mask:0xf" : "+v"(ACC) : "v"(B), "v"(A))
__device__ __forceinline__ void valu_chunk(float (&acc)[16], const f32x4 a, const f32x4 b) {
#define ROW(M) acc[0] = fmaf(a[M], b[M], acc[0]); \
    FMAC_DPP(acc[1], b[M], a[M], 1); FMAC_DPP(acc[2], b[M], a[M], 2); FMAC_DPP(acc[3], b[M], a[M], 3); \
    FMAC_DPP(acc[4], b[M], a[M], 4); FMAC_DPP(acc[5], b[M], a[M], 5); FMAC_DPP(acc[6], b[M], a[M], 6); \
    FMAC_DPP(acc[7], b[M], a[M], 7); FMAC_DPP(acc[8], b[M], a[M], 8); FMAC_DPP(acc[9], b[M], a[M], 9); \
    FMAC_DPP(acc[10], b[M], a[M], 10); FMAC_DPP(acc[11], b[M], a[M], 11); FMAC_DPP(acc[12], b[M], a[M], 12); \
    FMAC_DPP(acc[13], b[M], a[M], 13); FMAC_DPP(acc[14], b[M], a[M], 14); FMAC_DPP(acc[15], b[M], a[M], 15);
    ROW(0) ROW(1) ROW(2) ROW(3)
#undef ROW
}

// mode 0: all 8 waves MFMA (n chunks each).  mode 1: waves 0-3 MFMA, 4-7 VALU (n chunks each).  mode 2: all VALU.
__global__ __launch_bounds__(512) void k(int n, int mode, float* sink, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 a = {threadIdx.x * 1e-3f, 1.f, 2.f, 3.f}, b = {1.0f + blockIdx.x, 0.5f, 0.25f, 0.125f};
    f32x4 m0 = {0, 0, 0, 0}, m1 = {0, 0, 0, 0};
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool use_valu = mode == 2 || (mode == 1 && wave >= 4);
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (use_valu) {
        for (int i = 0; i < n; ++i) { valu_chunk(acc, a, b); a[0] += 1e-7f; b[0] += 1e-7f; b[1] -= 1e-7f; b[2] += 2e-7f; b[3] -= 2e-7f; }
    } else {
        for (int i = 0; i < n; ++i) {
            m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], m1, 0, 0, 0);
            m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], m1, 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float s = m0[0] + m1[0];
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.f) sink[0] = s;
}
int main() {
    float* sink; hipMalloc(&sink, 4);
    unsigned long long* cyc; hipMalloc(&cyc, 8 * 8 * 256);
    unsigned long long h[8 * 256];
    const int n = 32;    // tile-chunks per wave (= 128 MFMAs or 2048 DPP-FMAs)
    for (int mode : {0, 1, 2}) {
        for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k, dim3(192), dim3(512), 0, 0, n, mode, sink, cyc); hipDeviceSynchronize(); }
        hipMemcpy(h, cyc, 8 * 8 * 192, hipMemcpyDeviceToHost);
        unsigned long long mx = 0, mxm = 0, mxv = 0;
        for (int i = 0; i < 192 * 8; ++i) { mx = h[i] > mx ? h[i] : mx; if ((i & 7) < 4) mxm = h[i] > mxm ? h[i] : mxm; else mxv = h[i] > mxv ? h[i] : mxv; }
        printf("mode %d: %d tile-chunks per wave, 8 waves: slowest wave %llu cycles (waves 0-3: %llu, waves 4-7: %llu); all-MFMA ideal %d\n",
               mode, n, mx, mxm, mxv, n * 4 * 32 * 2);
    }
    return 0;
}
