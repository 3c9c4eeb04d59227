// f32 MFMA GEMM for the batched (time-independent) products of the hot path:
// input/output Linear forward+backward and the dK = [X;Hprev]^T . dG weight
// gradients.  C[M,N] (+)= op(A)[M,K] . op(B)[K,N] (+ bias[N]).
//
// gfx950 design: 128x128x16 block tile, 4 waves (2x2), each wave a 64x64
// sub-tile = 2x2 v_mfma_f32_32x32x2_f32 accumulators (exact f32, 157 TF peak);
// LDS tiles are stored K-major ([k][m], [k][n]) so that the A/B fragment of the
// 32x32x2 MFMA (lane l -> row l&31, k = l>>5) is one conflict-free ds_read_b32
// per lane; global loads are 2x float4 per thread per operand, register-staged
// one K-tile ahead of the MFMAs; split-K (+ f32 atomics) fills the 256 CUs when
// M*N is small and K is the 32k-frame axis.
#include "gemm_core.h"

namespace amdspeech {

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[2][2][BK * LDS_LD];  // [buf][A|B]
    if (g.gate != nullptr) {
        if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            bool late = false;
            while (__hip_atomic_load(g.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > g.gate_need) {
                if (wall_clock64() - t0 > g.gate_limit) { late = true; break; }
                __builtin_amdgcn_s_sleep(32);
            }
            if (late && g.gate_err != nullptr) atomicOr(g.gate_err, 4u);
        }
        __syncthreads();
    }
    WorkgroupBarrier bar;
    gemm_tile<A_KC, B_KC>(g, blockIdx.x, blockIdx.z, &smem[0][0][0], threadIdx.x, 0, true, bar);
}

__global__ void fill_strided_kernel(float* C, int M, int N, int ldc, float v) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)M * N) C[(i / N) * ldc + (i % N)] = v;
}

int gemm_f32(hipStream_t s, bool transA, bool transB, int M, int N, int K, const float* A, int lda,
             const float* B, int ldb, float* C, int ldc, const float* bias, bool accumulate, float* colsum,
             const int* gate, int gate_need, unsigned* gate_err) {
    AS_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: non-positive shape %d %d %d", M, N, K);
    AS_CHECK_ARG(A && B && C, "gemm: null operand");
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.colsum = colsum;
    g.gate = gate; g.gate_need = gate_need; g.gate_limit = 300000000ull;    // 3 s
    g.gate_err = gate_err;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    const int tiles_m = ceil_div(M, BM), tiles_n = ceil_div(N, BN);
    g.tiles_n = tiles_n;
    const int tiles = tiles_m * tiles_n;
    // split K until there are >= ~2 workgroups per CU, keeping >= 16 K-tiles per split
    int splits = 1;
    if (tiles < 512) {
        splits = ceil_div(512, tiles);
        const int max_splits = K / (BK * 16) > 0 ? K / (BK * 16) : 1;
        if (splits > max_splits) splits = max_splits;
    }
    g.k_chunk = ceil_div(ceil_div(K, splits), BK) * BK;
    splits = ceil_div(K, g.k_chunk);
    g.atomic = (accumulate || splits > 1) ? 1 : 0;
    g.a_vec = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0);
    g.b_vec = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0);
    if (!accumulate && splits > 1) {
        const long n = (long)M * N;
        hipLaunchKernelGGL(fill_strided_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, C, M, N, ldc, 0.0f);
    }
    dim3 grid(tiles, 1, splits), block(256);
    // A "KC" = k contiguous = NOT transposed storage [M,K]; B "KC" = stored [N,K] = transposed.
    if (!transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, s, g);
    else if (!transA && transB) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, s, g);
    else if (transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, s, g);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

// out[c] += sum_r x[r*ld + c]   (bias gradients).  One block per 64 columns,
// 4 waves stride the rows, LDS reduce, one atomic per column.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int rows, int cols,
                                                     int ld, float* __restrict__ out, int rows_per_block) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    float acc = 0.0f;
    if (c < cols)
        for (int r = r0 + w; r < r1; r += 4) acc += x[(size_t)r * ld + c];
    red[w][threadIdx.x & 63] = acc;
    __syncthreads();
    if (w == 0 && c < cols) {
        const int l = threadIdx.x;
        unsafeAtomicAdd(out + c, red[0][l] + red[1][l] + red[2][l] + red[3][l]);
    }
}

int colsum_accumulate(hipStream_t s, const float* x, int rows, int cols, int ld, float* out) {
    AS_CHECK_ARG(x && out && rows > 0 && cols > 0, "colsum: bad arguments");
    const int col_blocks = ceil_div(cols, 64);
    int row_blocks = ceil_div(1024, col_blocks);
    if (row_blocks > ceil_div(rows, 64)) row_blocks = ceil_div(rows, 64);
    if (row_blocks < 1) row_blocks = 1;
    const int rpb = ceil_div(rows, row_blocks);
    row_blocks = ceil_div(rows, rpb);
    hipLaunchKernelGGL(colsum_kernel, dim3(col_blocks, row_blocks), dim3(256), 0, s, x, rows, cols, ld, out, rpb);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

}  // namespace amdspeech
