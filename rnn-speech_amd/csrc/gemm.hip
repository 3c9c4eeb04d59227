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
#include "common.h"

namespace amdspeech {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDS_LD = BM + 4;  // +4 floats: breaks the 128-float stride for the transposing writes

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc;
    int k_chunk;      // K range per split (multiple of BK)
    int tiles_n;
    int atomic;       // 1: atomicAdd into C, 0: plain store
    int a_vec, b_vec; // 1: operand rows are 16-byte aligned -> float4 loads
    float* colsum;    // optional: colsum[n] += sum_k B[k][n] (bias gradient), done by the tm == 0 tiles
    const int* gate;  // optional: wait until *gate <= gate_need before touching the operands (a producer kernel
    int gate_need;    //           running concurrently on another CU partition counts *gate down as it finishes rows)
    unsigned long long gate_limit;   // wall_clock64 ticks the wait may last
    unsigned* gate_err;              // bit 2 is raised when the wait times out (the operands are then NOT complete)
};

// One operand tile = 128 "rows" (m or n) x 16 k.
//   KC ("k contiguous"): element (r, k) at P[r*ld + k]
//   MC ("row contiguous"): element (r, k) at P[k*ld + r]
template <bool KC>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int row0, int nrows,
                                          int k0, int kend, bool vec, float (&reg)[8]) {
    const int tid = threadIdx.x;
    if (KC) {
        const int r = row0 + (tid >> 1);
        const int k = k0 + (tid & 1) * 8;
        if (r < nrows && k + 8 <= kend && vec) {
            const float4* p = reinterpret_cast<const float4*>(P + (size_t)r * ld + k);
            float4 v0 = p[0], v1 = p[1];
            reg[0] = v0.x; reg[1] = v0.y; reg[2] = v0.z; reg[3] = v0.w;
            reg[4] = v1.x; reg[5] = v1.y; reg[6] = v1.z; reg[7] = v1.w;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                reg[q] = (r < nrows && k + q < kend) ? P[(size_t)r * ld + k + q] : 0.0f;
        }
    } else {
        const int k = k0 + (tid >> 4);
        const int r = row0 + (tid & 15) * 8;
        if (k < kend && r + 8 <= nrows && vec) {
            const float4* p = reinterpret_cast<const float4*>(P + (size_t)k * ld + r);
            float4 v0 = p[0], v1 = p[1];
            reg[0] = v0.x; reg[1] = v0.y; reg[2] = v0.z; reg[3] = v0.w;
            reg[4] = v1.x; reg[5] = v1.y; reg[6] = v1.z; reg[7] = v1.w;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                reg[q] = (k < kend && r + q < nrows) ? P[(size_t)k * ld + r + q] : 0.0f;
        }
    }
}

template <bool KC>
__device__ __forceinline__ void store_tile(float* __restrict__ S, const float (&reg)[8]) {
    const int tid = threadIdx.x;
    if (KC) {
        const int r = tid >> 1, k = (tid & 1) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) S[(k + q) * LDS_LD + r] = reg[q];
    } else {
        const int k = tid >> 4, r = (tid & 15) * 8;
        float4* p = reinterpret_cast<float4*>(S + k * LDS_LD + r);
        p[0] = make_float4(reg[0], reg[1], reg[2], reg[3]);
        p[1] = make_float4(reg[4], reg[5], reg[6], reg[7]);
    }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[2][2][BK * LDS_LD];  // [buf][A|B]
    const int tile = blockIdx.x;
    const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.z * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
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

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float ra[8], rb[8];
    // bias gradient fused into the weight-gradient GEMM: the first row of tiles also sums its B tile
    // over k (B is dY [K = frames, N]); every B element is visited by exactly one such workgroup
    const bool do_colsum = g.colsum != nullptr && tm == 0 && threadIdx.x < BN;
    float csum = 0.0f;
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        load_tile<A_KC>(g.A, g.lda, m0, g.M, kbeg, kend, g.a_vec, ra);
        load_tile<B_KC>(g.B, g.ldb, n0, g.N, kbeg, kend, g.b_vec, rb);
        store_tile<A_KC>(smem[0][0], ra);
        store_tile<B_KC>(smem[0][1], rb);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            load_tile<A_KC>(g.A, g.lda, m0, g.M, kbeg + (kt + 1) * BK, kend, g.a_vec, ra);
            load_tile<B_KC>(g.B, g.ldb, n0, g.N, kbeg + (kt + 1) * BK, kend, g.b_vec, rb);
        }
        const float* As = smem[cur][0] + (lane >> 5) * LDS_LD + wm * 64 + (lane & 31);
        const float* Bs = smem[cur][1] + (lane >> 5) * LDS_LD + wn * 64 + (lane & 31);
        if (do_colsum) {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) csum += smem[cur][1][kk * LDS_LD + threadIdx.x];
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a0 = As[kk * LDS_LD], a1 = As[kk * LDS_LD + 32];
            float b0 = Bs[kk * LDS_LD], b1 = Bs[kk * LDS_LD + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            store_tile<A_KC>(smem[cur ^ 1][0], ra);
            store_tile<B_KC>(smem[cur ^ 1][1], rb);
        }
        __syncthreads();
    }

    if (do_colsum && n0 + (int)threadIdx.x < g.N) unsafeAtomicAdd(g.colsum + n0 + threadIdx.x, csum);
    const bool add_bias = g.bias != nullptr && blockIdx.z == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= g.N) continue;
            const float bv = add_bias ? g.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= g.M) continue;
                float* c = g.C + (size_t)row * g.ldc + col;
                const float v = acc[i][j][r] + bv;
                if (g.atomic) unsafeAtomicAdd(c, v);
                else *c = v;
            }
        }
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
