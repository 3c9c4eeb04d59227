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

#ifndef GEMM_XCD_REMAP
#define GEMM_XCD_REMAP 1
#endif
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [buf][A|B][BK * LDS_LD]
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
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Renumber them so that XCD x gets a CONTIGUOUS
    // range of (split, tile) pairs: with split K one XCD then works on ONE K range of the operands (every byte of A and B
    // crosses the fabric once instead of once per XCD).
    int v = blockIdx.x;
    const int nwg = gridDim.x;
#if GEMM_XCD_REMAP
    if (g.xcd_remap) {
        const int x = v & 7, q = nwg >> 3, r = nwg & 7;
        v = x * q + min(x, r) + (v >> 3);
    }
#endif
    const int tiles = g.tiles_n * g.tiles_m;
    WorkgroupBarrier bar;
    gemm_tile<A_KC, B_KC>(g, v % tiles, v / tiles, smem, threadIdx.x, 0, true, bar);
}

// Up to GEMM_GROUP_MAX problems of ONE shape in one launch (the 2 L weight-gradient GEMMs of a backward pass): with several
// workgroups per CU in flight the atomics epilogue of one hides under the main loop of the next, and the launch / drain
// cost is paid once.  Workgroup w sits on XCD w % 8; its s = w / 8 -th turn there is problem s / per, pair (w % 8) * per +
// s % per of that problem's (split, tile) pairs: an XCD works on one K range of one problem at a time.
struct GemmGroupArgs {
    GemmArgs g;                       // shape, strides, split geometry (operands unused)
    const float* A[GEMM_GROUP_MAX]; const float* B[GEMM_GROUP_MAX]; float* C[GEMM_GROUP_MAX]; float* colsum[GEMM_GROUP_MAX];
    int count, pairs;                 // problems; (split, tile) pairs per problem
    int bm, bn;                       // > 0: an XCD's turn is a bm x bn BLOCK of tiles of one split (see gemm_f32_tn_group), not bm*bn tiles of a row
};
__global__ __launch_bounds__(256) void gemm_f32_tn_group_kernel(GemmGroupArgs a) {
    GemmArgs g = a.g;
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
    int problem, pair;
    if ((a.pairs & 7) == 0) {
        const int per = a.pairs >> 3, x = blockIdx.x & 7, sidx = blockIdx.x >> 3;
        problem = sidx / per;
        pair = x * per + sidx % per;
    } else {
        problem = blockIdx.x / a.pairs;
        pair = blockIdx.x % a.pairs;
    }
    g.A = a.A[problem]; g.B = a.B[problem]; g.C = a.C[problem]; g.colsum = a.colsum[problem];
    const int tiles = g.tiles_n * g.tiles_m;
    int tile = pair % tiles;
    if (a.bm > 0) {
        // the `per` = bm * bn consecutive pairs of an XCD's turn as a 2-D block: bm A strips + bn B strips cross the fabric into that
        // XCD's L2 instead of 1 + per (3x512: 4 + 4 strips of 128 columns instead of 1 + 16 -- half the operand bytes)
        const int per = a.bm * a.bn, q = tile / per, j = tile % per, blocks_n = g.tiles_n / a.bn;
        tile = ((q / blocks_n) * a.bm + j / a.bn) * g.tiles_n + (q % blocks_n) * a.bn + j % a.bn;
    }
    gemm_tile_tn_direct(g, tile, pair / tiles, threadIdx.x, true);
}

// A k-contiguous, no LDS (gemm_tile_kc_direct).  Workgroup -> tile: XCD x (= blockIdx % 8) gets a contiguous range of tiles,
// walked in blocks of 8 row tiles x 4 column tiles: what an XCD's 32 CUs work on at one time shares 8 A strips and 4 B strips
// through its L2.
template <bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kc_direct_kernel(GemmArgs g) {
    const int nwg = gridDim.x;
    int v = blockIdx.x;
    {
        const int x = v & 7, q = nwg >> 3, r = nwg & 7;
        v = x * q + min(x, r) + (v >> 3);
    }
    const int tiles = g.tiles_n * g.tiles_m;
    const int split = v / tiles;
    int t = v % tiles, tm, tn;
    if ((g.tiles_n & 3) == 0) {
        const int band = 8 * g.tiles_n;                    // tiles in a band of 8 row tiles
        const int tmb = t / band, rem = t % band;
        const int rows = min(8, g.tiles_m - tmb * 8);      // (the last band may be shorter)
        const int blk = rem / (rows * 4), r2 = rem % (rows * 4);
        tm = tmb * 8 + r2 / 4; tn = blk * 4 + r2 % 4;
    } else {
        tm = t / g.tiles_n; tn = t % g.tiles_n;
    }
    gemm_tile_kc_direct<B_KC>(g, tm, tn, split, threadIdx.x);
}

__global__ void fill_strided_kernel(float* C, int M, int N, int ldc, float v) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)M * N) C[(i / N) * ldc + (i % N)] = v;
}

// C_i [M,N] (+)= A_i^T . B_i for `count` problems of one shape, both operands row contiguous ([K][M], [K][N]); see
// gemm_tile_tn_direct.  Returns AMDSPEECH_OK, or -1 when the shape / alignment does not qualify (nothing launched).
static bool tn_direct_ok(int M, int N, int K, const float* A, int lda, const float* B, int ldb) {
    static const bool enabled = runtime_switch("AMDSPEECH_GEMM_DIRECT", 1) != 0;
    return enabled && M >= 2 && N >= 2 && (uintptr_t)A % 16 == 0 && lda % 4 == 0 && (uintptr_t)B % 16 == 0 && ldb % 4 == 0 &&
           (size_t)(K + 4 * GEMM_TN_DEPTH) * (size_t)(lda > ldb ? lda : ldb) * 4 < (1ull << 32);
}
bool gemm_f32_tn_group_ok(int M, int N, int K, const float* A, int lda, const float* B, int ldb) {
    return tn_direct_ok(M, N, K, A, lda, B, ldb);
}
int gemm_f32_tn_group(hipStream_t s, int count, int M, int N, int K, const float* const* A, int lda, const float* const* B,
                      int ldb, float* const* C, int ldc, float* const* colsum, bool accumulate,
                      const int* gate, int gate_need, unsigned* gate_err) {
    AS_CHECK_ARG(count >= 1 && count <= GEMM_GROUP_MAX && M > 0 && N > 0 && K > 0, "gemm group: bad shape");
    GemmGroupArgs a;
    GemmArgs& g = a.g;
    g.A = nullptr; g.B = nullptr; g.C = nullptr; g.bias = nullptr; g.colsum = nullptr;
    g.gate = gate; g.gate_need = gate_need; g.gate_limit = 300000000ull; g.gate_err = gate_err;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.tiles_m = ceil_div(M, BM); g.tiles_n = ceil_div(N, BN);
    const int tiles = g.tiles_m * g.tiles_n;
    // one workgroup (one wave per SIMD) per CU keeps the MFMA pipe full here: split K up to 256 workgroups per problem,
    // no further (every split ends in a tile of f32 atomics)
    static const int target_wgs = dev_knob("AMDSPEECH_GEMM_TN_WGS", 256);
    int splits = 1;
    if (tiles * count < target_wgs) {
        splits = ceil_div(target_wgs, tiles * count);
        const int max_splits = K / 256 > 0 ? K / 256 : 1;
        if (splits > max_splits) splits = max_splits;
    }
    g.k_chunk = ceil_div(ceil_div(K, splits), 2) * 2;
    splits = ceil_div(K, g.k_chunk);
    g.atomic = (accumulate || splits > 1) ? 1 : 0;
    g.a_vec = g.b_vec = 1; g.xcd_remap = 0;
    a.count = count; a.pairs = tiles * splits;
    // What an XCD's 32 CUs work on at one time (`per` pairs of one split of one problem) decides what crosses the fabric into its L2:
    // a row of `per` tiles streams 1 A strip and `per` B strips, a bm x bn block bm + bn.  The block must tile the output.
    a.bm = a.bn = 0;
#ifndef GEMM_TN_BLOCKED
#define GEMM_TN_BLOCKED 1         // (dev: -DGEMM_TN_BLOCKED=0 = rows of tiles, rounds 2-5)
#endif
    static const int blocked = dev_knob("AMDSPEECH_GEMM_TN_BLOCKED", GEMM_TN_BLOCKED);
    if (blocked && (a.pairs & 7) == 0) {
        const int per = a.pairs >> 3;
        if (per > 1 && tiles % per == 0) {
            int best = 0;
            for (int bm = 1; bm <= per; ++bm) {
                if (per % bm != 0 || g.tiles_m % bm != 0 || g.tiles_n % (per / bm) != 0) continue;
                if (best == 0 || bm + per / bm < best + per / best) best = bm;
            }
            if (best > 1) { a.bm = best; a.bn = per / best; }
        }
    }
    for (int i = 0; i < GEMM_GROUP_MAX; ++i) {
        const int j = i < count ? i : 0;
        AS_CHECK_ARG(A[j] && B[j] && C[j] && tn_direct_ok(M, N, K, A[j], lda, B[j], ldb), "gemm group: operand %d does not qualify", j);
        a.A[i] = A[j]; a.B[i] = B[j]; a.C[i] = C[j]; a.colsum[i] = colsum ? colsum[j] : nullptr;
        if (i < count && !accumulate && splits > 1) {
            const long n = (long)M * N;
            hipLaunchKernelGGL(fill_strided_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, C[j], M, N, ldc, 0.0f);
        }
    }
    // occupancy: an (unused) LDS request caps the workgroups per CU
    // one workgroup = one wave per SIMD per CU: a second streaming wave on a SIMD slows both (8.5 -> 7.3 ms for the two
    // 1024 x 4096 x 64064 products of a cfg3 layer)
    static const int occ_lds = dev_knob("AMDSPEECH_GEMM_TN_LDS", 96 * 1024);
    static unsigned long long tn_attr_seen = 0;
    if (DeviceOnce once{&tn_attr_seen}) {
        AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_tn_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once.done();
    }
    hipLaunchKernelGGL(gemm_f32_tn_group_kernel, dim3(count * a.pairs), dim3(256), (size_t)occ_lds, s, a);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

int gemm_f32(hipStream_t s, bool transA, bool transB, int M, int N, int K, const float* A, int lda,
             const float* B, int ldb, float* C, int ldc, const float* bias, bool accumulate, float* colsum,
             const int* gate, int gate_need, unsigned* gate_err) {
    AS_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: non-positive shape %d %d %d", M, N, K);
    AS_CHECK_ARG(A && B && C, "gemm: null operand");
    // one short axis against the 32k-frame M axis (the dense layers either side of the stack): gemm_skinny.hip
    if (!transA && colsum == nullptr && gate == nullptr) {
        const int took = gemm_skinny(s, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate);
        if (took != 0) return took < 0 ? took : AMDSPEECH_OK;
    }
    if (transA && !transB && bias == nullptr && gate == nullptr) {
        const int took = gemm_skinny_tn(s, M, N, K, A, lda, B, ldb, C, ldc, accumulate, colsum);
        if (took != 0) return took < 0 ? took : AMDSPEECH_OK;
    }
    // both operands row contiguous and tiles that are mostly full: the LDS-free kernel (narrow outputs -- the dense layers'
    // 40- and 80-wide weight gradients -- measured faster through LDS)
    if (transA && !transB && bias == nullptr && M >= 96 && N >= 96 && tn_direct_ok(M, N, K, A, lda, B, ldb))
        return gemm_f32_tn_group(s, 1, M, N, K, &A, lda, &B, ldb, &C, ldc, colsum ? &colsum : nullptr, accumulate, gate, gate_need, gate_err);
    // A k-contiguous (no transA), K a multiple of 64, 16-byte aligned rows, output at least a tile wide, no fused column sums:
    // the LDS-free kernel for the dZ_0 / dX products (transB) and the x.W products
    static const bool kc_direct = runtime_switch("AMDSPEECH_GEMM_KC_DIRECT", 1) != 0;
    // (short K: the pipeline fill per tile is not amortised -- K = 1024 x.W products measured 3 % faster through LDS)
    static const int kc_min_k = dev_knob("AMDSPEECH_KC_MIN_K", 2048);
    if (kc_direct && !transA && colsum == nullptr && gate == nullptr && K % 64 == 0 && K >= kc_min_k && M >= 128 && N >= 96 &&
        (uintptr_t)A % 16 == 0 && lda % 4 == 0 && (uintptr_t)B % 16 == 0 && ldb % 4 == 0 &&
        (size_t)M * lda * 4 < (1ull << 32) && (size_t)(transB ? N : K + 64) * ldb * 4 < (1ull << 32)) {
        GemmArgs g;
        g.A = A; g.B = B; g.C = C; g.bias = bias; g.colsum = nullptr; g.gate = nullptr; g.gate_err = nullptr; g.gate_need = 0; g.gate_limit = 0;
        g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
        g.tiles_m = ceil_div(M, BM); g.tiles_n = ceil_div(N, BN);
        const int tiles = g.tiles_m * g.tiles_n;
        int splits = 1;
        if (tiles < 192) {
            splits = ceil_div(256, tiles);
            if (splits > K / 256) splits = K / 256 > 0 ? K / 256 : 1;
        }
        g.k_chunk = ceil_div(ceil_div(K, splits), 64) * 64;
        splits = ceil_div(K, g.k_chunk);
        g.atomic = (accumulate || splits > 1) ? 1 : 0;
        g.a_vec = g.b_vec = 1; g.xcd_remap = 1;
        if (!accumulate && splits > 1) {
            const long n = (long)M * N;
            hipLaunchKernelGGL(fill_strided_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, C, M, N, ldc, 0.0f);
        }
        if (transB) hipLaunchKernelGGL(gemm_f32_kc_direct_kernel<true>, dim3(tiles * splits), dim3(256), 0, s, g);
        else hipLaunchKernelGGL(gemm_f32_kc_direct_kernel<false>, dim3(tiles * splits), dim3(256), 0, s, g);
        AS_CHECK_LAUNCH();
        return AMDSPEECH_OK;
    }
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
    static const int target_wgs = dev_knob("AMDSPEECH_GEMM_WGS", 512);
    if (tiles < target_wgs) {
        splits = ceil_div(target_wgs, tiles);
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
    g.tiles_m = tiles_m;
    g.xcd_remap = splits > 1 ? 1 : 0;
    dim3 grid(tiles * splits), block(256);
    constexpr size_t lds = (size_t)2 * 2 * BK * LDS_LD * sizeof(float);
    static unsigned long long lds_seen = 0;
    if (DeviceOnce once{&lds_seen}) {
        AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        once.done();
    }
    // A "KC" = k contiguous = NOT transposed storage [M,K]; B "KC" = stored [N,K] = transposed.
    if (!transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, lds, s, g);
    else if (!transA && transB) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, lds, s, g);
    else if (transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, lds, s, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, lds, s, g);
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

// The same for 16-byte aligned rows (HBM roofline: the reduced-precision path sums 1 GB of dG per layer this way): a lane owns four
// adjacent columns, a wave 1 KiB of a row, the four waves stride the rows eight deep -- 32 x 16 bytes in flight per lane.
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ x, int rows, int cols,
                                                      int ld, float* __restrict__ out, int rows_per_block) {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    __shared__ f32x4_t red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (c < cols) {
        const float* p = x + c;
        int r = r0 + w;
        for (; r + 28 < r1; r += 32) {
            f32x4_t v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p + (size_t)(r + 4 * q) * ld));
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q & 1] += v[q];
        }
        for (; r < r1; r += 4) acc[0] += *reinterpret_cast<const f32x4_t*>(p + (size_t)r * ld);
    }
    red[w][lane] = acc[0] + acc[1];
    __syncthreads();
    if (w == 0 && c < cols) {
        const f32x4_t t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) unsafeAtomicAdd(out + c + q, t[q]);
    }
}

int colsum_accumulate(hipStream_t s, const float* x, int rows, int cols, int ld, float* out) {
    AS_CHECK_ARG(x && out && rows > 0 && cols > 0, "colsum: bad arguments");
    if (cols % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const int col_blocks = ceil_div(cols, 256);
        int row_blocks = ceil_div(2048, col_blocks);
        if (row_blocks > ceil_div(rows, 128)) row_blocks = ceil_div(rows, 128);
        if (row_blocks < 1) row_blocks = 1;
        const int rpb = ceil_div(ceil_div(rows, row_blocks), 4) * 4;
        row_blocks = ceil_div(rows, rpb);
        hipLaunchKernelGGL(colsum4_kernel, dim3(col_blocks, row_blocks), dim3(256), 0, s, x, rows, cols, ld, out, rpb);
        AS_CHECK_LAUNCH();
        return AMDSPEECH_OK;
    }
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
