// The batched products of the H = 1024 path in plain bf16 (amdspeech_lstm_desc.precision = 2, BASELINE configs[4] "bf16 MFMA"), round 5:
// bf16 COPIES of the operands in memory, k-contiguous, and a 256 x 256 tile MFMA kernel that streams them into LDS with
// global_load_lds.  Replaces, for the shapes it takes, gemm_bf16 (gemm_bf3.hip), which reads f32 operands and converts them on the way
// into LDS -- 377 - 445 TFLOP/s at the 63872 x 4096 x 1024 products of configs[4] (profiles/r04_cfg5_bf16_kernel_stats.csv: 55 of the
// step's 125 ms), staging-bound: four bytes per operand value through the load path, eight dword loads per thread for an operand whose
// contraction index is the slow one.  The products: the hoisted x . W_ih of a layer, dX = dG . W_ih^T, dK = [Z ; Hprev]^T . dG
// (/root/reference/models/AcousticModel.py:223-237: the [x ; h] . K product of BasicLSTMCell and its gradients).
//
//   C[M, N] (+)= op(A)[M, K] . op(B)[K, N] (+ bias[N]),  f32 in memory on both sides of the call, f32 accumulation.
//   1. each operand is copied ONCE as bf16 (round to nearest even: the values gemm_bf16 uses), k-contiguous: a plain conversion
//      when the contraction index is the operand's fast one, a 64 x 64 LDS transpose when it is not;
//   2. the kernel: 512 threads = 8 waves (2 x 4), a wave owns 128 x 64 of the 256 x 256 tile = 4 x 2 accumulators of
//      v_mfma_f32_32x32x16_bf16.  The LDS image of a K tile is FRAGMENT-LINEAR: every (32 rows x 16 k) MFMA operand is one
//      contiguous 1 KiB block in lane order (lane l: row l & 31, eight k at 8 (l >> 5)), so a fragment is read with one
//      conflict-free ds_read_b128 per lane and written by ONE global_load_lds_dwordx4 wave-instruction (LDS destination =
//      wave-uniform base + 16 lane: the per-lane SOURCE address carries the whole permutation; no swizzle on either side).
//      Four LDS stages of 32 k (32 KiB each); the loads of the next THREE tiles are in flight across each tile's barrier;
//   3. split K (the weight gradients: M x N = 1024 x 4096 is 64 tiles) writes f32 partial tiles to scratch; one reduce pass adds
//      them to C (no atomics).
#include "common.h"

namespace amdspeech {
namespace {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x2v_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack2(float a, float b) {      // two f32 -> two bf16 (rne), v_cvt_pk_bf16_f32
    const f32x2v_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v_t));
}

// ---- operand copies --------------------------------------------------------------------------------------------------------------
// src [rows][ld] f32 (cols used) -> dst [rows][cols] bf16; cols % 8 == 0.  One thread = eight values (32 B in, 16 B out).
__global__ __launch_bounds__(256) void cvt_rows_kernel(const float* __restrict__ src, long ld, long rows, int cols, bf16_t* __restrict__ dst) {
    const long per = cols / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * per) return;
    const long r = i / per;
    const int c = (int)(i - r * per) * 8;
    const float4 a = *reinterpret_cast<const float4*>(src + r * ld + c), b = *reinterpret_cast<const float4*>(src + r * ld + c + 4);
    u32x4v_t o = {pack2(a.x, a.y), pack2(a.z, a.w), pack2(b.x, b.y), pack2(b.z, b.w)};
    *reinterpret_cast<u32x4v_t*>(dst + r * cols + c) = o;
}
// src [rows][ld] f32 -> dst [cols][ldd] bf16 (dst[c][r] = src[r][c]); rows % 64 == 0, cols % 64 == 0, ldd % 8 == 0.
// One workgroup = one 64 x 64 tile through LDS.  colsum != nullptr: colsum[c] += sum_r src[r][c] (the bias gradient rides along).
// plain != nullptr: also the row-major copy plain[r][c] (dense, pitch cols) from the same read of the tile.
__global__ __launch_bounds__(256) void cvt_transpose_kernel(const float* __restrict__ src, long ld, long rows, int cols,
                                                            bf16_t* __restrict__ dst, long ldd, float* __restrict__ colsum,
                                                            bf16_t* __restrict__ plain) {
    __shared__ bf16_t tile[64][72];      // [c][r], rows of 144 B: 16-byte aligned, bank-spread
    __shared__ float csum[4][64];
    const int tiles_c = cols / 64;
    const long tr = blockIdx.x / tiles_c;
    const int tc = blockIdx.x % tiles_c;
    const long r0 = tr * 64;
    const int c0 = tc * 64;
    const int t = threadIdx.x, cq = (t & 15) * 4, rr = t >> 4;      // four columns, rows rr, rr + 16, rr + 32, rr + 48
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = rr + 16 * k;
        const float4 v = *reinterpret_cast<const float4*>(src + (r0 + r) * ld + c0 + cq);
        tile[cq][r] = (bf16_t)(pack2(v.x, 0.f) & 0xffffu); tile[cq + 1][r] = (bf16_t)(pack2(v.y, 0.f) & 0xffffu);
        tile[cq + 2][r] = (bf16_t)(pack2(v.z, 0.f) & 0xffffu); tile[cq + 3][r] = (bf16_t)(pack2(v.w, 0.f) & 0xffffu);
        s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
        if (plain != nullptr) {
            const uint2 o = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
            *reinterpret_cast<uint2*>(plain + (r0 + r) * cols + c0 + cq) = o;
        }
    }
    if (colsum != nullptr) {
        // sixteen row groups hold partial sums of every column: xor-reduce over the lanes that share cq (lane bits 4, 5 and the wave)
        s0 += __shfl_xor(s0, 16); s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16); s3 += __shfl_xor(s3, 16);
        s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32); s3 += __shfl_xor(s3, 32);
        if ((t & 63) < 16) { csum[t >> 6][cq] = s0; csum[t >> 6][cq + 1] = s1; csum[t >> 6][cq + 2] = s2; csum[t >> 6][cq + 3] = s3; }
    }
    __syncthreads();
    if (colsum != nullptr && t < 64) atomicAdd(colsum + c0 + t, csum[0][t] + csum[1][t] + csum[2][t] + csum[3][t]);
    // 64 columns x 64 rows out: thread -> column t >> 2, sixteen rows at (t & 3) * 16: two 16-byte stores
    const int c = t >> 2, rq = (t & 3) * 16;
    const u32x4v_t lo = *reinterpret_cast<const u32x4v_t*>(&tile[c][rq]), hi = *reinterpret_cast<const u32x4v_t*>(&tile[c][rq + 8]);
    bf16_t* o = dst + (long)(c0 + c) * ldd + r0 + rq;
    *reinterpret_cast<u32x4v_t*>(o) = lo;
    *reinterpret_cast<u32x4v_t*>(o + 8) = hi;
}

// src [rows][cols] bf16 (dense) -> dst [cols][ldd] bf16: the transposed operand copy from the row-major one (lstm_bwd_big1 writes dG as
// bf16 itself); rows % 64 == 0, cols % 64 == 0.  One workgroup = one 64 x 64 tile through LDS.
__global__ __launch_bounds__(256) void bf16_transpose_kernel(const bf16_t* __restrict__ src, long rows, int cols, bf16_t* __restrict__ dst, long ldd) {
    __shared__ bf16_t tile[64][72];      // [c][r]
    const int tiles_c = cols / 64;
    const long tr = blockIdx.x / tiles_c;
    const int tc = blockIdx.x % tiles_c;
    const long r0 = tr * 64;
    const int c0 = tc * 64;
    const int t = threadIdx.x, cq = (t & 15) * 4, rr = t >> 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = rr + 16 * k;
        const uint2 v = *reinterpret_cast<const uint2*>(src + (r0 + r) * cols + c0 + cq);
        tile[cq][r] = (bf16_t)(v.x & 0xffffu); tile[cq + 1][r] = (bf16_t)(v.x >> 16);
        tile[cq + 2][r] = (bf16_t)(v.y & 0xffffu); tile[cq + 3][r] = (bf16_t)(v.y >> 16);
    }
    __syncthreads();
    const int c = t >> 2, rq = (t & 3) * 16;
    const u32x4v_t lo = *reinterpret_cast<const u32x4v_t*>(&tile[c][rq]), hi = *reinterpret_cast<const u32x4v_t*>(&tile[c][rq + 8]);
    bf16_t* o = dst + (long)(c0 + c) * ldd + r0 + rq;
    *reinterpret_cast<u32x4v_t*>(o) = lo;
    *reinterpret_cast<u32x4v_t*>(o + 8) = hi;
}

// ---- the product ------------------------------------------------------------------------------------------------------------------
constexpr int BM = 256, BN = 256;
constexpr int BKT = 32;                        // k per tile: two k steps of the MFMA
constexpr int NST = 4;                         // LDS stages: the loads of THREE tiles are in flight while one is multiplied
constexpr int KS = BKT / 16;
constexpr int FRAG = 1024;                     // bytes of one (32 rows x 16 k) operand fragment
constexpr int OPER = 8 * KS * FRAG;            // one operand of a K tile: 8 row blocks x KS k steps
constexpr int STAGE = 2 * OPER;                // A + B

struct PackedArgs {
    const bf16_t* A; const bf16_t* B;          // A [M][lda], B [N][ldb]: k contiguous
    float* C; const float* bias;
    long lda, ldb, ldc;
    int M, N, K;
    int tiles_n, splits, k_tiles_per_split;    // grid = tiles_m * tiles_n * splits
    float* partial;                            // splits > 1: [split][tile][256][256] f32 partial tiles (then gemm_bf16p_reduce_kernel)
    int accumulate;                            // splits == 1: C += (else C =)
};

// The first version kept ONE tile of loads in flight (two 64 KiB stages, vmcnt(0) in front of every barrier): 690 - 810 TFLOP/s --
// a tile's 32 MFMAs per wave are 2048 cycles per SIMD, an operand that comes from memory takes twice that to land, and neither a
// register-double-buffered fragment read nor an XCD-aware tile order moved it.  Now: four stages of 32 k, the loads of tiles
// kt+1 .. kt+3 stay in flight ACROSS the barrier of tile kt (a counted s_waitcnt vmcnt in front of a raw s_barrier: __syncthreads()
// makes hipcc drain vmcnt).
// (Round 5, measured and not kept: the two wave rows of the tile half a trip apart -- [barrier; refill; read the tile's fragments;
//  barrier; multiply] with row 1 entering the loop one barrier late, so that one row's 48 KiB of fragment reads run under the other's
//  MFMAs: parity-green, 208 VGPRs, and the same 500 - 625 TFLOP/s stand-alone, 72 - 73 ms per configs[4] step either way.  The LDS and
//  the matrix pipe are not what serialises here.)
__global__ __launch_bounds__(512) void gemm_bf16p_kernel(PackedArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    int bid = blockIdx.x;
    const int split = bid % g.splits; bid /= g.splits;
    const int tn = bid % g.tiles_n, tm = bid / g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kt0 = split * g.k_tiles_per_split;
    const int nkt_all = g.K / BKT;
    const int nkt = min(g.k_tiles_per_split, nkt_all - kt0);

    // this wave's source rows: row block w of A and of B, lane l -> row l & 31, k octet l >> 5
    const bf16_t* a_src = g.A + (long)min(m0 + w * 32 + (lane & 31), g.M - 1) * g.lda + 8 * (lane >> 5);
    const bf16_t* b_src = g.B + (long)min(n0 + w * 32 + (lane & 31), g.N - 1) * g.ldb + 8 * (lane >> 5);
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    typedef __attribute__((address_space(1))) const unsigned char glb_byte;
    auto fill = [&](int stage, int kt) __attribute__((always_inline)) {      // 2 KS wave-instructions
        lds_byte* base = (lds_byte*)(smem + stage * STAGE + w * KS * FRAG);
        const long k0 = (long)(kt0 + kt) * BKT;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            __builtin_amdgcn_global_load_lds((glb_byte*)(a_src + k0 + ks * 16), (lds_byte*)(base + ks * FRAG), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_byte*)(b_src + k0 + ks * 16), (lds_byte*)(base + OPER + ks * FRAG), 16, 0, 0);
        }
    };
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nkt) fill(t, t);
    for (int kt = 0; kt < nkt; ++kt) {
        // tile kt has landed when at most the loads of the tiles behind it are still in flight (each 2 KS per wave, in order)
        const int behind = min(NST - 2, nkt - 1 - kt);
        if (behind >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * 2 * KS) : "memory");
        else if (behind == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * KS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // behind the barrier everybody's loads of tile kt have landed, and nobody still reads the stage of tile kt - 1
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + NST - 1 < nkt) fill((kt + NST - 1) % NST, kt + NST - 1);
        const unsigned char* sa = smem + (kt % NST) * STAGE + lane * 16;
        const unsigned char* sb = sa + OPER;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8_t a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(sa + ((wm * 4 + i) * KS + ks) * FRAG);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(sb + ((wn * 2 + j) * KS + ks) * FRAG);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // ---- out: accumulator register r of lane l is (row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31) of its 32 x 32 block
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
    if (g.splits > 1) {
        float* p = g.partial + ((long)split * (gridDim.x / g.splits) + (long)tm * g.tiles_n + tn) * (BM * BN);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    p[(wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + row_l) * BN + wn * 64 + j * 32 + col_l] = acc[i][j][r];
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + col_l;
        const float bv = (g.bias != nullptr && col < g.N) ? g.bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + row_l;
                if (row < g.M && col < g.N) {
                    float* c = g.C + (long)row * g.ldc + col;
                    *c = acc[i][j][r] + bv + (g.accumulate ? *c : 0.0f);
                }
            }
    }
}

// C[row][col] (+)= sum over splits of the partial tiles (+ bias)
__global__ __launch_bounds__(256) void gemm_bf16p_reduce_kernel(const float* __restrict__ partial, int splits, int tiles_m, int tiles_n,
                                                                float* __restrict__ C, long ldc, int M, int N, const float* __restrict__ bias,
                                                                int accumulate) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;      // four consecutive columns of one tile row
    const long per_tile = (long)BM * BN, ntiles = (long)tiles_m * tiles_n;
    if (i >= ntiles * per_tile) return;
    const long tile = i / per_tile, e = i - tile * per_tile;
    const int tm = (int)(tile / tiles_n), tn = (int)(tile % tiles_n);
    const int row = tm * BM + (int)(e / BN), col = tn * BN + (int)(e % BN);
    float4 s = *reinterpret_cast<const float4*>(partial + i);
    for (int k = 1; k < splits; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(partial + k * ntiles * per_tile + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (row >= M) return;
    float* c = C + (long)row * ldc + col;
    const float vals[4] = {s.x, s.y, s.z, s.w};
    for (int q = 0; q < 4; ++q)
        if (col + q < N) c[q] = vals[q] + (bias ? bias[col + q] : 0.0f) + (accumulate ? c[q] : 0.0f);
}

}  // namespace

// ---- host side: the two steps separately (lstm.hip shares copies between products), and the one-call form -----------------------
// bf16 copy of a row-major f32 matrix src [rows][ld] (cols used).  !transpose: dst [rows][cols] (cols % 8 == 0);
// transpose: dst [cols][ldd] (rows % 64 == 0, cols % 64 == 0, ldd % 8 == 0), colsum[c] += sum_r src[r][c] when given, and -- `plain` --
// the dense row-major copy too, from the same read.
int bf16p_copy(hipStream_t s, const float* src, long ld, long rows, int cols, bool transpose, unsigned short* dst, long ldd, float* colsum,
               unsigned short* plain) {
    AS_CHECK_ARG(src && dst && rows > 0 && cols > 0 && ld % 4 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "bf16p_copy: bad arguments");
    if (!transpose) {
        AS_CHECK_ARG(cols % 8 == 0 && colsum == nullptr && plain == nullptr && ldd == cols, "bf16p_copy: plain copies are dense, cols %% 8 == 0");
        hipLaunchKernelGGL(cvt_rows_kernel, dim3(ceil_div(rows * (cols / 8), 256)), dim3(256), 0, s, src, ld, rows, cols, dst);
    } else {
        AS_CHECK_ARG(rows % 64 == 0 && cols % 64 == 0 && ldd % 8 == 0 && ldd >= rows, "bf16p_copy: transposing copies work in 64 x 64 tiles");
        hipLaunchKernelGGL(cvt_transpose_kernel, dim3((unsigned)((rows / 64) * (cols / 64))), dim3(256), 0, s, src, ld, rows, cols, dst, ldd, colsum, plain);
    }
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}
int bf16p_transpose(hipStream_t s, const unsigned short* src, long rows, int cols, unsigned short* dst, long ldd) {
    AS_CHECK_ARG(src && dst && rows > 0 && rows % 64 == 0 && cols % 64 == 0 && ldd % 8 == 0 && ldd >= rows, "bf16p_transpose: 64 x 64 tiles");
    hipLaunchKernelGGL(bf16_transpose_kernel, dim3((unsigned)((rows / 64) * (cols / 64))), dim3(256), 0, s, src, rows, cols, dst, ldd);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}
static int bf16p_splits(int M, int N, int K) {
    const int tiles = ceil_div(M, BM) * ceil_div(N, BN), nkt = K / BKT;
    int splits = 1;
    if (tiles < 192 && nkt >= 64) {
        splits = 256 / tiles;
        if (splits > 8) splits = 8;
        if (splits < 1) splits = 1;
    }
    const int per = ceil_div(nkt, splits);
    return ceil_div(nkt, per);
}
size_t bf16p_partial_bytes(int M, int N, int K) {
    const int splits = bf16p_splits(M, N, K);
    return splits > 1 ? (size_t)splits * ceil_div(M, BM) * ceil_div(N, BN) * BM * BN * sizeof(float) : 0;
}
// C[M][N] (+)= Ak[M][K] . Bk[N][K]^T (+ bias), both operands bf16 with k contiguous (lda, ldb in elements, multiples of 8)
int bf16p_gemm(hipStream_t s, int M, int N, int K, const unsigned short* Ak, long lda, const unsigned short* Bk, long ldb, float* C, long ldc,
               const float* bias, bool accumulate, void* partial, size_t partial_bytes) {
    AS_CHECK_ARG(M > 0 && N > 0 && K >= BKT && K % BKT == 0 && lda % 8 == 0 && ldb % 8 == 0 && Ak && Bk && C, "bf16p_gemm: bad arguments");
    PackedArgs g;
    g.A = Ak; g.B = Bk; g.C = C; g.bias = bias; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    const int tiles_m = ceil_div(M, BM);
    g.tiles_n = ceil_div(N, BN);
    const int tiles = tiles_m * g.tiles_n, nkt = K / BKT;
    g.splits = bf16p_splits(M, N, K);
    g.k_tiles_per_split = ceil_div(nkt, g.splits);
    AS_CHECK_ARG(g.splits == 1 || (partial != nullptr && partial_bytes >= bf16p_partial_bytes(M, N, K)), "bf16p_gemm: no room for the split-K partial tiles");
    g.partial = static_cast<float*>(partial); g.accumulate = accumulate ? 1 : 0;
    static unsigned long long attr_done = 0;
    if (DeviceOnce once{&attr_done}) {
        AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
        once.done();
    }
    hipLaunchKernelGGL(gemm_bf16p_kernel, dim3(tiles * g.splits), dim3(512), NST * STAGE, s, g);
    if (g.splits > 1)
        hipLaunchKernelGGL(gemm_bf16p_reduce_kernel, dim3(ceil_div((long)tiles * BM * BN / 4, 256)), dim3(256), 0, s, g.partial, g.splits, tiles_m,
                           g.tiles_n, C, ldc, M, N, bias, accumulate ? 1 : 0);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

// Shapes the one-call form takes (else: gemm_bf16): K a multiple of 64; an operand that has to be transposed needs rows (= K) and
// columns in multiples of 64; 16-byte aligned f32 rows.
bool gemm_bf16_packed_ok(bool transA, bool transB, int M, int N, int K, int lda, int ldb) {
    if (M < 256 || N < 256 || K < 64 || K % 64 != 0) return false;
    if (lda % 4 != 0 || ldb % 4 != 0) return false;
    if (transA && M % 64 != 0) return false;          // A stored [K][M]: transposed in 64 x 64 tiles
    if (!transB && N % 64 != 0) return false;         // B stored [K][N]
    return true;
}
size_t gemm_bf16_packed_scratch_bytes(int M, int N, int K) {
    return align_up((size_t)M * K * 2, 256) + align_up((size_t)N * K * 2, 256) + align_up(bf16p_partial_bytes(M, N, K), 256) + 256;
}
// C[M,N] (+)= op(A) . op(B) (+ bias); colsum != nullptr (needs !transB, i.e. B stored [K][N]): colsum[n] += sum_k B[k][n].
int gemm_bf16_packed(hipStream_t s, bool transA, bool transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, const float* bias, bool accumulate, float* colsum, void* scratch, size_t scratch_bytes) {
    AS_CHECK_ARG(gemm_bf16_packed_ok(transA, transB, M, N, K, lda, ldb), "gemm_bf16_packed: shape not taken");
    AS_CHECK_ARG(scratch != nullptr && scratch_bytes >= gemm_bf16_packed_scratch_bytes(M, N, K) && ((uintptr_t)scratch % 256) == 0,
                 "gemm_bf16_packed: scratch too small or misaligned");
    AS_CHECK_ARG(colsum == nullptr || !transB, "gemm_bf16_packed: the column sums ride on the transposing copy of B");
    unsigned short* ak = static_cast<unsigned short*>(scratch);
    unsigned short* bk = reinterpret_cast<unsigned short*>(static_cast<char*>(scratch) + align_up((size_t)M * K * 2, 256));
    char* partial = reinterpret_cast<char*>(bk) + align_up((size_t)N * K * 2, 256);
    if (int rc = bf16p_copy(s, A, lda, transA ? K : M, transA ? M : K, transA, ak, K, nullptr, nullptr)) return rc;      // A as [M][K]
    if (int rc = bf16p_copy(s, B, ldb, transB ? N : K, transB ? K : N, !transB, bk, K, colsum, nullptr)) return rc;      // B as [N][K]
    return bf16p_gemm(s, M, N, K, ak, K, bk, K, C, ldc, bias, accumulate, partial, bf16p_partial_bytes(M, N, K));
}

}  // namespace amdspeech

using namespace amdspeech;

extern "C" size_t amdspeech_gemm_bf16_packed_scratch_bytes(int trans_a, int trans_b, int M, int N, int K, int lda, int ldb) {
    if (!gemm_bf16_packed_ok(trans_a != 0, trans_b != 0, M, N, K, lda, ldb)) return 0;
    return gemm_bf16_packed_scratch_bytes(M, N, K);
}
extern "C" int amdspeech_gemm_bf16_packed(void* stream, int trans_a, int trans_b, int M, int N, int K, const float* A, int lda,
                                          const float* B, int ldb, float* C, int ldc, const float* bias, int accumulate,
                                          void* scratch, size_t scratch_bytes) {
    AS_CHECK_ARG(A && B && C, "gemm_bf16_packed: null pointer");
    return gemm_bf16_packed(static_cast<hipStream_t>(stream), trans_a != 0, trans_b != 0, M, N, K, A, lda, B, ldb, C, ldc, bias,
                            accumulate != 0, nullptr, scratch, scratch_bytes);
}
