// Split-precision ("bf16x3") GEMM for the batched products of the H = 1024 path when the caller asked for
// precision = bf16x3 (amdspeech_lstm_desc.precision = 1; BASELINE configs[4] "bf16 MFMA"): the hoisted x . W_ih of every layer
// (/root/reference/models/AcousticModel.py:223-237, the input half of the BasicLSTMCell product), its backward
// dX = dG . W_ih^T and the weight gradients dK = [Z ; Hprev]^T . dG.  C[M,N] (+)= op(A)[M,K] . op(B)[K,N] (+ bias[N]).
//
// Operands and result stay f32 in memory.  Every value x is used as bf16 hi = rne(x) and bf16 lo = rne(x - hi) (16 significant
// bits) and every product as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 with f32 accumulation -- the same arithmetic
// the recurrence kernels use in this mode (lstm.hip flow_bf3_*); the dropped lo.lo term is <= 2^-16 relative.  Three bf16
// MFMAs of 8 passes cover K = 16 where exact f32 needs eight 32x32x2 MFMAs of 16 passes: 5.3x less matrix-pipe time.
//
// gfx950 design: 128x128 block tile, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile = 2x2 accumulators of 32x32.
// K advances in steps of 32.  The split happens ONCE per element, on the way from global memory into LDS (so its VALU cost --
// ~3 instructions per value -- is shared by the four waves; splitting in every consuming wave would make the kernel VALU
// bound): LDS holds bf16 hi and lo planes [128 rows][32 k], k contiguous, which is exactly the MFMA operand layout
// (lane l: row l & 31, eight consecutive k at 8 (l >> 5)): one conflict-free ds_read_b128 per fragment (64-byte rows, the
// octets of a row XOR-swizzled: plane_off).  Global loads: a k-contiguous operand is read as 64
// contiguous bytes per thread; a row-contiguous one as eight dword loads per thread (one per k; 64 lanes = 64 consecutive
// rows = 256 contiguous bytes per instruction) so that a thread owns eight consecutive k of ONE row and writes them as one
// 16-byte LDS word per plane.  Register-staged one K step ahead of the MFMAs; one 32 KiB LDS stage, three workgroups per CU.
#include "common.h"
#include <stdlib.h>

namespace amdspeech {

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int TM = 128, TN = 128, TK = 32;
constexpr int ROW_BYTES = 64;                         // 32 k x 2 bytes, no padding: the four 16-byte octets of a row are XOR-swizzled
// byte offset of octet `oct` (8 consecutive k) of row `row` inside a plane: with the octet index XORed by (row >> 1) & 3 eight
// consecutive rows of one octet land in eight distinct 16-byte bank groups (dword index mod 32 = 0,16,4,20,8,24,12,28)
__device__ __forceinline__ int plane_off(int row, int oct) { return row * ROW_BYTES + ((oct ^ ((row >> 1) & 3)) << 4); }
constexpr int PLANE_BYTES = 128 * ROW_BYTES;          // one operand, one of {hi, lo}
constexpr int STAGE_BYTES = 4 * PLANE_BYTES;          // A hi, A lo, B hi, B lo

struct Bf3Args {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc;
    int k_chunk;          // K range per split (multiple of TK)
    int tiles_n;
    int atomic;           // 1: atomicAdd into C (split K / accumulate), 0: plain store
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// eight f32 -> 16 bytes of bf16 hi and 16 bytes of bf16 lo.  Written with vector conversions so that hipcc emits the gfx950
// pack instructions: v_cvt_pk_bf16_f32 (round to nearest even, two values), and / shift, v_pk_add_f32, v_cvt_pk_bf16_f32 --
// 2.5 VALU instructions per value (an integer restatement of the rounding cost 16 and made the kernel VALU bound).
__device__ __forceinline__ void split8(const float (&x)[8], u32x4_t& hi, u32x4_t& lo) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f32x2_t v = {x[2 * p], x[2 * p + 1]};
        const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
        const f32x2_t rest = v - __builtin_convertvector(h, f32x2_t);
        const bf16x2_t l = __builtin_convertvector(rest, bf16x2_t);
        hi[p] = __builtin_bit_cast(unsigned, h);
        lo[p] = __builtin_bit_cast(unsigned, l);
    }
}

// One operand tile (128 rows x 32 k) from global memory into this thread's staging registers: 16 floats = two octets of
// eight consecutive k of one row each.  Raw buffer loads: the K range ends where the buffer resource ends (a read past it
// returns zeros: no tail code, no predicated load -- hipcc turns `cond ? load : 0` into a branch and an s_waitcnt per load);
// rows past the operand are clamped to the last row (their products land in rows / columns of C that are never stored).
//   KC ("k contiguous", element (r, k) at P[r * ld + k]; K % 32 == 0): thread t -> row t >> 1, k half (t & 1) * 16
//   RC ("row contiguous", element (r, k) at P[k * ld + r]): thread t -> row t & 127, octets (t >> 7) and (t >> 7) + 2;
//       the frame index sits in the SCALAR offset (t >> 7 is wave-uniform), the row in the one loop-invariant vector offset
typedef __amdgpu_buffer_rsrc_t bufrsrc_t;
template <bool KC>
struct OperandView {
    bufrsrc_t rs;
    unsigned voff;        // this thread's loop-invariant byte offset
    unsigned ld4;         // RC: bytes between consecutive k
    int oct0;             // RC: this wave's first octet (0 or 1)
};
template <bool KC>
__device__ __forceinline__ OperandView<KC> make_view(const float* P, int ld, int row0, int nrows, int kend, int tid) {
    OperandView<KC> v;
    if (KC) {
        const int r = min(row0 + (tid >> 1), nrows - 1);
        v.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, (unsigned)(((size_t)(nrows - 1) * ld + kend) * 4), 0x00020000);
        v.voff = (unsigned)(((size_t)r * ld + (tid & 1) * 16) * 4);
        v.ld4 = 0; v.oct0 = 0;
    } else {
        const int r = min(row0 + (tid & 127), nrows - 1);
        v.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, (unsigned)((size_t)kend * ld * 4), 0x00020000);
        v.voff = (unsigned)(r * 4);
        v.ld4 = (unsigned)ld * 4u;
        v.oct0 = __builtin_amdgcn_readfirstlane(tid >> 7);
    }
    return v;
}
template <bool KC>
__device__ __forceinline__ void load_operand(const OperandView<KC>& v, int k0, float (&reg)[16]) {
    if (KC) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4_t w = __builtin_amdgcn_raw_buffer_load_b128(v.rs, v.voff + (unsigned)(q * 16), (unsigned)k0 * 4u, 0);
            reg[4 * q] = __uint_as_float(w[0]); reg[4 * q + 1] = __uint_as_float(w[1]);
            reg[4 * q + 2] = __uint_as_float(w[2]); reg[4 * q + 3] = __uint_as_float(w[3]);
        }
    } else {
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned so = (unsigned)(k0 + (v.oct0 + 2 * o) * 8 + q) * v.ld4;
                reg[8 * o + q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(v.rs, v.voff, so, 0));
            }
    }
}

template <bool KC, bool SINGLE = false>
__device__ __forceinline__ void store_operand(unsigned char* hi_plane, unsigned char* lo_plane, const float (&reg)[16], int tid) {
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        float x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = reg[8 * o + q];
        u32x4_t hi, lo;
        split8(x, hi, lo);
        int row, oct;
        if (KC) { row = tid >> 1; oct = (tid & 1) * 2 + o; }
        else { row = tid & 127; oct = (tid >> 7) + 2 * o; }
        *reinterpret_cast<u32x4_t*>(hi_plane + plane_off(row, oct)) = hi;
        if (!SINGLE) *reinterpret_cast<u32x4_t*>(lo_plane + plane_off(row, oct)) = lo;
    }
}

// SINGLE (round 4, precision = bf16): every value as ONE bf16 (round to nearest even) and every product as ONE MFMA -- plain
// bf16 operands with f32 accumulation, the arithmetic BASELINE configs[4] names; the lo planes and two thirds of the MFMAs go.
template <bool A_KC, bool B_KC, bool SINGLE = false>
__global__ __launch_bounds__(256) void gemm_bf3_kernel(Bf3Args g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [A hi | A lo | B hi | B lo]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD x (= blockIdx % 8) gets a contiguous range of (split, tile) pairs: it streams ONE K range / neighbouring tiles
    int v = blockIdx.x;
    {
        const int nwg = gridDim.x, x = v & 7, q = nwg >> 3, r = nwg & 7;
        v = x * q + min(x, r) + (v >> 3);
    }
    const int tiles = g.tiles_n * ((g.M + TM - 1) / TM);
    const int tile = v % tiles, split = v / tiles;
    const int m0 = (tile / g.tiles_n) * TM, n0 = (tile % g.tiles_n) * TN;
    const int kbeg = split * g.k_chunk, kend = min(g.K, kbeg + g.k_chunk);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float ra[16], rb[16];
    const int nsteps = (kend - kbeg + TK - 1) / TK;
    const OperandView<A_KC> va = make_view<A_KC>(g.A, g.lda, m0, g.M, kend, tid);
    const OperandView<B_KC> vb = make_view<B_KC>(g.B, g.ldb, n0, g.N, kend, tid);
    if (nsteps > 0) {
        load_operand<A_KC>(va, kbeg, ra);
        load_operand<B_KC>(vb, kbeg, rb);
        store_operand<A_KC, SINGLE>(smem, smem + PLANE_BYTES, ra, tid);
        store_operand<B_KC, SINGLE>(smem + 2 * PLANE_BYTES, smem + 3 * PLANE_BYTES, rb, tid);
    }
    __syncthreads();
    // ONE LDS stage of 32 KiB (two barriers per K step) rather than two of 32: three workgroups fit a CU instead of two, and it
    // is the other workgroups that fill a workgroup's barriers, load latencies and split phase (measured at the 5x1024 shapes:
    // 1 / 2 / 3 workgroups per CU = 187-213 / 258-286 / ~300 TFLOP/s-equivalent)
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) {
            load_operand<A_KC>(va, kbeg + (s + 1) * TK, ra);
            load_operand<B_KC>(vb, kbeg + (s + 1) * TK, rb);
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {               // two MFMA k-steps of 16 per staged K step
            bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int off = plane_off(wm * 64 + i * 32 + (lane & 31), (lane >> 5) + 2 * sub);
                ah[i] = *reinterpret_cast<const bf16x8_t*>(smem + off);
                if (!SINGLE) al[i] = *reinterpret_cast<const bf16x8_t*>(smem + PLANE_BYTES + off);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int off = plane_off(wn * 64 + j * 32 + (lane & 31), (lane >> 5) + 2 * sub);
                bh[j] = *reinterpret_cast<const bf16x8_t*>(smem + 2 * PLANE_BYTES + off);
                if (!SINGLE) bl[j] = *reinterpret_cast<const bf16x8_t*>(smem + 3 * PLANE_BYTES + off);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    if (!SINGLE) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    }
                }
        }
        if (s + 1 < nsteps) {
            __syncthreads();                               // every wave has read its fragments of step s
            store_operand<A_KC, SINGLE>(smem, smem + PLANE_BYTES, ra, tid);
            store_operand<B_KC, SINGLE>(smem + 2 * PLANE_BYTES, smem + 3 * PLANE_BYTES, rb, tid);
            __syncthreads();
        }
    }

    const bool add_bias = g.bias != nullptr && split == 0;
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
                const float val = acc[i][j][r] + bv;
                if (g.atomic) unsafeAtomicAdd(c, val);
                else *c = val;
            }
        }
}

__global__ void bf3_fill_kernel(float* C, int M, int N, int ldc, float v) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)M * N) C[(i / N) * ldc + (i % N)] = v;
}

}  // namespace

// Same contract as gemm_f32 (common.h) without the fused column sum / gate: transX != 0 means the operand is stored
// transposed (A as [K,M], B as [N,K]).
static int gemm_bf_any(hipStream_t s, bool single, bool transA, bool transB, int M, int N, int K, const float* A, int lda, const float* B,
                       int ldb, float* C, int ldc, const float* bias, bool accumulate) {
    AS_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0, "gemm_bf3: bad arguments");
    AS_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "gemm_bf3: operands must be 16-byte aligned");
    {
        // what the buffer addressing of the kernel needs; anything else takes the exact-f32 kernel (a superset in accuracy)
        const bool a_kc = !transA, b_kc = transB;
        const size_t a_bytes = (a_kc ? (size_t)M * lda : (size_t)K * lda) * 4, b_bytes = (b_kc ? (size_t)N * ldb : (size_t)K * ldb) * 4;
        const bool ok = a_bytes < (1ull << 32) && b_bytes < (1ull << 32) && (!a_kc || (K % TK == 0 && lda % 4 == 0)) &&
                        (!b_kc || (K % TK == 0 && ldb % 4 == 0));
        if (!ok) return gemm_f32(s, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate);
    }
    Bf3Args g;
    g.A = A; g.B = B; g.C = C; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    const int tiles_m = ceil_div(M, TM);
    g.tiles_n = ceil_div(N, TN);
    const int tiles = tiles_m * g.tiles_n;
    // split K until every CU has its three workgroups (they overlap each other's phases), as long as a split keeps >= 16 K steps
    int splits = 1;
    if (tiles < 768) {
        splits = ceil_div(768, tiles);
        const int max_splits = K / (16 * TK);
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    g.k_chunk = ceil_div(ceil_div(K, splits), TK) * TK;
    splits = ceil_div(K, g.k_chunk);
    g.atomic = (accumulate || splits > 1) ? 1 : 0;
    if (!accumulate && splits > 1) {
        const long n = (long)M * N;
        hipLaunchKernelGGL(bf3_fill_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, C, M, N, ldc, 0.0f);
    }
    constexpr size_t lds = (size_t)STAGE_BYTES;           // 32 KiB
    static unsigned long long seen = 0;
    if (DeviceOnce once{&seen}) {
        AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf3_kernel<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf3_kernel<true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf3_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf3_kernel<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf3_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf3_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf3_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf3_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        once.done();
    }
    dim3 grid(tiles * splits), block(256);
    static const size_t lds_req = (size_t)dev_knob("AMDSPEECH_BF3_LDS", (int)(lds / 1024)) * 1024;      // dev: occupancy probe
    // A "KC" = k contiguous = NOT transposed storage [M,K]; B "KC" = stored [N,K] = transposed.
    if (single) {      // (same 32 KiB request: the lo planes stay unused, the occupancy is what the kernel was tuned at)
        if (!transA && !transB) hipLaunchKernelGGL((gemm_bf3_kernel<true, false, true>), grid, block, lds_req, s, g);
        else if (!transA && transB) hipLaunchKernelGGL((gemm_bf3_kernel<true, true, true>), grid, block, lds_req, s, g);
        else if (transA && !transB) hipLaunchKernelGGL((gemm_bf3_kernel<false, false, true>), grid, block, lds_req, s, g);
        else hipLaunchKernelGGL((gemm_bf3_kernel<false, true, true>), grid, block, lds_req, s, g);
    } else {
        if (!transA && !transB) hipLaunchKernelGGL((gemm_bf3_kernel<true, false>), grid, block, lds_req, s, g);
        else if (!transA && transB) hipLaunchKernelGGL((gemm_bf3_kernel<true, true>), grid, block, lds_req, s, g);
        else if (transA && !transB) hipLaunchKernelGGL((gemm_bf3_kernel<false, false>), grid, block, lds_req, s, g);
        else hipLaunchKernelGGL((gemm_bf3_kernel<false, true>), grid, block, lds_req, s, g);
    }
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}
int gemm_bf3(hipStream_t s, bool transA, bool transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
             float* C, int ldc, const float* bias, bool accumulate) {
    return gemm_bf_any(s, false, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate);
}
int gemm_bf16(hipStream_t s, bool transA, bool transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
              float* C, int ldc, const float* bias, bool accumulate) {
    return gemm_bf_any(s, true, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate);
}

}  // namespace amdspeech

// Exposed for tests / bench: the split-precision counterpart of amdspeech_gemm_f32.
extern "C" int amdspeech_gemm_bf16x3(void* stream, int transA, int transB, int M, int N, int K, const float* A, int lda,
                                     const float* B, int ldb, float* C, int ldc, const float* bias, int accumulate) {
    return amdspeech::gemm_bf3(static_cast<hipStream_t>(stream), transA != 0, transB != 0, M, N, K, A, lda, B, ldb, C, ldc, bias,
                               accumulate != 0);
}

// ... and the plain-bf16 one (precision = bf16: one bf16 per value, one MFMA per product, f32 accumulation).
extern "C" int amdspeech_gemm_bf16(void* stream, int transA, int transB, int M, int N, int K, const float* A, int lda,
                                   const float* B, int ldb, float* C, int ldc, const float* bias, int accumulate) {
    return amdspeech::gemm_bf16(static_cast<hipStream_t>(stream), transA != 0, transB != 0, M, N, K, A, lda, B, ldb, C, ldc, bias,
                                accumulate != 0);
}
