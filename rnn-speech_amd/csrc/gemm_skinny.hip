// The dense layers either side of the LSTM stack are GEMMs with one SHORT axis and a 32k-frame M axis:
//   output Linear   logits[M,80]  = ztop[M,H] . W_o[H,80] + b_o        (N short)      models/AcousticModel.py:241-247
//   its input grad  dztop[M,H]    = dlogits[M,80] . W_o^T              (K short)
//   input Linear    Z_0[M,H]      = x[M,40] . W_i[40,H] + b_i           (K short)      models/AcousticModel.py:205-213
// The 128x128x16 LDS kernel (gemm.hip) wastes 3/8 of its tile on an 80-wide output (and splits K = 512 with atomics to find
// enough workgroups), and pays its pipeline fill for five K steps on an 80-long reduction.  Two kernels for these shapes, exact
// f32 on v_mfma_f32_16x16x4_f32, no split K, no atomics, every operand byte read once from HBM:
//
//  * gemm_skinny_n_kernel<NT> (N <= 16 NT <= 96):  a wave owns 16 rows and ALL N columns (NT accumulators); the weight
//    chunk [64 k][N] goes through LDS as it lies in memory (double buffered, one barrier per chunk; row pitch chosen so that
//    both the b128 fill and the per-MFMA ds_read_b32 are bank-conflict free); the rows stream from HBM as one b128 per lane
//    per 16 k, a chunk ahead.
//  * gemm_skinny_k_kernel<KT, B_KC> (K <= 16 KT <= 80):  a wave owns 32 rows; their whole K extent lives in registers
//    (2 x KT b128 fragments) while the wave sweeps the N axis two 16-column tiles at a time; the weight fragments come
//    straight from L2 (the matrix is a few hundred KiB), a tile pair ahead, into ping-pong register sets.
//
// Both use the operand-swapped product (weights on the MFMA's A port, rows on B): the accumulator of a lane is then FOUR
// CONSECUTIVE COLUMNS of one row, i.e. one 16-byte store (and one 16-byte bias / accumulate load).
// The k index of MFMA j inside a 16-k group is 4 (lane >> 4) + j for both operands (any permutation of k is a valid product), which
// is what makes a lane's b128 the operand of four consecutive MFMAs.
#include "common.h"

namespace amdspeech {

typedef __amdgpu_buffer_rsrc_t sk_rsrc_t;
typedef unsigned sk_u32x4 __attribute__((ext_vector_type(4)));

struct SkinnyArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc, accumulate;
};

constexpr unsigned SK_OOB = 0x80000000u;      // an offset past every descriptor's extent: the load returns zeros

__device__ __forceinline__ f32x4 sk_load4(sk_rsrc_t rs, unsigned byte_off) {
    const sk_u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0);
    f32x4 r;
    r.x = __uint_as_float(w.x); r.y = __uint_as_float(w.y); r.z = __uint_as_float(w.z); r.w = __uint_as_float(w.w);
    return r;
}
__device__ __forceinline__ float sk_load1(sk_rsrc_t rs, unsigned byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 0));
}

// Epilogue of one lane: four consecutive columns of one row.  Branch-free (out-of-range lanes get an offset past the descriptor:
// their loads return zero, their stores are dropped; a null bias / a non-accumulating call gets an EMPTY descriptor), so that
// the compiler keeps exact vmcnt counts around it -- a store inside a branch turns the wait for the NEXT tile's prefetched
// fragments into vmcnt(0), i.e. into a wait for this tile's stores to be acknowledged.
struct SkinnyOut {
    sk_rsrc_t c, c_in, bias;
    __device__ __forceinline__ void init(const SkinnyArgs& g) {
        const unsigned cbytes = (unsigned)(((size_t)(g.M - 1) * g.ldc + g.N) * 4);
        c = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, cbytes, 0x00020000);
        c_in = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, g.accumulate ? cbytes : 0u, 0x00020000);
        bias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.bias), 0, g.bias != nullptr ? (unsigned)g.N * 4u : 0u, 0x00020000);
    }
    __device__ __forceinline__ unsigned offset(const SkinnyArgs& g, int row, int n) const {
        return (row < g.M && n < g.N) ? (unsigned)((size_t)row * g.ldc + n) * 4u : SK_OOB;      // (N % 4 == 0: all in or all out)
    }
    __device__ __forceinline__ f32x4 addend(const SkinnyArgs& g, unsigned off, int n) const {   // bias + old contents
        return sk_load4(bias, n < g.N ? (unsigned)n * 4u : SK_OOB) + sk_load4(c_in, off);
    }
    __device__ __forceinline__ void store(unsigned off, f32x4 v) const {
        sk_u32x4 w;
        w.x = __float_as_uint(v.x); w.y = __float_as_uint(v.y); w.z = __float_as_uint(v.z); w.w = __float_as_uint(v.w);
        __builtin_amdgcn_raw_buffer_store_b128(w, c, off, 0, 0);
    }
};

template <int NT>
__global__ __launch_bounds__(256) void gemm_skinny_n_kernel(SkinnyArgs g) {
    // weight chunk as it lies in memory, [k][n]: rows of LD = 16 NT + 4 floats.  4 LD = 16 (mod 32) banks puts the four k of one
    // MFMA (lanes >> 4) on different bank halves: the ds_read_b32 of a fragment is conflict free, and so is the b128 write.
    constexpr int KC = 64, LD = NT * 16 + ((NT & 1) ? 4 : 12);
    static_assert((4 * LD) % 32 == 16 && LD % 4 == 0, "bank spread");
    __shared__ __attribute__((aligned(16))) float wt[2][KC][LD];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, m = lane & 15, kg = lane >> 4;
    const int row = blockIdx.x * 64 + w * 16 + m;
    const sk_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, (unsigned)(((size_t)(g.M - 1) * g.lda + g.K) * 4), 0x00020000);
    const sk_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, (unsigned)(((size_t)(g.K - 1) * g.ldb + g.N) * 4), 0x00020000);
    const int chunks = (g.K + KC - 1) / KC;

    f32x4 wreg[NT], a_cur[4], a_nxt[4];
    auto load_w = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int slot = i * 256 + tid, k = slot / (4 * NT), n4 = slot % (4 * NT);
            const int kk = c * KC + k;
            const bool ok = n4 * 4 < g.N && kk < g.K;
            wreg[i] = sk_load4(rb, ok ? (unsigned)(kk * g.ldb + n4 * 4) * 4u : SK_OOB);
        }
    };
    auto put_w = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int slot = i * 256 + tid, k = slot / (4 * NT), n4 = slot % (4 * NT);
            *reinterpret_cast<f32x4*>(&wt[buf][k][n4 * 4]) = wreg[i];
        }
    };
    auto load_a = [&](int c, f32x4* a) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kk = c * KC + 16 * s + 4 * kg;
            const bool ok = row < g.M && kk < g.K;
            a[s] = sk_load4(ra, ok ? (unsigned)((size_t)row * g.lda + kk) * 4u : SK_OOB);
        }
    };

    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    load_w(0);
    load_a(0, a_cur);
    put_w(0);
    __syncthreads();
    for (int c = 0; c < chunks; ++c) {
        const int buf = c & 1;
        load_w(c + 1);                               // (no branch: past the last chunk every offset is out of range -> zeros)
        load_a(c + 1, a_nxt);
        __builtin_amdgcn_sched_barrier(0);           // (keep the prefetch above the MFMAs it hides under)
        float bq[2][NT][4];                          // weight fragments of one 16-k group, read a group ahead of their MFMAs
        auto read_b = [&](int s, float (*b)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float* col = &wt[buf][16 * s + 4 * kg][nt * 16 + m];
                b[nt][0] = col[0]; b[nt][1] = col[LD]; b[nt][2] = col[2 * LD]; b[nt][3] = col[3 * LD];
            }
        };
        read_b(0, bq[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3) read_b(s + 1, bq[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[s & 1][nt][j], a_cur[s][j], acc[nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        put_w(buf ^ 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) a_cur[s] = a_nxt[s];
        __syncthreads();
    }
    SkinnyOut out;
    out.init(g);
    unsigned off[NT];
    f32x4 add[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        off[nt] = out.offset(g, row, nt * 16 + 4 * kg);
        add[nt] = out.addend(g, off[nt], nt * 16 + 4 * kg);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) out.store(off[nt], acc[nt] + add[nt]);
}

template <int KT, bool B_KC>
__global__ __launch_bounds__(256) void gemm_skinny_k_kernel(SkinnyArgs g) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, m = lane & 15, kg = lane >> 4;
    const int row0 = blockIdx.x * 128 + w * 32;
    const sk_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, (unsigned)(((size_t)(g.M - 1) * g.lda + g.K) * 4), 0x00020000);
    const size_t b_extent = B_KC ? (size_t)(g.N - 1) * g.ldb + g.K : (size_t)(g.K - 1) * g.ldb + g.N;
    const sk_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, (unsigned)(b_extent * 4), 0x00020000);
    SkinnyOut out;
    out.init(g);

    f32x4 a[2][KT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int row = row0 + mt * 16 + m, kk = 16 * kt + 4 * kg;
            const bool ok = row < g.M && kk < g.K;
            a[mt][kt] = sk_load4(ra, ok ? (unsigned)((size_t)row * g.lda + kk) * 4u : SK_OOB);
        }
    // everything the pair of 16-column tiles (nt, nt + 1) needs from memory: weight fragments, bias, old contents.  LOADS ONLY (no
    // arithmetic on them here: it would be scheduled next to the loads and wait for them a whole tile early).
    struct Tile { f32x4 b[2][KT], bias[2], old[2][2]; unsigned off[2][2]; };
    auto fetch = [&](int nt, Tile& t) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int n = (nt + q) * 16 + m;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int kk = 16 * kt + 4 * kg;
                const bool ok = n < g.N && kk < g.K;
                if (B_KC) {
                    t.b[q][kt] = sk_load4(rb, ok ? (unsigned)(n * g.ldb + kk) * 4u : SK_OOB);
                } else {
                    const unsigned o = ok ? (unsigned)(kk * g.ldb + n) * 4u : SK_OOB;
                    const unsigned st = ok ? (unsigned)g.ldb * 4u : 0u;
                    t.b[q][kt].x = sk_load1(rb, o);
                    t.b[q][kt].y = sk_load1(rb, o + st);
                    t.b[q][kt].z = sk_load1(rb, o + 2 * st);
                    t.b[q][kt].w = sk_load1(rb, o + 3 * st);
                }
            }
            const int nc = (nt + q) * 16 + 4 * kg;
            t.bias[q] = sk_load4(out.bias, nc < g.N ? (unsigned)nc * 4u : SK_OOB);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                t.off[mt][q] = out.offset(g, row0 + mt * 16 + m, nc);
                t.old[mt][q] = sk_load4(out.c_in, t.off[mt][q]);
            }
        }
    };
    auto tile = [&](int nt, const Tile& cur, Tile& nxt) __attribute__((always_inline)) {
        fetch(nt + 2, nxt);                          // (past the last tile: offsets out of range, the loads return zeros)
        __builtin_amdgcn_sched_barrier(0);           // (the scheduler otherwise sinks the prefetch below the MFMAs it is meant to hide under)
        f32x4 acc[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[mt][q] = cur.bias[q] + cur.old[mt][q];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.b[q][kt][j], a[mt][kt][j], acc[mt][q], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 2; ++q) out.store(cur.off[mt][q], acc[mt][q]);
        __builtin_amdgcn_sched_barrier(0);
    };
    // blockIdx.y: which slice of the N axis (32032 rows are only ~1000 waves of 32: two column slices put two waves on every SIMD)
    const int all_tiles = (g.N + 15) / 16;
    const int per = ((all_tiles + (int)gridDim.y - 1) / (int)gridDim.y + 1) & ~1;
    const int nt0 = (int)blockIdx.y * per, ntiles = min(all_tiles, nt0 + per);
    Tile t0, t1;                                     // ping-pong: no register copies between tiles
    fetch(nt0, t0);
    for (int nt = nt0; nt < ntiles; nt += 4) {
        tile(nt, t0, t1);
        if (nt + 2 >= ntiles) break;
        tile(nt + 2, t1, t0);
    }
}

// ---- the dense layers' WEIGHT GRADIENTS: a 32k-frame reduction onto a narrow output ------------------------------------------
//   dW_i[40,H] += x^T . dZ_0      db_i += colsum(dZ_0)        dW_o[H,80] += ztop^T . dlogits      db_o += colsum(dlogits)
// C (+)= S^T . W over K rows, S [K, s] the SMALL operand (s <= 124), W [K, wide] the WIDE one; the output is [s, wide] (small is
// the GEMM's A) or [wide, s] (small is its B).  The general kernel reaches this shape through 128x128 tiles that are 1/3 to
// 2/3 empty and 4.2 M atomics.  Here a workgroup of 8 waves owns ONE [s, 64] output tile and a chunk of rows: every wave
// streams its own rows, 4 per row group (lane = (column group i, row kq)), 2 - 4 row groups in flight.  A lane's b128 of the
// wide slice -- columns 4 i .. 4 i + 3 -- is the operand of FOUR column-strided MFMA tiles (tile e = columns {4 i + e}); the
// small operand is cut the same way: FULL b128 fragments of 64 columns (4 tiles each) and a remainder of RT <= 4 floats per lane
// (tile e = columns {64 FULL + RT i + e}), so 41 columns cost 3 tiles and 80 cost 5, not 4 and 8.
// The eight partial tiles meet through LDS (register dumps, pairwise) and leave as ONE set of global atomics per workgroup
// (32 chunks x s x wide: ~1 M).  The 64-column slices of one row chunk sit on ONE XCD, so the small operand crosses the
// fabric once.  The bias gradient costs nothing: colsum of the WIDE operand is a column of ones appended to the small
// operand (one more output row), colsum of the SMALL operand is a few VALU adds per row group on the fragments of slice 0.
#ifndef SKTN_DIAG
#define SKTN_DIAG 0        // dev: 1 no global atomics, 2 no MFMAs, 3 no loads in the loop
#endif
struct SkinnyTnArgs {
    const float* S; const float* W; float* C; float* colsum;
    int K, s, wide, lds, ldw, ldc;
    int chunk;              // rows per workgroup (a multiple of 32)
    int nslices;            // ceil(wide / 64)
    int small_is_a;         // 1: C[s][wide]; 0: C[wide][s]
    int colsum_small;       // colsum (if not null) is over the small operand's columns (else over the wide one's)
};

template <int FULL, int RT>
__global__ __launch_bounds__(512) void gemm_skinny_tn_kernel(SkinnyTnArgs g) {
    constexpr int NTS = 4 * FULL + RT;               // tiles of the small operand
    constexpr int PD = NTS <= 4 ? 4 : 2;             // row groups in flight (what 256 registers leave room for)
    constexpr int SP = 64 * FULL + 16 * RT, LDR = 65;
    constexpr int NF = FULL > 0 ? FULL : 1, NR = RT > 0 ? RT : 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];     // 2 waves' accumulators, later red[SP][LDR] + red_cs[SP]
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, i = lane & 15, kq = lane >> 4;
    // workgroup -> (row chunk, column slice): the slices of one chunk share an XCD (workgroups are dealt round-robin to the 8 XCDs)
    const int v = blockIdx.x, xcd = v & 7, q = v >> 3;
    const int chunk_id = xcd + 8 * (q / g.nslices), slice = q % g.nslices;
    const int r0 = chunk_id * g.chunk;
    if (r0 >= g.K) return;
    const sk_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.S), 0, (unsigned)(((size_t)(g.K - 1) * g.lds + g.s) * 4), 0x00020000);
    const sk_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.W), 0, (unsigned)(((size_t)(g.K - 1) * g.ldw + g.wide) * 4), 0x00020000);

    const int wcol = slice * 64 + 4 * i;
    const bool w_ok = wcol < g.wide;
    const bool ones_wanted = g.colsum != nullptr && !g.colsum_small;
    const int rem = g.s - 64 * FULL;                 // columns of the remainder; the ones column is its column `rem`
    const int rcol = 64 * FULL + RT * i;             // this lane's first remainder column
    bool r_valid[NR];
#pragma unroll
    for (int e = 0; e < NR; ++e) r_valid[e] = RT > 0 && rcol + e < g.s;
    const bool ones_lane = ones_wanted && RT > 0 && i == rem / NR;
    const int ones_e = rem % NR;
    const bool cs_small = g.colsum != nullptr && g.colsum_small && slice == 0;
    const int groups = g.chunk / 32;                                  // row groups of this wave: rows r0 + 4 (8 t + w) + kq
    f32x4 smf[PD][NF], wd[PD];
    float smr[PD][NR];
    auto fetch = [&](int t, int p) __attribute__((always_inline)) {
        const int row = r0 + 4 * (8 * t + w) + kq;
        const bool r_ok = row < g.K && t < groups;
        wd[p] = sk_load4(rw, (r_ok && w_ok) ? (unsigned)((size_t)row * g.ldw + wcol) * 4u : SK_OOB);
#pragma unroll
        for (int b = 0; b < FULL; ++b) smf[p][b] = sk_load4(rs, r_ok ? (unsigned)((size_t)row * g.lds + 64 * b + 4 * i) * 4u : SK_OOB);
        if (RT > 0) {
            // (a lane's RT floats may run past column s into the next row: those elements are selected away below; past the
            // end of the operand the descriptor returns zeros)
            const unsigned off = (r_ok && r_valid[0]) ? (unsigned)((size_t)row * g.lds + rcol) * 4u : SK_OOB;
            if (RT == 1) {
                smr[p][0] = sk_load1(rs, off);
            } else if (RT == 2) {
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                const u2 x = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
                smr[p][0] = __uint_as_float(x.x); smr[p][1 % NR] = __uint_as_float(x.y);
            } else if (RT == 3) {
                typedef unsigned u3 __attribute__((ext_vector_type(3)));
                const u3 x = __builtin_amdgcn_raw_buffer_load_b96(rs, off, 0, 0);
                smr[p][0] = __uint_as_float(x.x); smr[p][1 % NR] = __uint_as_float(x.y); smr[p][2 % NR] = __uint_as_float(x.z);
            } else {
                const f32x4 x = sk_load4(rs, off);
                smr[p][0] = x.x; smr[p][1 % NR] = x.y; smr[p][2 % NR] = x.z; smr[p][3 % NR] = x.w;
            }
        }
    };
    f32x4 acc[NTS][4], csf[NF];
    float csr[NR];
#pragma unroll
    for (int b = 0; b < NF; ++b) csf[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < NR; ++e) csr[e] = 0.f;
#pragma unroll
    for (int ts = 0; ts < NTS; ++ts)
#pragma unroll
        for (int ew = 0; ew < 4; ++ew) acc[ts][ew] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < PD; ++p) fetch(p, p);
    for (int t0 = 0; t0 < groups; t0 += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            float sv[NTS];
            const f32x4 wv = wd[p];
#pragma unroll
            for (int b = 0; b < FULL; ++b) {
#pragma unroll
                for (int es = 0; es < 4; ++es) sv[4 * b + es] = smf[p][b][es];
            }
#pragma unroll
            for (int e = 0; e < RT; ++e) sv[4 * FULL + e] = r_valid[e] ? smr[p][e] : 0.f;
            if (cs_small) {
#pragma unroll
                for (int b = 0; b < FULL; ++b) csf[b] += smf[p][b];
#pragma unroll
                for (int e = 0; e < RT; ++e) csr[e] += sv[4 * FULL + e];
            }
            const int row = r0 + 4 * (8 * (t0 + p) + w) + kq;
            if (ones_lane && row < g.K && t0 + p < groups) {
#pragma unroll
                for (int e = 0; e < RT; ++e)
                    if (e == ones_e) sv[4 * FULL + e] = 1.0f;
            }
#if SKTN_DIAG != 3
            fetch(t0 + p + PD, p);                   // (its registers were just copied out)
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ts = 0; ts < NTS; ++ts)
#pragma unroll
                for (int ew = 0; ew < 4; ++ew)
#if SKTN_DIAG == 2
                    acc[ts][ew] += sv[ts] * wv[ew];
#else
                    acc[ts][ew] = __builtin_amdgcn_mfma_f32_16x16x4f32(sv[ts], wv[ew], acc[ts][ew], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // The eight partial tiles -> one, through LDS as plain register dumps (b128, conflict free; LDS float atomics run at about a
    // lane per clock and cost 45 us here): (6,7 -> 2,3), (4,5 -> 0,1), (2,3 -> 0,1), (1 -> 0); at most two waves dump at a time.
    f32x4* dump = reinterpret_cast<f32x4*>(lds);
    constexpr int NACC = NTS * 4;
    auto put = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int ts = 0; ts < NTS; ++ts)
#pragma unroll
            for (int ew = 0; ew < 4; ++ew) dump[(slot * NACC + ts * 4 + ew) * 64 + lane] = acc[ts][ew];
    };
    auto take = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int ts = 0; ts < NTS; ++ts)
#pragma unroll
            for (int ew = 0; ew < 4; ++ew) acc[ts][ew] += dump[(slot * NACC + ts * 4 + ew) * 64 + lane];
    };
    if (w >= 6) put(w - 6);
    __syncthreads();
    if (w == 2 || w == 3) take(w - 2);
    __syncthreads();
    if (w == 4 || w == 5) put(w - 4);
    __syncthreads();
    if (w < 2) take(w);
    __syncthreads();
    if (w == 2 || w == 3) put(w - 2);
    __syncthreads();
    if (w < 2) take(w);
    __syncthreads();
    if (w == 1) put(0);
    __syncthreads();
    if (w == 0) take(0);
    __syncthreads();
    // wave 0 lays the total out as red[small column][wide column of the slice].  Accumulator register r of a lane is MFMA row
    // 4 kq + r of the small tile: small column 64 b + 4 (4 kq + r) + es of a full fragment, 64 FULL + RT (4 kq + r) + e of the
    // remainder; its wide column is 4 i + ew.
    float* red = lds;
    float* red_cs = lds + SP * LDR;
    if (w == 0) {
#pragma unroll
        for (int ts = 0; ts < NTS; ++ts)
#pragma unroll
            for (int ew = 0; ew < 4; ++ew)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = ts < 4 * FULL ? 64 * (ts / 4) + 4 * (4 * kq + r) + (ts % 4) : 64 * FULL + RT * (4 * kq + r) + (ts - 4 * FULL);
                    red[m * LDR + 4 * i + ew] = acc[ts][ew][r];
                }
    }
    if (tid < SP) red_cs[tid] = 0.f;
    __syncthreads();
    if (cs_small) {                                  // (the four row phases of a wave first, in registers)
#pragma unroll
        for (int b = 0; b < FULL; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = csf[b][e];
                x += __shfl_xor(x, 16);
                x += __shfl_xor(x, 32);
                if (kq == 0) atomicAdd(&red_cs[64 * b + 4 * i + e], x);
            }
#pragma unroll
        for (int e = 0; e < RT; ++e) {
            float x = csr[e];
            x += __shfl_xor(x, 16);
            x += __shfl_xor(x, 32);
            if (kq == 0) atomicAdd(&red_cs[rcol + e], x);
        }
    }
    __syncthreads();
#if SKTN_DIAG == 1
    return;
#endif
    if (g.small_is_a) {                              // C[m][slice cols]: consecutive threads -> consecutive wide columns
        for (int e = tid; e < SP * 64; e += 512) {
            const int m = e >> 6, n = e & 63, col = slice * 64 + n;
            if (col >= g.wide) continue;
            const float val = red[m * LDR + n];
            if (m < g.s) unsafeAtomicAdd(g.C + (size_t)m * g.ldc + col, val);
            else if (m == g.s && ones_wanted) unsafeAtomicAdd(g.colsum + col, val);
        }
    } else {                                         // C[slice cols][m]: consecutive threads -> consecutive small columns
        for (int e = tid; e < SP * 64; e += 512) {
            const int n = e / SP, m = e % SP, col = slice * 64 + n;
            if (col >= g.wide) continue;
            const float val = red[m * LDR + n];
            if (m < g.s) unsafeAtomicAdd(g.C + (size_t)col * g.ldc + m, val);
            else if (m == g.s && ones_wanted) unsafeAtomicAdd(g.colsum + col, val);
        }
    }
    if (cs_small && tid < g.s) unsafeAtomicAdd(g.colsum + tid, red_cs[tid]);
}

__global__ void skinny_zero_kernel(float* C, int rows, int cols, int ldc) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (long)rows * cols) C[(e / cols) * ldc + e % cols] = 0.f;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// C[M,N] (+)= A^T . B with A [K,M], B [K,N] (+ colsum[N] += column sums of B).  1 = taken, 0 = not this shape, < 0 = error.
int gemm_skinny_tn(hipStream_t st, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                   bool accumulate, float* colsum) {
    static const bool enabled = dev_knob("AMDSPEECH_GEMM_SKINNY", 1) != 0;      // (dev A/B: the general 128x128 kernels instead)
    if (!enabled || K < 4096) return 0;
    const bool small_is_a = M <= N;
    const int s = small_is_a ? M : N, wide = small_is_a ? N : M;
    if (s > 124 || wide < 128) return 0;
    if ((s | wide | lda | ldb) & 3) return 0;
    if (!aligned16(A) || !aligned16(B)) return 0;
    const size_t lim = (size_t)SK_OOB;
    if (((size_t)(K - 1) * lda + M) * 4 >= lim || ((size_t)(K - 1) * ldb + N) * 4 >= lim) return 0;
    SkinnyTnArgs g;
    g.S = small_is_a ? A : B; g.lds = small_is_a ? lda : ldb;
    g.W = small_is_a ? B : A; g.ldw = small_is_a ? ldb : lda;
    g.C = C; g.ldc = ldc; g.colsum = colsum; g.K = K; g.s = s; g.wide = wide;
    g.small_is_a = small_is_a ? 1 : 0;
    g.colsum_small = small_is_a ? 0 : 1;             // (the GEMM's B is the operand whose columns are summed)
    g.nslices = ceil_div(wide, 64);
    int nchunks = ceil_div(256, g.nslices);
    nchunks = ceil_div(nchunks, 8) * 8;
    g.chunk = ceil_div(ceil_div(K, nchunks), 32) * 32;
    if (!accumulate) {
        const long n = (long)M * N;
        hipLaunchKernelGGL(skinny_zero_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, C, M, N, ldc);
    }
    const dim3 grid(g.nslices * nchunks), block(512);
    // tiles of the small operand (+ the ones column when the wide operand's column sums are wanted): FULL fragments of 64, then
    // a remainder of RT x 16
    const int s_eff = s + ((colsum != nullptr && !g.colsum_small) ? 1 : 0);
    const int full = s_eff / 64, rt = ceil_div(s_eff - 64 * full, 16);
    const size_t lds = (size_t)2 * (4 * full + rt) * 4 * 64 * 16;          // two waves' accumulators (>= red + red_cs)
#define SK_TN(F, R)                                                                                                          \
    do {                                                                                                                      \
        static unsigned long long seen = 0;                                                                                   \
        if (DeviceOnce once{&seen}) {                                                                                         \
            AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_skinny_tn_kernel<F, R>),                      \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (4 * F + R) * 4 * 64 * 16));      \
            once.done();                                                                                                      \
        }                                                                                                                     \
        hipLaunchKernelGGL((gemm_skinny_tn_kernel<F, R>), grid, block, lds, st, g);                                           \
    } while (0)
    switch (full * 8 + rt) {
        case 1: SK_TN(0, 1); break;
        case 2: SK_TN(0, 2); break;
        case 3: SK_TN(0, 3); break;
        case 4: SK_TN(0, 4); break;
        case 8: SK_TN(1, 0); break;
        case 9: SK_TN(1, 1); break;
        case 10: SK_TN(1, 2); break;
        case 11: SK_TN(1, 3); break;
        case 12: SK_TN(1, 4); break;
        default: return 0;
    }
#undef SK_TN
    AS_CHECK_LAUNCH();
    return 1;
}

// Returns 1 when one of the two kernels took the product, 0 when the shape is not theirs (the caller goes on to the general
// kernels), a negative AMDSPEECH_E* on a launch error.
int gemm_skinny(hipStream_t s, bool transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                const float* bias, bool accumulate) {
    static const bool enabled = dev_knob("AMDSPEECH_GEMM_SKINNY", 1) != 0;      // (dev A/B: the general 128x128 kernels instead)
    if (!enabled || M < 256) return 0;
    if ((N | K | lda | ldb | ldc) & 3) return 0;
    if (!aligned16(A) || !aligned16(B) || !aligned16(C) || (bias != nullptr && !aligned16(bias))) return 0;
    const size_t lim = (size_t)SK_OOB;
    if (((size_t)(M - 1) * lda + K) * 4 >= lim) return 0;
    // (the OUTPUT too: its store offsets are 32-bit with SK_OOB as the "outside" mark -- a wide or strided C of 2 GiB and more
    //  would have its stores dropped or wrapped instead of going to the general kernel)
    if (((size_t)(M - 1) * ldc + N) * 4 >= lim) return 0;
    SkinnyArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.accumulate = accumulate ? 1 : 0;
    if (K <= 80 && N >= 64) {
        const size_t be = transB ? (size_t)(N - 1) * ldb + K : (size_t)(K - 1) * ldb + N;
        if (be * 4 >= lim) return 0;
        const int row_blocks = ceil_div(M, 128);
        const int kt = ceil_div(K, 16);
        // (two column slices measured faster for the 40-long reduction of the input layer, slower for the 80-long one: 27.5 vs 29.5
        // and 39.5 vs 33.6 us at 32032 x 512)
        const dim3 grid(row_blocks, (row_blocks < 512 && N >= 256 && kt <= 3) ? 2 : 1), block(256);
#define SK_K(KT)                                                                                           \
        do {                                                                                               \
            if (transB) hipLaunchKernelGGL((gemm_skinny_k_kernel<KT, true>), grid, block, 0, s, g);        \
            else hipLaunchKernelGGL((gemm_skinny_k_kernel<KT, false>), grid, block, 0, s, g);              \
        } while (0)
        if (kt <= 3) SK_K(3);
        else SK_K(5);
#undef SK_K
        AS_CHECK_LAUNCH();
        return 1;
    }
    if (!transB && N <= 96 && K >= 64) {
        if (((size_t)(K - 1) * ldb + N) * 4 >= lim) return 0;
        const dim3 grid(ceil_div(M, 64)), block(256);
        const int nt = ceil_div(N, 16);
        switch (nt) {
            case 1: hipLaunchKernelGGL(gemm_skinny_n_kernel<1>, grid, block, 0, s, g); break;
            case 2: hipLaunchKernelGGL(gemm_skinny_n_kernel<2>, grid, block, 0, s, g); break;
            case 3: hipLaunchKernelGGL(gemm_skinny_n_kernel<3>, grid, block, 0, s, g); break;
            case 4: hipLaunchKernelGGL(gemm_skinny_n_kernel<4>, grid, block, 0, s, g); break;
            case 5: hipLaunchKernelGGL(gemm_skinny_n_kernel<5>, grid, block, 0, s, g); break;
            default: hipLaunchKernelGGL(gemm_skinny_n_kernel<6>, grid, block, 0, s, g); break;
        }
        AS_CHECK_LAUNCH();
        return 1;
    }
    return 0;
}

}  // namespace amdspeech
