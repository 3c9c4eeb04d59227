// Core of the f32 MFMA GEMM (see gemm.hip): tile constants, argument block, operand staging and the
// per-tile body as a device function, shared by gemm_f32_kernel and the in-kernel GEMM workers of lstm_bwd_flow.
#pragma once
#include "common.h"
#include <type_traits>

namespace amdspeech {

#ifndef GEMM_PRELOAD
#define GEMM_PRELOAD 0
#endif
#ifndef GEMM_BK
#define GEMM_BK 16
#endif
constexpr int BM = 128, BN = 128, BK = GEMM_BK, KSUB = BK / 16;      // a K tile is KSUB sub-tiles of 16 (one staging round each)
constexpr int LDS_LD = BM + 4;  // +4 floats: breaks the 128-float stride for the transposing writes

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc;
    int k_chunk;      // K range per split (multiple of BK)
    int tiles_n, tiles_m;
    int xcd_remap;    // 1: renumber the workgroups so that each XCD gets a contiguous range of (split, tile) pairs
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
                                          int k0, int kend, bool vec, float (&reg)[8], int tid) {
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
__device__ __forceinline__ void store_tile(float* __restrict__ S, const float (&reg)[8], int tid) {
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

// One 128x128 output tile over one K split: the body of gemm_f32_kernel, also run by the GEMM worker workgroups
// inside lstm_bwd_flow (two 256-thread teams per 512-thread workgroup, each with its own LDS area and its own
// TeamBarrier).  `nk_force` > 0 runs exactly that many K tiles (tiles past the split's end load zeros); `commit`
// false computes but stores nothing.
// Barrier policies of gemm_tile: the whole workgroup (gemm_f32_kernel: one 256-thread team per workgroup), or one
// 256-thread team of a larger workgroup through an LDS counter (the GEMM workers of lstm_bwd_flow: two teams per
// workgroup that must NOT run in lock step -- one team's operand staging overlaps the other's MFMAs).
struct WorkgroupBarrier {
    __device__ __forceinline__ void sync() { __syncthreads(); }
};
struct TeamBarrier {
    unsigned* count;      // LDS word of this team, zeroed once; every wave adds 1 per barrier
    unsigned target = 0;
    int waves;
    __device__ __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        target += waves;
        if ((threadIdx.x & 63) == 0) atomicAdd(count, 1u);
        while (*reinterpret_cast<volatile unsigned*>(count) < target) { }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
};

template <bool A_KC, bool B_KC, class Barrier>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, int tile, int split, float* smem_base, int tid,
                                          int nk_force, bool commit, Barrier& bar) {
    float (*smem)[2][BK * LDS_LD] = reinterpret_cast<float (*)[2][BK * LDS_LD]>(smem_base);   // [buf][A|B]
    const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = split * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float ra[KSUB][8], rb[KSUB][8];
    // bias gradient fused into the weight-gradient GEMM: the first row of tiles also sums its B tile
    // over k (B is dY [K = frames, N]); every B element is visited by exactly one such workgroup
    const bool do_colsum = g.colsum != nullptr && tm == 0 && tid < BN;
    float csum = 0.0f;
    const int nk = nk_force > 0 ? nk_force : (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
#pragma unroll
        for (int u = 0; u < KSUB; ++u) {
            load_tile<A_KC>(g.A, g.lda, m0, g.M, kbeg + u * 16, kend, g.a_vec, ra[u], tid);
            load_tile<B_KC>(g.B, g.ldb, n0, g.N, kbeg + u * 16, kend, g.b_vec, rb[u], tid);
        }
#pragma unroll
        for (int u = 0; u < KSUB; ++u) {
            store_tile<A_KC>(smem[0][0] + u * 16 * LDS_LD, ra[u], tid);
            store_tile<B_KC>(smem[0][1] + u * 16 * LDS_LD, rb[u], tid);
        }
    }
    bar.sync();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
#pragma unroll
            for (int u = 0; u < KSUB; ++u) {
                load_tile<A_KC>(g.A, g.lda, m0, g.M, kbeg + (kt + 1) * BK + u * 16, kend, g.a_vec, ra[u], tid);
                load_tile<B_KC>(g.B, g.ldb, n0, g.N, kbeg + (kt + 1) * BK + u * 16, kend, g.b_vec, rb[u], tid);
            }
        }
        const float* As = smem[cur][0] + (lane >> 5) * LDS_LD + wm * 64 + (lane & 31);
        const float* Bs = smem[cur][1] + (lane >> 5) * LDS_LD + wn * 64 + (lane & 31);
        if (do_colsum) {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) csum += smem[cur][1][kk * LDS_LD + tid];
        }
#if GEMM_PRELOAD
        // all fragments of this K tile first (BK x 2 registers), then the MFMAs back to back: the compiler otherwise reuses
        // four registers and waits for LDS (lgkmcnt(0)) in front of every group of four MFMAs
        float af[BK / 2][2], bf[BK / 2][2];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            af[kk / 2][0] = As[kk * LDS_LD]; af[kk / 2][1] = As[kk * LDS_LD + 32];
            bf[kk / 2][0] = Bs[kk * LDS_LD]; bf[kk / 2][1] = Bs[kk * LDS_LD + 32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][0], bf[kk][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][0], bf[kk][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][1], bf[kk][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][1], bf[kk][1], acc[1][1], 0, 0, 0);
        }
#else
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a0 = As[kk * LDS_LD], a1 = As[kk * LDS_LD + 32];
            float b0 = Bs[kk * LDS_LD], b1 = Bs[kk * LDS_LD + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
#endif
        if (kt + 1 < nk) {
#pragma unroll
            for (int u = 0; u < KSUB; ++u) {
                store_tile<A_KC>(smem[cur ^ 1][0] + u * 16 * LDS_LD, ra[u], tid);
                store_tile<B_KC>(smem[cur ^ 1][1] + u * 16 * LDS_LD, rb[u], tid);
            }
        }
        bar.sync();
    }

    if (!commit) return;
    if (do_colsum && n0 + (int)tid < g.N) unsafeAtomicAdd(g.colsum + n0 + tid, csum);
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
                const float v = acc[i][j][r] + bv;
                if (g.atomic) unsafeAtomicAdd(c, v);
                else *c = v;
            }
        }
}

// ---- C[M,N] += A^T . B with BOTH operands "row contiguous" (A [K][M], B [K][N]: the weight-gradient GEMMs, K = frames):
// no LDS, no barrier.  The 32x32x2 MFMA wants lane l to hold A[k + (l >> 5)][m + (l & 31)] -- for one k a contiguous run
// of the operand's row -- so the fragments come straight from L2 / L1 into registers: one 8-byte load per lane and operand
// gives TWO fragments (even / odd rows of a 64-row strip; the accumulators are simply stored with that row order).  A wave
// is then an independent stream "4 MFMAs, 2 loads" with the loads GEMM_TN_DEPTH k-pairs ahead; nothing stalls on a
// workgroup barrier, and (measured) two such waves per SIMD keep the MFMA pipe busier than the LDS-staged
// tile does.  Rows past the split's K range read as zeros (buffer bounds check): no tail code.  The four waves of a
// 128x128 tile share A / B strips through the CU's L1.
#ifndef GEMM_TN_DEPTH
#define GEMM_TN_DEPTH 12
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gemm_tile_tn_direct(const GemmArgs& g, int tile, int split, int tid, bool commit) {
    constexpr int D = GEMM_TN_DEPTH;
    const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = split * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l32 = lane & 31;
    // (wave-uniform by construction; inside the GEMM workers of lstm_bwd_flow the compiler cannot see it: the team index
    //  comes from threadIdx.x >> 8 -- the resource descriptors and scalar offsets have to live in SGPRs)
    auto uni_ptr = [](const float* p) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
    };
    const int krows = __builtin_amdgcn_readfirstlane(kend - kbeg);
    const int lda = __builtin_amdgcn_readfirstlane(g.lda), ldb = __builtin_amdgcn_readfirstlane(g.ldb);
    const auto ra_src = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(g.A + (size_t)kbeg * g.lda), 0, (unsigned)((size_t)krows * lda * 4), 0x00020000);
    const auto rb_src = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(g.B + (size_t)kbeg * g.ldb), 0, (unsigned)((size_t)krows * ldb * 4), 0x00020000);
    // (a lane whose rows / columns lie past M / N reads its neighbour's data or zeros: those accumulators are never stored)
    const unsigned voa = (unsigned)((half * lda + min(m0 + wm * 64 + 2 * l32, lda - 2)) * 4);
    const unsigned vob = (unsigned)((half * ldb + min(n0 + wn * 64 + 2 * l32, ldb - 2)) * 4);
    const unsigned sa = (unsigned)(2 * lda * 4), sb = (unsigned)(2 * ldb * 4);      // bytes per k-pair
    // The loads and their waits are inline assembly: left to the compiler, the ring of D fragment registers costs an
    // s_waitcnt vmcnt(0) on every trip of the loop (the prefetch distance collapses) and register copies on the back edge.
    // "+v" keeps every fragment in ONE physical register for the whole loop; the MFMAs hang on the wait through it.
#define TN_LOAD(dst, rsrc, vo, so) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "+v"(dst) : "v"(vo), "s"(rsrc), "s"(so))
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const bool do_colsum = g.colsum != nullptr && tm == 0 && wm == 0;      // (uniform per wave)
    f32x2 cs = {0.0f, 0.0f};
    const int nsteps = (krows + 1) / 2;
    // (the whole pipeline -- prologue loads, loop, final wait -- lives in ONE instantiation per variant: fragments that cross
    //  from one region to another are copied by the compiler while their loads are still in flight)
    auto run = [&](auto with_colsum) {
        f32x2 fa[D], fb[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            fa[j] = (f32x2){0.f, 0.f}; fb[j] = (f32x2){0.f, 0.f};
            TN_LOAD(fa[j], ra_src, voa, (unsigned)j * sa);
            TN_LOAD(fb[j], rb_src, vob, (unsigned)j * sb);
        }
        unsigned soa = (unsigned)D * sa, sob = (unsigned)D * sb;       // offsets of the k-pair D steps ahead
        for (int s0 = 0; s0 < nsteps; s0 += D) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                // 2 D loads are in flight, this step's pair is the oldest.  (The bias-gradient column sums ride in the same
                // statement: as separate C++ the compiler defers the adds and keeps copies of the fragments alive.)
                if (decltype(with_colsum)::value)
                    asm volatile("s_waitcnt vmcnt(%3)\n\tv_pk_add_f32 %2, %2, %1" : "+v"(fa[j]), "+v"(fb[j]), "+v"(cs) : "n"(2 * D - 2));
                else
                    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(fa[j]), "+v"(fb[j]) : "n"(2 * D - 2));
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j][0], fb[j][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j][0], fb[j][1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j][1], fb[j][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j][1], fb[j][1], acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                TN_LOAD(fa[j], ra_src, voa, soa);       // (past the split's end: zeros)
                TN_LOAD(fb[j], rb_src, vob, sob);
                soa += sa; sob += sb;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the trailing loads must land before their registers are reused
#pragma unroll
        for (int j = 0; j < D; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(fa[j]), "+v"(fb[j]));
    };
    if (do_colsum) run(std::true_type{}); else run(std::false_type{});
    float cs0 = cs[0], cs1 = cs[1];
#undef TN_LOAD
    if (!commit) return;
    if (do_colsum) {
        cs0 += __shfl_xor(cs0, 32); cs1 += __shfl_xor(cs1, 32);
        const int col = n0 + wn * 64 + 2 * l32;
        if (half == 0) {
            if (col < g.N) unsafeAtomicAdd(g.colsum + col, cs0);
            if (col + 1 < g.N) unsafeAtomicAdd(g.colsum + col + 1, cs1);
        }
    }
    const bool add_bias = g.bias != nullptr && split == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + 2 * l32 + j;
            if (col >= g.N) continue;
            const float bv = add_bias ? g.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * half) + i;
                if (row >= g.M) continue;
                float* c = g.C + (size_t)row * g.ldc + col;
                const float v = acc[i][j][r] + bv;
                if (g.atomic) unsafeAtomicAdd(c, v);
                else *c = v;
            }
        }
}

// ---- C[M,N] (+)= A . op(B) with A "k contiguous" ([M][K] row-major: the dZ_0 / dX products A = dG, and every x.W product),
// again without LDS.  Lane (r = l & 31, h = l >> 5) reads 16 bytes of ITS row: k = kb + 16 h + 4 q + c, c = 0..3 -- so the two
// halves of a wave consume one 128-byte line of every row per 32-k block, and MFMA step (q, c) contracts the k pair
// {kb + 4q + c, kb + 16 + 4q + c}.  Which k a lane holds does not matter as long as A and B agree:
//   B_KC  (B [N][K], "NT"): the same 16-byte loads on B's rows;
//   !B_KC (B [K][N], "NN"): one 8-byte load per step and lane at row kb + 16 h + 4 q + c (two adjacent columns, as in
//                           gemm_tile_tn_direct).
// A unit = one q (4 steps, 16 MFMAs); a ring of 8 units (64 k) is in flight.  Needs K % 64 == 0 per split (a k past the
// end of a row would read the next row) -- the hot path's K are 512 ... 4096.
template <bool B_KC>
__device__ __forceinline__ void gemm_tile_kc_direct(const GemmArgs& g, int tile_m, int tile_n, int split, int tid) {
    constexpr int D = 8;      // ring: 8 units = 2 blocks of 32 k
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l32 = lane & 31;
    const int lda = g.lda, ldb = g.ldb;
    const auto ra_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, (unsigned)((size_t)g.M * lda * 4), 0x00020000);
    const auto rb_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0,
                                                          (unsigned)((size_t)(B_KC ? g.N : g.K) * ldb * 4), 0x00020000);
    // (rows past M / N are clamped: valid memory, results never stored)
    unsigned voa[2], vob[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        voa[i] = (unsigned)((min(m0 + wm * 64 + i * 32 + l32, g.M - 1) * lda + 16 * half + kbeg) * 4);
        vob[i] = B_KC ? (unsigned)((min(n0 + wn * 64 + i * 32 + l32, g.N - 1) * ldb + 16 * half + kbeg) * 4) : 0u;
    }
    const unsigned vob_mc = (unsigned)(((kbeg + 16 * half) * ldb + min(n0 + wn * 64 + 2 * l32, ldb - 2)) * 4);
    const unsigned sb_row = (unsigned)(ldb * 4);
#define KC_LOAD4(dst, rsrc, vo, so) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(dst) : "v"(vo), "s"(rsrc), "s"(so))
#define KC_LOAD2(dst, rsrc, vo, so) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "+v"(dst) : "v"(vo), "s"(rsrc), "s"(so))
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    f32x4 fa[D][2];
    f32x4 fbk[B_KC ? D : 1][2];         // NT: B fragments like A's
    f32x2 fbm[B_KC ? 1 : D][4];         // NN: per step one 8-byte load (two columns)
    constexpr int LOADS = B_KC ? 4 : 6;      // per unit
    auto unit_loads = [&](int j, unsigned kq) {      // kq: k offset (floats, half 0) of the unit: kb + 4 q
        KC_LOAD4(fa[j][0], ra_src, voa[0], kq * 4u);
        KC_LOAD4(fa[j][1], ra_src, voa[1], kq * 4u);
        if (B_KC) {
            KC_LOAD4(fbk[B_KC ? j : 0][0], rb_src, vob[0], kq * 4u);
            KC_LOAD4(fbk[B_KC ? j : 0][1], rb_src, vob[1], kq * 4u);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) KC_LOAD2(fbm[B_KC ? 0 : j][c], rb_src, vob_mc, (kq + (unsigned)c) * sb_row);
        }
    };
#pragma unroll
    for (int j = 0; j < D; ++j) {
        fa[j][0] = fa[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (B_KC) fbk[B_KC ? j : 0][0] = fbk[B_KC ? j : 0][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        else
#pragma unroll
            for (int c = 0; c < 4; ++c) fbm[B_KC ? 0 : j][c] = (f32x2){0.f, 0.f};
        unit_loads(j, (unsigned)(32 * (j / 4) + 4 * (j % 4)));
    }
    // Unit j is reloaded (for the next trip of the ring) as soon as its MFMAs have been issued.  (Reloading the four units of a
    // 32-k block together -- the four 16-byte pieces of a line back to back -- measured 5 % slower.)
    const int ktrips = (kend - kbeg) / (8 * D);      // (a unit holds 8 k: 4 from each half of the wave)
    unsigned kb = 8 * D;            // k offset of the trip whose loads are issued next
    for (int trip = 0; trip < ktrips; ++trip) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (B_KC)
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(fa[j][0]), "+v"(fa[j][1]), "+v"(fbk[B_KC ? j : 0][0]), "+v"(fbk[B_KC ? j : 0][1])
                             : "n"(LOADS * D - LOADS));
            else
                asm volatile("s_waitcnt vmcnt(%6)" : "+v"(fa[j][0]), "+v"(fa[j][1]), "+v"(fbm[B_KC ? 0 : j][0]), "+v"(fbm[B_KC ? 0 : j][1]),
                             "+v"(fbm[B_KC ? 0 : j][2]), "+v"(fbm[B_KC ? 0 : j][3]) : "n"(LOADS * D - LOADS));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float a0 = fa[j][0][c], a1 = fa[j][1][c];
                const float b0 = B_KC ? fbk[B_KC ? j : 0][0][c] : fbm[B_KC ? 0 : j][c][0];
                const float b1 = B_KC ? fbk[B_KC ? j : 0][1][c] : fbm[B_KC ? 0 : j][c][1];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            unit_loads(j, kb + (unsigned)(32 * (j / 4) + 4 * (j % 4)));      // (past the end: the next row or zeros, never used)
            __builtin_amdgcn_sched_barrier(0);
        }
        kb += 8 * D;
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
        if (B_KC) asm volatile("s_waitcnt vmcnt(0)" : "+v"(fa[j][0]), "+v"(fa[j][1]), "+v"(fbk[B_KC ? j : 0][0]), "+v"(fbk[B_KC ? j : 0][1]));
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(fa[j][0]), "+v"(fa[j][1]), "+v"(fbm[B_KC ? 0 : j][0]), "+v"(fbm[B_KC ? 0 : j][1]),
                          "+v"(fbm[B_KC ? 0 : j][2]), "+v"(fbm[B_KC ? 0 : j][3]));
    }
#undef KC_LOAD4
#undef KC_LOAD2
    const bool add_bias = g.bias != nullptr && split == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + (B_KC ? j * 32 + l32 : 2 * l32 + j);
            if (col >= g.N) continue;
            const float bv = add_bias ? g.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row >= g.M) continue;
                float* c = g.C + (size_t)row * g.ldc + col;
                const float v = acc[i][j][r] + bv;
                if (g.atomic) unsafeAtomicAdd(c, v);
                else *c = v;
            }
        }
}

}  // namespace amdspeech
