// CTC loss + gradient and greedy decode for gfx950 (replaces tf.nn.ctc_loss with
// ignore_longer_outputs_than_inputs=True and its gradient,
// /root/reference/models/AcousticModel.py:356-357, the label sparsification at
// :155-159, and stands in for the decoder at :312).
//
// Design: the alpha/beta recursions are a length-T dependent chain over S=2U+1
// states per utterance -- latency bound, HBM traffic tiny.  One 64-lane wavefront
// per (utterance, direction) keeps all states in registers (blocked: lane owns R
// consecutive states, so the s-1 / s-2 neighbours are in-lane except at the block
// edge, which takes two cross-lane moves per frame); no LDS, no barriers.  The
// alpha and beta waves of one utterance run concurrently (grid = B x 2).  log-softmax
// and the posterior -> dlogits pass are separate, fully parallel, HBM-streaming
// kernels over the T*B rows.
#include "common.h"
#include "ctc_core.h"

namespace amdspeech {

// ---- label preparation: one wave per utterance --------------------------------
__global__ __launch_bounds__(64) void ctc_prepare_kernel(const int* __restrict__ dense, const int* __restrict__ lengths,
                                                         int T, int U, int C, int smax, int* __restrict__ ext,
                                                         int* __restrict__ slen, int* __restrict__ valid) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int blank = C - 1;
    int* e = ext + (size_t)b * smax;
    for (int s = lane; s < smax; s += 64) e[s] = blank;
    __syncthreads();
    int kept = 0, ntgt = 0;
    bool finished = false;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int u0 = 0; u0 < U; u0 += 64) {
        const int u = u0 + lane;
        const int v = (u < U) ? dense[(size_t)b * U + u] : 0;
        const bool nz = v != 0;
        const unsigned long long mnz = __ballot(nz);
        const unsigned long long mterm = __ballot(nz && v >= blank);
        kept += __popcll(mnz);
        if (!finished) {
            unsigned long long before = ~0ull;
            if (mterm) { before = (1ull << (__ffsll((long long)mterm) - 1)) - 1ull; }
            const unsigned long long mt = mnz & before;           // targets in this chunk
            if (nz && ((mt >> lane) & 1ull)) e[2 * (ntgt + __popcll(mt & lt)) + 1] = v;
            ntgt += __popcll(mt);
            if (mterm) finished = true;
        }
    }
    if (lane == 0) {
        if (kept == 0) kept = 1;                      // sparse_fill_empty_rows -> [C-1]
        const int len = min(lengths[b], T);   // the reference can hand over untruncated lengths > T_max
        slen[b] = 2 * ntgt + 1;
        valid[b] = (len > 0 && kept <= len) ? 1 : 0;  // required_time = raw (kept) label count
    }
}

// ---- log-softmax over C, one wave per row -----------------------------------------
__global__ __launch_bounds__(256) void log_softmax_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          long rows, int C) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + row * C;
    const float lse = ctc_row_lse(xr, C, lane);
    for (int c = lane; c < C; c += 64) y[row * C + c] = xr[c] - lse;
}

// ---- alpha / beta: grid (B, 2), NW waves per (utterance, direction) -----------------------------
// The recursion over frames is a dependent chain whose cost per frame is R log-sum-exps per thread
// (3 v_exp_f32 + 1 v_log_f32 each, quarter rate) plus one neighbour exchange, so the states of one
// utterance are spread over NW = 4 waves (one per SIMD of a CU: 4x the transcendental rate, R = ceil(S/256)
// states per thread; 2 for the 161-label targets of the benchmark config).  Thread i owns the
// adjacent states i*R .. i*R+R-1; a step needs the previous frame's states s-1, s-2 (alpha) or the next
// frame's s+1, s+2 (beta): in-thread for the inner ones, from thread i-1 / i+1 for the block edges.  The
// edge values go through a parity-double-buffered LDS array with ONE workgroup barrier per frame.
// The per-frame label gathers come from L2/HBM (~1 us away) and are prefetched a block of PF frames
// ahead into a second register set.
// The per-frame barrier orders the LDS edge exchange ONLY: __syncthreads() would also drain vmcnt, i.e. wait at every frame
// for the label gathers prefetched a block ahead and for the alpha / beta rows just stored.
__device__ __forceinline__ void ctc_frame_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <int RMAX, int PF, int NW>
__global__ __launch_bounds__(NW * 64) void ctc_alpha_beta_kernel(const float* __restrict__ logp, const int* __restrict__ ext,
                                                                 const int* __restrict__ slen, const int* __restrict__ valid,
                                                                 const int* __restrict__ lengths, int T, int B, int C,
                                                                 int smax, float* __restrict__ alpha,
                                                                 float* __restrict__ beta, float* __restrict__ ll) {
    static_assert(RMAX >= 2, "RMAX >= 2");
    constexpr int NT = NW * 64;
    __shared__ float2 edge[2][NT + 4];       // [parity][2 + thread]: pads of NEG_INF at both ends
    __shared__ float fin[NT];
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    if (!valid[b]) { if (dir == 0 && tid == 0) ll[b] = 0.f; return; }
    const int S = slen[b];
    const int Tb = min(lengths[b], T);
    const int R = (S + NT - 1) / NT;             // states per thread actually used (<= RMAX)
    const int blank = C - 1;
    const int* e = ext + (size_t)b * smax;
    int lab[RMAX]; bool skip[RMAX]; bool act[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int s = tid * R + r;
        act[r] = r < R && s < S;
        lab[r] = act[r] ? e[s] : blank;
        if (dir == 0) skip[r] = act[r] && s >= 2 && lab[r] != blank && lab[r] != e[s - 2];
        else skip[r] = act[r] && s + 2 < S && e[s + 2] != blank && e[s + 2] != lab[r];
    }
    if (tid < 4) {
        const int slot = tid < 2 ? tid : NT + tid;            // 0, 1, NT+2, NT+3
        edge[0][slot] = make_float2(NEG_INF, NEG_INF);
        edge[1][slot] = make_float2(NEG_INF, NEG_INF);
    }
    const size_t rowstride = (size_t)B * C;
    const float* lp = logp + (size_t)b * C;
    float* out = (dir == 0 ? alpha : beta) + (size_t)b * T * smax;

    float cur[RMAX];
    // ---- step 0
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int s = tid * R + r;
        if (dir == 0) cur[r] = (act[r] && s < 2) ? lp[lab[r]] * LOG2E : NEG_INF;          // alpha_0
        else cur[r] = (act[r] && (s == S - 1 || s == S - 2)) ? 0.f : NEG_INF;               // beta_{Tb-1} (excludes y_t)
        if (act[r]) out[(size_t)(dir == 0 ? 0 : Tb - 1) * smax + s] = cur[r] * LN2;
    }

    auto load_block = [&](int i0, float (&buf)[PF][RMAX]) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int i = min(i0 + q, Tb - 1);                  // clamped: loads stay unconditional
            const int fr = dir == 0 ? i : Tb - i;               // (i >= 1, so fr <= Tb - 1)
#pragma unroll
            for (int r = 0; r < RMAX; ++r) buf[q][r] = lp[(size_t)fr * rowstride + lab[r]] * LOG2E;
        }
    };
    auto step = [&](int i, const float (&lpv)[RMAX]) {
        float newv[RMAX];
        float2* ed = edge[i & 1] + 2;                           // ed[thread]
        if (dir == 0) {
            // publish this thread's last two states of the previous frame: (s_last, s_last - 1)
            float last1 = cur[0], last2 = NEG_INF;
#pragma unroll
            for (int r = 1; r < RMAX; ++r) if (r < R) { last2 = last1; last1 = cur[r]; }
            ed[tid] = make_float2(last1, last2);
            ctc_frame_barrier();
            const float2 n1 = ed[tid - 1];
            float p1 = n1.x;                                      // alpha_{t-1}(s-1) for r = 0
            float p2 = (R == 1) ? ed[tid - 2].x : n1.y;           // alpha_{t-1}(s-2) for r = 0
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
                if (r < R) {
                    const float v = lse3_2(cur[r], p1, skip[r] ? p2 : NEG_INF) + lpv[r];
                    newv[r] = act[r] ? v : NEG_INF;
                    p2 = p1; p1 = cur[r];
                } else newv[r] = NEG_INF;
            }
        } else {
            float nb[RMAX];   // beta_{t+1}(s) + logp_{t+1}(l'_s)
#pragma unroll
            for (int r = 0; r < RMAX; ++r) nb[r] = act[r] ? cur[r] + lpv[r] : NEG_INF;
            // publish this thread's first two states: (s_first, s_first + 1)
            ed[tid] = make_float2(nb[0], R >= 2 ? nb[1] : NEG_INF);
            ctc_frame_barrier();
            const float2 m1 = ed[tid + 1];
            const float dn1 = m1.x;
            const float dn2 = (R == 1) ? ed[tid + 2].x : m1.y;
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
                if (r < R) {
                    // neighbours s+1, s+2: in-thread while r+1 / r+2 < R, else thread+1's states 0 / 1
                    const float in1 = nb[(r + 1 < RMAX) ? r + 1 : 0];
                    const float in2 = nb[(r + 2 < RMAX) ? r + 2 : 0];
                    const float n1 = (r + 1 < R) ? in1 : dn1;
                    const float n2 = (r + 2 < R) ? in2 : ((r + 1 < R) ? dn1 : dn2);
                    const float v = lse3_2(nb[r], n1, skip[r] ? n2 : NEG_INF);
                    newv[r] = act[r] ? v : NEG_INF;
                } else newv[r] = NEG_INF;
            }
        }
        float* o = out + (size_t)(dir == 0 ? i : Tb - 1 - i) * smax;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) { cur[r] = newv[r]; if (act[r]) o[tid * R + r] = cur[r] * LN2; }
    };

    if (Tb > 1) {
        float bufA[PF][RMAX], bufB[PF][RMAX];
        load_block(1, bufA);
        for (int i0 = 1; i0 < Tb; i0 += 2 * PF) {
            load_block(i0 + PF, bufB);
#pragma unroll
            for (int q = 0; q < PF; ++q) if (i0 + q < Tb) step(i0 + q, bufA[q]);
            load_block(i0 + 2 * PF, bufA);
#pragma unroll
            for (int q = 0; q < PF; ++q) if (i0 + PF + q < Tb) step(i0 + PF + q, bufB[q]);
        }
    }
    if (dir == 0) {
        // log p(l|x) = lse(alpha_{Tb-1}(S-1), alpha_{Tb-1}(S-2))
        float mine = NEG_INF;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const int s = tid * R + r;
            if (act[r] && (s == S - 1 || s == S - 2)) mine = lse3_2(mine, cur[r], NEG_INF);
        }
        fin[tid] = mine;
        __syncthreads();
        if (tid < 64) {
            float tot = NEG_INF;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot = lse3_2(tot, fin[w * 64 + tid], NEG_INF);
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) tot = lse3_2(tot, __shfl_xor(tot, o2), NEG_INF);
            if (tid == 0) ll[b] = tot * LN2;
        }
    }
}

// ---- alpha / beta, TWO frames per exchange (S <= 512: 256 threads x 2 adjacent states) -------------------------------
// A frame of ctc_alpha_beta_kernel is ~800 clocks of which more than half are the LDS round trip and the barrier of the edge
// exchange.  Here a thread also fetches the FOUR states below its own two (threads i-1 and i-2), recomputes
// the two-state halo of the first frame itself and so advances two frames per exchange: 6 log-sum-exps instead of 4, one
// barrier and one LDS round trip instead of two.
// Both directions run the SAME code: beta in reversed coordinates (state r = S-1-s, frames from the end) obeys alpha's
// recursion on gamma_t(s) = beta_t(s) + log p_t(l_s) -- the skip condition is symmetric in (s, s+2) -- and beta itself is the
// log-sum-exp before the emission is added, so nothing is subtracted.
template <int PF>      // frames per prefetch block (even)
__global__ __launch_bounds__(256) void ctc_alpha_beta2_kernel(const float* __restrict__ logp, const int* __restrict__ ext,
                                                              const int* __restrict__ slen, const int* __restrict__ valid,
                                                              const int* __restrict__ lengths, int T, int B, int C,
                                                              int smax, float* __restrict__ alpha,
                                                              float* __restrict__ beta, float* __restrict__ ll) {
    static_assert(PF % 2 == 0, "PF even");
    constexpr int NT = 256;
    __shared__ float2 edge[2][NT + 2];       // [parity][2 + thread]: two pads of NEG_INF in front
    __shared__ float fin[NT];
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    if (!valid[b]) { if (dir == 0 && tid == 0) ll[b] = 0.f; return; }
    const int S = slen[b];
    const int Tb = min(lengths[b], T);
    const int blank = C - 1;
    const int* e = ext + (size_t)b * smax;
    // states r = 2 tid - 2 .. 2 tid + 1 in recursion coordinates (k = 0, 1: the halo; k = 2, 3: this thread's own)
    int lab[4]; bool skip[4]; bool act[4]; int sidx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = 2 * tid - 2 + k;
        act[k] = r >= 0 && r < S;
        sidx[k] = dir == 0 ? r : S - 1 - r;
        lab[k] = act[k] ? e[sidx[k]] : blank;
        skip[k] = act[k] && r >= 2 && lab[k] != blank && lab[k] != e[dir == 0 ? r - 2 : S + 1 - r];
    }
    if (tid < 2) { edge[0][tid] = make_float2(NEG_INF, NEG_INF); edge[1][tid] = make_float2(NEG_INF, NEG_INF); }
    const size_t rowstride = (size_t)B * C;
    const float* lp = logp + (size_t)b * C;
    float* out = (dir == 0 ? alpha : beta) + (size_t)b * T * smax;
    auto frame = [&](int i) { return dir == 0 ? i : Tb - 1 - i; };   // recursion step -> frame

    // ---- step 0
    float cur0, cur1;
    {
        const float* l0 = lp + (size_t)frame(0) * rowstride;
        cur0 = (act[2] && 2 * tid < 2) ? l0[lab[2]] * LOG2E : NEG_INF;
        cur1 = (act[3] && 2 * tid + 1 < 2) ? l0[lab[3]] * LOG2E : NEG_INF;
        float* o = out + (size_t)frame(0) * smax;
        if (act[2]) o[sidx[2]] = dir == 0 ? cur0 * LN2 : (2 * tid < 2 ? 0.f : NEG_INF);
        if (act[3]) o[sidx[3]] = dir == 0 ? cur1 * LN2 : (2 * tid + 1 < 2 ? 0.f : NEG_INF);
    }
    // emissions of a block of PF steps: own two states every step, the halo's two for the first step of every pair
    auto load_block = [&](int i0, float (&own)[PF][2], float (&halo)[PF / 2][2]) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const float* row = lp + (size_t)frame(min(i0 + q, Tb - 1)) * rowstride;      // clamped: loads stay unconditional
#if CTC_DIAG == 2
            own[q][0] = -3.f - (float)(i0 & 1); own[q][1] = -4.f;
            if ((q & 1) == 0) { halo[q >> 1][0] = -3.f; halo[q >> 1][1] = -4.f; }
            (void)row;
#else
            own[q][0] = row[lab[2]]; own[q][1] = row[lab[3]];
            if ((q & 1) == 0) { halo[q >> 1][0] = row[lab[0]]; halo[q >> 1][1] = row[lab[1]]; }
#endif
        }
    };
    int par = 0;
    // steps i and i+1 (the second only if it exists)
    auto pair = [&](int i, const float (&o0)[2], const float (&o1)[2], const float (&h)[2]) {
#if CTC_DIAG == 3
        const float2 e1 = make_float2(cur1 - 1.f, cur0 - 2.f), e2 = make_float2(cur0 - 3.f, cur1 - 1.5f);
#else
        edge[par][2 + tid] = make_float2(cur0, cur1);
        ctc_frame_barrier();
        const float2 e1 = edge[par][2 + tid - 1], e2 = edge[par][2 + tid - 2];      // states 2tid-2, 2tid-1 and 2tid-4, 2tid-3
#endif
        par ^= 1;
        // step i: the halo's two states and this thread's two
        const float vh0 = lse3_2(e1.x, e2.y, skip[0] ? e2.x : NEG_INF);
        const float vh1 = lse3_2(e1.y, e1.x, skip[1] ? e2.y : NEG_INF);
        const float v0 = lse3_2(cur0, e1.y, skip[2] ? e1.x : NEG_INF);
        const float v1 = lse3_2(cur1, cur0, skip[3] ? e1.y : NEG_INF);
        const float hn0 = act[0] ? vh0 + h[0] * LOG2E : NEG_INF;
        const float hn1 = act[1] ? vh1 + h[1] * LOG2E : NEG_INF;
        const float n0 = act[2] ? v0 + o0[0] * LOG2E : NEG_INF;
        const float n1 = act[3] ? v1 + o0[1] * LOG2E : NEG_INF;
#if CTC_DIAG != 1
        {
            float* o = out + (size_t)frame(i) * smax;
            if (act[2]) o[sidx[2]] = (dir == 0 ? n0 : v0) * LN2;
            if (act[3]) o[sidx[3]] = (dir == 0 ? n1 : v1) * LN2;
        }
#endif
        cur0 = n0; cur1 = n1;
        if (i + 1 < Tb) {
            const float w0 = lse3_2(n0, hn1, skip[2] ? hn0 : NEG_INF);
            const float w1 = lse3_2(n1, n0, skip[3] ? hn1 : NEG_INF);
            const float m0 = act[2] ? w0 + o1[0] * LOG2E : NEG_INF;
            const float m1 = act[3] ? w1 + o1[1] * LOG2E : NEG_INF;
#if CTC_DIAG != 1
            float* o = out + (size_t)frame(i + 1) * smax;
            if (act[2]) o[sidx[2]] = (dir == 0 ? m0 : w0) * LN2;
            if (act[3]) o[sidx[3]] = (dir == 0 ? m1 : w1) * LN2;
#endif
            cur0 = m0; cur1 = m1;
        }
    };
    if (Tb > 1) {
        float ownA[PF][2], ownB[PF][2], haloA[PF / 2][2], haloB[PF / 2][2];
        load_block(1, ownA, haloA);
        for (int i0 = 1; i0 < Tb; i0 += 2 * PF) {
            load_block(i0 + PF, ownB, haloB);
#pragma unroll
            for (int q = 0; q < PF; q += 2) if (i0 + q < Tb) pair(i0 + q, ownA[q], ownA[q + 1], haloA[q >> 1]);
            load_block(i0 + 2 * PF, ownA, haloA);
#pragma unroll
            for (int q = 0; q < PF; q += 2) if (i0 + PF + q < Tb) pair(i0 + PF + q, ownB[q], ownB[q + 1], haloB[q >> 1]);
        }
    }
    if (dir == 0) {
        // log p(l|x) = lse(alpha_{Tb-1}(S-1), alpha_{Tb-1}(S-2))
        float mine = NEG_INF;
        if (act[2] && (2 * tid == S - 1 || 2 * tid == S - 2)) mine = lse3_2(mine, cur0, NEG_INF);
        if (act[3] && (2 * tid + 1 == S - 1 || 2 * tid + 1 == S - 2)) mine = lse3_2(mine, cur1, NEG_INF);
        fin[tid] = mine;
        __syncthreads();
        if (tid < 64) {
            float tot = NEG_INF;
#pragma unroll
            for (int w = 0; w < 4; ++w) tot = lse3_2(tot, fin[w * 64 + tid], NEG_INF);
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) tot = lse3_2(tot, __shfl_xor(tot, o2), NEG_INF);
            if (tid == 0) ll[b] = tot * LN2;
        }
    }
}

// Round 3.  What a frame of the kernel above costs (ablations at T = 1001, B = 32, S = 323: 297 us for the recursion): the LDS
// round trip + workgroup barrier of the edge exchange 35 %, the transcendentals 31 % (the two-frames-per-exchange halo makes it
// 3 log-sum-exps per frame and thread, and only 162 of the 256 threads own states), the emission gathers 19 %, the stores 6 %.
// Here the exchange leaves the frame loop: a thread's two states need only the two states of the thread BELOW it, and inside a
// wave that is one DPP wave-shift per value -- no LDS, no barrier.  Across waves the windows OVERLAP: a wave holds 128
// consecutive states, the lowest 32 of them (16 lanes) copies of the previous wave's highest 32.  The copies are recomputed along
// with everything else; what is wrong about them (lane 0 has nobody below it) creeps up one lane per frame, so after 16 frames
// lanes 0 .. 15 are stale and lanes 16 .. 63 -- the states the wave owns and stores -- still exact.  Every 16 frames the copies
// are refreshed through LDS (one barrier per 16 frames instead of one per 2).  2 log-sum-exps per frame and thread, four busy
// waves (96 owned states each: S <= 384).
template <int PF>      // frames per prefetch block; the refresh period is 2 PF = 16
__global__ __launch_bounds__(256) void ctc_alpha_beta3_kernel(const float* __restrict__ logp, const int* __restrict__ ext,
                                                              const int* __restrict__ slen, const int* __restrict__ valid,
                                                              const int* __restrict__ lengths, int T, int B, int C,
                                                              int smax, float* __restrict__ alpha,
                                                              float* __restrict__ beta, float* __restrict__ ll) {
    static_assert(2 * PF == 16, "the copies of the previous wave's states last 16 frames");
    constexpr int OWN = 96, HALO = 32;
    __shared__ double2 edge[2][4][16];
    __shared__ double fin[256];
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    if (!valid[b]) { if (dir == 0 && tid == 0) ll[b] = 0.f; return; }
    const int S = slen[b];
    const int Tb = min(lengths[b], T);
    const int blank = C - 1;
    const int* e = ext + (size_t)b * smax;
    constexpr double NEG_INF_D = -__builtin_inf();
    // states r0 (EVEN: a blank of the extended target -- no skip transition, and its emission is the same for every lane) and
    // r0 + 1 (a label) in recursion coordinates (beta: reversed, see ctc_alpha_beta2_kernel; S is odd, so parity survives)
    const int r0 = w * OWN - HALO + 2 * lane;
    const bool act0 = r0 >= 0 && r0 < S, act1 = r0 + 1 >= 0 && r0 + 1 < S;
    const int sidx0 = dir == 0 ? r0 : S - 1 - r0, sidx1 = dir == 0 ? r0 + 1 : S - 2 - r0;
    const int lab1 = act1 ? e[sidx1] : blank;
    const bool skip1 = act1 && r0 + 1 >= 2 && lab1 != blank && lab1 != e[dir == 0 ? r0 - 1 : S - r0];
    const bool st0 = act0 && lane >= 16, st1 = act1 && lane >= 16;      // owned (stored) states; wave 0's copies are r < 0
    const size_t rowstride = (size_t)B * C;
    const float* lp = logp + (size_t)b * C;
    float* out = (dir == 0 ? alpha : beta) + (size_t)b * T * smax;
    // stores without branches: lanes that own nothing get an offset past the descriptor
    const auto r_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, (unsigned)((size_t)T * smax * 4), 0x00020000);
    const unsigned so0 = st0 ? (unsigned)sidx0 * 4u : 0x80000000u, so1 = st1 ? (unsigned)sidx1 * 4u : 0x80000000u;
    auto put = [&](int fr, float a0, float a1) __attribute__((always_inline)) {
        const unsigned row = (unsigned)fr * (unsigned)smax * 4u;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a0), r_out, so0 + (st0 ? row : 0u), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a1), r_out, so1 + (st1 ? row : 0u), 0, 0);
    };
    auto frame = [&](int i) { return dir == 0 ? i : Tb - 1 - i; };

    double cur0, cur1;
    {
        const float* l0 = lp + (size_t)frame(0) * rowstride;
        cur0 = (act0 && r0 < 2) ? (double)(l0[blank] * LOG2E) : NEG_INF_D;
        cur1 = (act1 && r0 + 1 < 2) ? (double)(l0[lab1] * LOG2E) : NEG_INF_D;
        put(frame(0), dir == 0 ? (float)cur0 * LN2 : (r0 < 2 ? 0.f : NEG_INF), dir == 0 ? (float)cur1 * LN2 : (r0 + 1 < 2 ? 0.f : NEG_INF));
    }
    // emissions of a block of PF frames: the label's by a gather, the blank's from a wave-uniform address (a scalar load)
    auto load_block = [&](int i0, float (&em)[PF][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const float* row = lp + (size_t)frame(min(i0 + q, Tb - 1)) * rowstride;      // clamped: loads stay unconditional
            em[q][0] = row[blank]; em[q][1] = row[lab1];
        }
    };
    auto step = [&](int i, const float (&em)[2]) __attribute__((always_inline)) {
        const double below1 = ctc_from_lane_below(cur1);      // state r0 - 1 (the lane below's label state)
        const double v0 = lse2_2d(cur0, below1);
        const double v1 = lse3_2d(cur1, cur0, skip1 ? below1 : NEG_INF_D);
        const double n0 = act0 ? v0 + (double)(em[0] * LOG2E) : NEG_INF_D;
        const double n1 = act1 ? v1 + (double)(em[1] * LOG2E) : NEG_INF_D;
        // (what the gradient kernel reads back is rounded ONCE, here: 3e-5 of an exponent, not a thousand roundings of it)
        put(frame(i), (float)((dir == 0 ? n0 : v0) * (double)LN2), (float)((dir == 0 ? n1 : v1) * (double)LN2));
        cur0 = n0; cur1 = n1;
    };
    int par = 0;
    auto refresh = [&]() __attribute__((always_inline)) {             // the previous wave's highest 32 states -> this wave's lanes 0 .. 15
        if (lane >= 48) edge[par][w][lane - 48] = make_double2(cur0, cur1);
        ctc_frame_barrier();
        if (lane < 16 && w > 0) { const double2 v = edge[par][w - 1][lane]; cur0 = v.x; cur1 = v.y; }
        par ^= 1;                      // (the slot is rewritten two refreshes later: one barrier in between)
    };
    if (Tb > 1) {
        float emA[PF][2], emB[PF][2];
        load_block(1, emA);
        int i0 = 1;
        for (; i0 + 2 * PF <= Tb; i0 += 2 * PF) {          // whole periods: straight-line code
            if (i0 > 1) refresh();
            load_block(i0 + PF, emB);
#pragma unroll
            for (int q = 0; q < PF; ++q) step(i0 + q, emA[q]);
            load_block(i0 + 2 * PF, emA);
#pragma unroll
            for (int q = 0; q < PF; ++q) step(i0 + PF + q, emB[q]);
        }
        if (i0 < Tb) {                                     // the last, partial period
            if (i0 > 1) refresh();
            load_block(i0 + PF, emB);
#pragma unroll
            for (int q = 0; q < PF; ++q) if (i0 + q < Tb) step(i0 + q, emA[q]);
#pragma unroll
            for (int q = 0; q < PF; ++q) if (i0 + PF + q < Tb) step(i0 + PF + q, emB[q]);
        }
    }
    if (dir == 0) {
        double mine = NEG_INF_D;
        if (st0 && (r0 == S - 1 || r0 == S - 2)) mine = lse2_2d(mine, cur0);
        if (st1 && (r0 + 1 == S - 1 || r0 + 1 == S - 2)) mine = lse2_2d(mine, cur1);
        fin[tid] = mine;
        __syncthreads();
        if (tid < 64) {
            double tot = NEG_INF_D;
#pragma unroll
            for (int k = 0; k < 4; ++k) tot = lse2_2d(tot, fin[k * 64 + tid]);
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) tot = lse2_2d(tot, __shfl_xor(tot, o2));
            if (tid == 0) ll[b] = (float)(tot * (double)LN2);
        }
    }
}

// ---- dlogits = softmax - posterior, one wave per (t, b) row -------------------------
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ logp, const float* __restrict__ alpha,
                                                       const float* __restrict__ beta, const int* __restrict__ ext,
                                                       const int* __restrict__ slen, const int* __restrict__ valid,
                                                       const int* __restrict__ lengths, const float* __restrict__ ll,
                                                       int T, int B, int C, int smax, float* __restrict__ dlogits,
                                                       float* __restrict__ loss) {
    extern __shared__ float occ_all[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + w;
    float* occ = occ_all + w * C;
    if (row >= (long)T * B) return;
    const int t = row / B, b = row % B;
    float* g = dlogits + row * C;
    const bool ok = valid[b] != 0;
    if (t == 0 && lane == 0) loss[b] = ok ? -ll[b] : 0.f;
    if (!ok || t >= lengths[b]) {
        for (int c = lane; c < C; c += 64) g[c] = 0.f;
        return;
    }
    const float* lp = logp + row * C;
    const float llb = ll[b];
    if (llb == NEG_INF) {                 // TF: "No valid path found" -> gradient = softmax
        for (int c = lane; c < C; c += 64) g[c] = expf(lp[c]);
        return;
    }
    for (int c = lane; c < C; c += 64) occ[c] = 0.f;
    __builtin_amdgcn_wave_barrier();
    const int S = slen[b];
    const float* al = alpha + ((size_t)b * T + t) * smax;
    const float* be = beta + ((size_t)b * T + t) * smax;
    const int* e = ext + (size_t)b * smax;
    // every even state of the extended target is the blank: those 162 of 323 posteriors would all be LDS atomics on ONE address
    // (fully serialised, and LDS float atomics run at about a lane per clock anyway) -- they are summed in registers instead
    float blank_occ = 0.f;
    for (int s = lane; s < S; s += 64) {
        // (alpha + beta - log p cancels numbers of magnitude 10^3 down to O(1): in float the two adds alone lose 2e-4)
        const float p = expf((float)((double)al[s] + (double)be[s] - (double)llb));
        if ((s & 1) == 0) blank_occ += p;
        else if (p > 0.f) atomicAdd(&occ[e[s]], p);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) blank_occ += __shfl_xor(blank_occ, o);
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (lane == 0) occ[C - 1] += blank_occ;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    for (int c = lane; c < C; c += 64) g[c] = expf(lp[c]) - occ[c];
}

// ---- greedy decode -----------------------------------------------------------
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, long rows, int C,
                                                          int* __restrict__ best) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + row * C;
    float bv = NEG_INF; int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        const float v = xr[c];
        if (v > bv || (v == bv && c < bi)) { bv = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) best[row] = bi;
}

__global__ __launch_bounds__(256) void collapse_kernel(const int* __restrict__ best, const int* __restrict__ lengths,
                                                       int T, int B, int C, int* __restrict__ ids,
                                                       int* __restrict__ out_len) {
    __shared__ int wave_tot[4];
    __shared__ int base_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int blank = C - 1;
    const int Tb = min(lengths[b], T);
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int t0 = 0; t0 < T; t0 += 256) {
        const int t = t0 + tid;
        bool keep = false; int k = blank;
        if (t < Tb) {
            k = best[(size_t)t * B + b];
            const int prev = t > 0 ? best[(size_t)(t - 1) * B + b] : -1;
            keep = k != blank && k != prev;
        }
        const unsigned long long m = __ballot(keep);
        const int pre = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[w] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int i = 0; i < w; ++i) off += wave_tot[i];
        if (keep) ids[(size_t)b * T + off + pre] = k;
        __syncthreads();
        if (tid == 0) base_s += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    const int n = base_s;
    for (int i = n + tid; i < T; i += 256) ids[(size_t)b * T + i] = C;
    if (tid == 0) out_len[b] = n;
}

// ---- merge_repeated: collapse consecutive duplicate labels of each decoded row, in place --------
// (TensorFlow's ctc_beam_search_decoder default post-processing of the top path, see beam.cpp)
__global__ __launch_bounds__(256) void merge_repeated_kernel(int* __restrict__ ids, int* __restrict__ lens, int T, int pad) {
    extern __shared__ int row[];                 // [T] copy of the input row
    __shared__ int wave_tot[4];
    __shared__ int base_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int* r = ids + (size_t)b * T;
    const int n = lens[b];
    for (int i = tid; i < n; i += 256) row[i] = r[i];
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        const bool keep = i < n && (i == 0 || row[i] != row[i - 1]);
        const unsigned long long m = __ballot(keep);
        const int pre = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[w] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int q = 0; q < w; ++q) off += wave_tot[q];
        if (keep) r[off + pre] = row[i];
        __syncthreads();
        if (tid == 0) base_s += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    const int kept = base_s;
    for (int i = kept + tid; i < n; i += 256) r[i] = pad;
    if (tid == 0) lens[b] = kept;
}

// ---- Levenshtein distance per utterance (replaces tf.edit_distance, AcousticModel.py:370) --------
// One wave per pair.  Row i of the DP table lives in LDS; within a row
//   cur[j] = min(cand[j], cur[j-1] + 1),  cand[j] = min(prev[j-1] + (a_i != b_j), prev[j] + 1)
// is a prefix minimum of cand[k] - k, done with a wave scan per 64-column chunk plus a carry.
__global__ __launch_bounds__(64) void edit_distance_kernel(const int* __restrict__ a, const int* __restrict__ alen, int lda,
                                                          const int* __restrict__ bb, const int* __restrict__ blen, int ldb,
                                                          int* __restrict__ out) {
    extern __shared__ int prev[];                // [m + 1]
    const int p = blockIdx.x, lane = threadIdx.x;
    const int n = alen[p], m = blen[p];
    const int* ar = a + (size_t)p * lda;
    const int* br = bb + (size_t)p * ldb;
    for (int j = lane; j <= m; j += 64) prev[j] = j;
    __builtin_amdgcn_wave_barrier();
    for (int i = 1; i <= n; ++i) {
        const int ai = ar[i - 1];
        int carry = i;                           // cur[0] = i; running prefix-min of (cur[k] - k) over finished columns
        int left_prev = prev[0];                 // prev[j-1] for the first column of the chunk
        if (lane == 0) prev[0] = i;
        for (int j0 = 1; j0 <= m; j0 += 64) {
            const int j = j0 + lane;
            const bool ok = j <= m;
            const int pj = ok ? prev[j] : 0;
            int pjm1 = __shfl_up(pj, 1);
            if (lane == 0) pjm1 = left_prev;
            left_prev = __shfl(pj, 63);
            int cand = ok ? min(pjm1 + (br[j - 1] != ai ? 1 : 0), pj + 1) : 0x3fffffff;
            int v = ok ? cand - j : 0x3fffffff;  // prefix-min trick
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(v, o);
                if (lane >= o) v = min(v, u);
            }
            v = min(v, carry);                    // carry already holds min over k < j0 of (cur[k] - k)
            if (ok) prev[j] = v + j;
            carry = min(carry, __shfl(v, 63));
        }
        __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) out[p] = prev[m];
}

}  // namespace amdspeech

using namespace amdspeech;

extern "C" int amdspeech_merge_repeated(void* stream, int* ids, int* lens, int T, int B, int pad) {
    AS_CHECK_ARG(ids && lens && T > 0 && B > 0, "merge_repeated: bad arguments");
    AS_CHECK_ARG((size_t)T * 4 <= 60 * 1024, "merge_repeated: T=%d too long for the LDS row", T);
    hipLaunchKernelGGL(merge_repeated_kernel, dim3(B), dim3(256), (size_t)T * 4, static_cast<hipStream_t>(stream), ids, lens, T, pad);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_edit_distance(void* stream, const int* a, const int* a_len, int lda, const int* b,
                                       const int* b_len, int ldb, int n_pairs, int* out) {
    AS_CHECK_ARG(a && a_len && b && b_len && out && n_pairs > 0 && lda > 0 && ldb > 0, "edit_distance: bad arguments");
    AS_CHECK_ARG((size_t)(ldb + 1) * 4 <= 60 * 1024, "edit_distance: second sequence too long for LDS");
    hipLaunchKernelGGL(edit_distance_kernel, dim3(n_pairs), dim3(64), (size_t)(ldb + 1) * 4, static_cast<hipStream_t>(stream),
                       a, a_len, lda, b, b_len, ldb, out);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}


// the extended targets of a mini-batch into the CTC workspace (ext, slen, valid): what stage 1 of the staged call does first;
// lstm.hip's fused head (ctc_flow.h) needs them before the forward kernel starts
namespace amdspeech {
int ctc_prepare_targets(hipStream_t s, const int* dense_labels, const int* lengths, int T, int B, int C, int U, void* ws) {
    const CtcLayout lo = ctc_layout(T, B, C, U);
    char* w = static_cast<char*>(ws);
    hipLaunchKernelGGL(ctc_prepare_kernel, dim3(B), dim3(64), 0, s, dense_labels, lengths, T, U, C, lo.smax,
                       reinterpret_cast<int*>(w + lo.ext), reinterpret_cast<int*>(w + lo.slen), reinterpret_cast<int*>(w + lo.valid));
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}
}  // namespace amdspeech

extern "C" size_t amdspeech_ctc_workspace_bytes(int T, int B, int C, int U) {
    if (T <= 0 || B <= 0 || C <= 1 || U <= 0) return 0;
    return ctc_layout(T, B, C, U).total;
}

extern "C" int amdspeech_ctc_loss_fwd_bwd(void* stream, const float* logits, const int* dense_labels,
                                          const int* lengths, int T, int B, int C, int U, float* loss,
                                          float* dlogits, void* ws) {
    return amdspeech_ctc_loss_fwd_bwd_staged(stream, logits, dense_labels, lengths, T, B, C, U, loss, dlogits, ws, 0);
}

extern "C" int amdspeech_ctc_loss_fwd_bwd_staged(void* stream, const float* logits, const int* dense_labels,
                                                 const int* lengths, int T, int B, int C, int U, float* loss,
                                                 float* dlogits, void* ws, int stage) {
    AS_CHECK_ARG(stage >= 0 && stage <= 2, "ctc: stage %d", stage);
    AS_CHECK_ARG(T > 0 && B > 0 && C > 1 && U > 0, "ctc: bad shape T=%d B=%d C=%d U=%d", T, B, C, U);
    AS_CHECK_ARG(logits && dense_labels && lengths && loss && dlogits && ws, "ctc: null pointer");
    AS_CHECK_ARG(C <= 4096, "ctc: C=%d too large", C);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const CtcLayout lo = ctc_layout(T, B, C, U);
    AS_CHECK_ARG(lo.smax <= 256 * 20, "ctc: label width U=%d exceeds the supported 2559", U);
    char* w = static_cast<char*>(ws);
    float* logp = reinterpret_cast<float*>(w + lo.logp);
    float* alpha = reinterpret_cast<float*>(w + lo.alpha);
    float* beta = reinterpret_cast<float*>(w + lo.beta);
    int* ext = reinterpret_cast<int*>(w + lo.ext);
    int* slen = reinterpret_cast<int*>(w + lo.slen);
    int* valid = reinterpret_cast<int*>(w + lo.valid);
    float* ll = reinterpret_cast<float*>(w + lo.ll);
    const long rows = (long)T * B;
    if (stage != 2) {       // the chip-wide, short part: extended targets and log-softmax into the workspace
        hipLaunchKernelGGL(ctc_prepare_kernel, dim3(B), dim3(64), 0, s, dense_labels, lengths, T, U, C, lo.smax, ext, slen, valid);
        hipLaunchKernelGGL(log_softmax_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, s, logits, logp, rows, C);
        AS_CHECK_LAUNCH();
        if (stage == 1) return AMDSPEECH_OK;
    }
    // 4 waves per (utterance, direction) once the targets are long enough to feed them
    const bool wide = lo.smax > 128;
    const int rneed = ceil_div(lo.smax, wide ? 256 : 64);
    dim3 grid(B, 2);
#define LAUNCH_AB(R, PF, NW) hipLaunchKernelGGL((ctc_alpha_beta_kernel<R, PF, NW>), grid, dim3(NW * 64), 0, s, logp, ext, slen, valid, lengths, T, B, C, lo.smax, alpha, beta, ll)
    if (wide) {
        static const int two = runtime_switch("AMDSPEECH_CTC_PAIR", 1);      // 0: one frame per exchange
        static const int shift = runtime_switch("AMDSPEECH_CTC_SHIFT", 1);   // 0: the LDS-exchange kernels
        if (lo.smax <= 384 && shift)
            hipLaunchKernelGGL((ctc_alpha_beta3_kernel<8>), grid, dim3(256), 0, s, logp, ext, slen, valid, lengths, T, B, C, lo.smax, alpha, beta, ll);
        else if (rneed <= 2 && two)
            hipLaunchKernelGGL((ctc_alpha_beta2_kernel<8>), grid, dim3(256), 0, s, logp, ext, slen, valid, lengths, T, B, C, lo.smax, alpha, beta, ll);
        else if (rneed <= 2) LAUNCH_AB(2, 8, 4);
        else if (rneed <= 4) LAUNCH_AB(4, 8, 4);
        else if (rneed <= 8) LAUNCH_AB(8, 4, 4);
        else LAUNCH_AB(20, 4, 4);
    } else LAUNCH_AB(2, 8, 1);
#undef LAUNCH_AB
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(ceil_div(rows, 4)), dim3(256), 4 * C * sizeof(float), s, logp, alpha, beta,
                       ext, slen, valid, lengths, ll, T, B, C, lo.smax, dlogits, loss);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_ctc_greedy_decode(void* stream, const float* logits, const int* lengths, int T, int B,
                                           int C, int* ids, int* out_len, int* ws) {
    AS_CHECK_ARG(T > 0 && B > 0 && C > 1, "greedy: bad shape");
    AS_CHECK_ARG(logits && lengths && ids && out_len && ws, "greedy: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long rows = (long)T * B;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, s, logits, rows, C, ws);
    hipLaunchKernelGGL(collapse_kernel, dim3(B), dim3(256), 0, s, ws, lengths, T, B, C, ids, out_len);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}
