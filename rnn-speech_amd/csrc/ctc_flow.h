// The CTC head INSIDE the whole-sequence LSTM kernels (round 5): output Linear + log-softmax + alpha as a consumer that follows the
// forward recurrence frame by frame, beta + posterior + dlogits + dZ_top as a producer that runs ahead of the backward recurrence.
// Replaces, for the shapes it takes, the launches that sat between the two recurrence kernels of a training step (output Linear,
// log-softmax, alpha/beta, gradient, dlogits . W_o^T: 0.45-0.55 ms of a 12.7 ms step at the headline configuration) --
// /root/reference/models/AcousticModel.py:241-247 (output layer), :356-357 (tf.nn.ctc_loss and its gradient).
//
//   forward  (lstm_fwd_flow2): the top layer's groups also store their masked output as packed panels (the layout the layers
//     below hand to each other: written through, sentinel pre-filled).  Workgroups of the XCDs without a recurrence group that
//     are neither x-product workers nor reserved for side-stream work run ctc_follower: a TEAM of four waves per utterance
//     (two utterances per team at batch 32) polls 16 frames of panels, multiplies them with W_o on the matrix cores (K split
//     over the four waves), adds the bias, writes the logits, takes the log-softmax (the wave-per-row arithmetic of
//     log_softmax_kernel) into LDS and memory, and advances the alpha recursion of ctc_alpha_beta3_kernel by those 16 frames
//     -- emissions from LDS, no gathers.  When the recurrence ends, log p(l|x) and the loss are ~20 us behind it.
//   backward (lstm_bwd_flow2): the weight-gradient workers idle until the recurrence has finished its first chunk of frames
//     (~0.5 ms).  Their teams first run ctc_leader, one utterance each: beta backwards in time, 16 frames per round, the
//     posterior from alpha (memory) and beta (registers -- it is never stored), dlogits = softmax - posterior, and
//     dZ_top = dlogits . W_o^T on the matrix cores, written through; the top layer's recurrence groups poll dZ_top like the
//     other layers poll the gradient from the layer above (sentinel).  A round costs ~10 us, the recurrence needs 16 x 5.5:
//     after its first 16 frames the leader is never waited for.
// The recursions, the log-softmax and the posterior use the device functions of ctc_core.h: from the same logits the loss is
// bit-identical to amdspeech_ctc_loss_fwd_bwd's.  dlogits differs in the order its LDS atomics meet (1e-7).
#pragma once
#include "ctc_core.h"

namespace amdspeech {

// W_o [H][C] -> B fragments of v_mfma_f32_16x16x4_f32: block (kb, j) lane (kq, n) holds W_o[kb*16 + 4 kq + m][j*16 + n], m = 0..3
__global__ __launch_bounds__(256) void ctc_pack_wo_kernel(const float* __restrict__ wo, float* __restrict__ pack, int H, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;      // one float4 of the pack
    if (i >= (H / 16) * CF_NTC * 64) return;
    const int lane = i & 63, j = (i >> 6) % CF_NTC, kb = (i >> 6) / CF_NTC;
    const int n = j * 16 + (lane & 15), k = kb * 16 + 4 * (lane >> 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < C) { v.x = wo[(size_t)k * C + n]; v.y = wo[(size_t)(k + 1) * C + n]; v.z = wo[(size_t)(k + 2) * C + n]; v.w = wo[(size_t)(k + 3) * C + n]; }
    reinterpret_cast<float4*>(pack)[i] = v;
}

// The per-utterance state of the recursion of ctc_alpha_beta3_kernel (see there): four waves, 96 owned + 32 copied states each,
// two states per lane (r0: a blank of the extended target, r0 + 1: a label), DPP neighbour shift, refresh every <= 16 frames.
struct CtcChain {
    int b, S, Tb, r0, lab1;
    bool real, ok, act0, act1, st0, st1, skip1;
    double cur0, cur1;
};
template <int DIR>      // 0: alpha, 1: beta (reversed state coordinates)
__device__ __forceinline__ void ctc_chain_init(CtcChain& u, const CtcFlow& c, int b0, int w, int lane) {
    const int T = c.T, B = c.B;
    const int* lengths = c.lengths;
    u.real = b0 < B;
    u.b = u.real ? b0 : B - 1;
    u.ok = c.valid[u.b] != 0;
    u.S = c.slen[u.b];
    u.Tb = min(lengths[u.b], T);
    const int blank = c.C - 1;
    const int* e = c.ext + (size_t)u.b * c.smax;
    u.r0 = w * 96 - 32 + 2 * lane;
    u.act0 = u.r0 >= 0 && u.r0 < u.S;
    u.act1 = u.r0 + 1 >= 0 && u.r0 + 1 < u.S;
    const int sidx1 = DIR == 0 ? u.r0 + 1 : u.S - 2 - u.r0;
    u.lab1 = u.act1 ? e[sidx1] : blank;
    u.skip1 = u.act1 && u.r0 + 1 >= 2 && u.lab1 != blank && u.lab1 != e[DIR == 0 ? u.r0 - 1 : u.S - u.r0];
    u.st0 = u.act0 && lane >= 16;
    u.st1 = u.act1 && lane >= 16;
    u.cur0 = u.cur1 = -__builtin_inf();
}
// the previous wave's highest 32 states -> this wave's lanes 0..15 (one workgroup barrier inside: every wave of the workgroup calls it)
__device__ __forceinline__ void ctc_chain_refresh(CtcChain& u, double2 (*edge)[4][16], int& par, int w, int lane) {
    if (lane >= 48) edge[par][w][lane - 48] = make_double2(u.cur0, u.cur1);
    __syncthreads();
    if (lane < 16 && w > 0) { const double2 v = edge[par][w - 1][lane]; u.cur0 = v.x; u.cur1 = v.y; }
    par ^= 1;
}

// ------------------------------------------------------------------------------------------------ forward: the follower
// wg / nwg: this workgroup's index among the follower workgroups of the launch.  UPT utterances per team, interleaved frame chunk
// by frame chunk (what is left to do when the recurrence ends is one chunk per utterance).  Every workgroup barrier is executed by
// both teams the same number of times: the chunk count is the launch's, never an utterance's.
template <int H, int UPT>
__device__ __forceinline__ void ctc_follower(const CtcFlow& c, float* lds, const int wg, const int nwg) {
    const unsigned long long t_begin = wall_clock64();
    const int T = c.T, B = c.B;
    unsigned* err = c.err;
    const unsigned long long limit = c.limit;
    const int team = threadIdx.x >> 8, tid = threadIdx.x & 255, w = tid >> 6, lane = tid & 63;
    const int nteams = nwg * 2, tm = wg * 2 + team;
    float* part = lds + (size_t)team * CF_FOLLOW_TEAM_FLOATS;                    // [4 waves][16 frames][CF_RP]
    float* lp = part + 4 * 16 * CF_RP;                                           // [16 frames][CF_RP]
    double2 (*edge)[4][16] = reinterpret_cast<double2 (*)[4][16]>(lp + 16 * CF_RP);
    double* fin = reinterpret_cast<double*>(lp + 16 * CF_RP + 2 * 4 * 16 * 4);
    const int C = c.C, blank = C - 1, nmt = (B + 15) / 16;
    const size_t bph = (size_t)nmt * 16 * H;
    const auto r_z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.ztp), 0, (unsigned)((size_t)T * bph * 4), 0x00020000);
    const auto r_al = __builtin_amdgcn_make_buffer_rsrc(c.alpha, 0, (unsigned)((size_t)B * T * c.smax * 4), 0x00020000);
    CtcChain us[UPT];
    unsigned so0[UPT], so1[UPT];
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
        ctc_chain_init<0>(us[u], c, tm + u * nteams, w, lane);
        const unsigned base = (unsigned)((size_t)us[u].b * T * c.smax * 4);
        so0[u] = (us[u].st0 && us[u].real && us[u].ok) ? base + (unsigned)us[u].r0 * 4u : 0x80000000u;
        so1[u] = (us[u].st1 && us[u].real && us[u].ok) ? base + (unsigned)(us[u].r0 + 1) * 4u : 0x80000000u;
    }
    bool dead = false;
    int par = 0;
    const int nch = (T + 15) / 16;
    for (int ch = 0; ch < nch; ++ch) {
        const int t0 = ch * 16;
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            CtcChain& U = us[u];
            // ---- 16 frames x K slice [128 w, 128 w + 128) of the top layer's output, row b of its batch tile: lane (j, kq) takes the
            // float4 (4 k) of frame t0 + j from that frame's panel (packed_off: block (b/16, k/16), float4 kq*16 + b%16)
            const int fj = min(t0 + (lane & 15), T - 1);
            const unsigned vo = (unsigned)((size_t)fj * bph * 4) +
                                (unsigned)(((((size_t)(U.b >> 4) * (H / 16) + (H / 64) * w) * 64) + (lane >> 4) * 16 + (U.b & 15)) * 16);
            u32x4_f av[H / 64];
            if (!dead) {
                // the probe: the chunk's LAST frame (one float4 per K block of this wave); the full load only behind it
                const unsigned vp = (unsigned)((size_t)min(t0 + 15, T - 1) * bph * 4) +
                                    (unsigned)(((((size_t)(U.b >> 4) * (H / 16) + (H / 64) * w + ((lane & 3) & (H / 64 - 1))) * 64) + (lane >> 4) * 16 + (U.b & 15)) * 16);
                while (true) {
                    const u32x4_f pr = __builtin_amdgcn_raw_buffer_load_b128(r_z, vp, 0, 16);
                    if (!__any(flow_pending(pr))) break;
                    if (wall_clock64() - t_begin > limit) { dead = true; if (lane == 0) atomicOr(err, 32u); break; }
                    __builtin_amdgcn_s_sleep(48);
                }
            }
            while (true) {
#pragma unroll
                for (int kb = 0; kb < H / 64; ++kb) av[kb] = __builtin_amdgcn_raw_buffer_load_b128(r_z, vo + (unsigned)(kb * 1024), 0, 16);
                bool pend = false;
#pragma unroll
                for (int kb = 0; kb < H / 64; ++kb) pend = pend || flow_pending(av[kb]);
                if (!__any(pend) || dead) break;
                if (wall_clock64() - t_begin > limit) { dead = true; if (lane == 0) atomicOr(err, 32u); break; }
                __builtin_amdgcn_s_sleep(8);
            }
            // ---- logits partials: [16 frames] x [80 labels] over this wave's K slice
            f32x4 acc[CF_NTC];
#pragma unroll
            for (int j = 0; j < CF_NTC; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float4* wp = reinterpret_cast<const float4*>(c.wo_pack) + (size_t)((H / 64) * w) * CF_NTC * 64 + lane;
#pragma unroll
            for (int kb = 0; kb < H / 64; ++kb) {
                float4 bw[CF_NTC];
#pragma unroll
                for (int j = 0; j < CF_NTC; ++j) bw[j] = wp[(kb * CF_NTC + j) * 64];
#pragma unroll
                for (int j = 0; j < CF_NTC; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av[kb][0]), bw[j].x, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av[kb][1]), bw[j].y, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av[kb][2]), bw[j].z, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av[kb][3]), bw[j].w, acc[j], 0, 0, 0);
                }
            }
            {
                float* pw = part + (size_t)w * 16 * CF_RP;
#pragma unroll
                for (int j = 0; j < CF_NTC; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) pw[(4 * (lane >> 4) + i) * CF_RP + j * 16 + (lane & 15)] = acc[j][i];
            }
            __syncthreads();
            // ---- rows 4 w .. 4 w + 3: sum of the four K slices + bias -> logits; log-softmax -> lp (LDS) and memory
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = 4 * w + rr, t = t0 + r;
                float* row = lp + r * CF_RP;
                for (int cc = lane; cc < C; cc += 64) {
                    const float* p0 = part + r * CF_RP + cc;
                    row[cc] = ((p0[0] + p0[16 * CF_RP]) + p0[2 * 16 * CF_RP]) + p0[3 * 16 * CF_RP] + c.bo[cc];
                }
                __builtin_amdgcn_wave_barrier();
                const bool out = t < T && U.real;
                float* lg = c.logits + ((size_t)t * B + U.b) * C;
                if (out) for (int cc = lane; cc < C; cc += 64) lg[cc] = row[cc];
                const float lse = ctc_row_lse(row, C, lane);
                float* lq = c.logp + ((size_t)t * B + U.b) * C;
                for (int cc = lane; cc < C; cc += 64) {
                    const float y = row[cc] - lse;
                    row[cc] = y;
                    if (out) lq[cc] = y;
                }
            }
            __syncthreads();
            // ---- alpha over these 16 frames (ctc_alpha_beta3_kernel's step, emissions from LDS)
            if (ch > 0) ctc_chain_refresh(U, edge, par, w, lane);
            // (a ROLLED loop, like every frame loop of the two roles: unrolled sixteen times their scalar temporaries drove the
            //  function's SGPR allocation into spilling the recurrence loops' own values -- see lstm_fwd_flow2's CF parameter)
            const int qn = U.ok ? min(16, U.Tb - t0) : 0;
#pragma unroll 1
            for (int q = 0; q < qn; ++q) {
                const int i = t0 + q;
                const float em0 = lp[q * CF_RP + blank], em1 = lp[q * CF_RP + U.lab1];
                const unsigned rowo = (unsigned)i * (unsigned)c.smax * 4u;
                double n0, n1;
                float s0, s1;
                if (i == 0) {
                    n0 = (U.act0 && U.r0 < 2) ? (double)(em0 * LOG2E) : -__builtin_inf();
                    n1 = (U.act1 && U.r0 + 1 < 2) ? (double)(em1 * LOG2E) : -__builtin_inf();
                    s0 = (float)n0 * LN2; s1 = (float)n1 * LN2;
                } else {
                    const double below1 = ctc_from_lane_below(U.cur1);
                    const double v0 = lse2_2d(U.cur0, below1);
                    const double v1 = lse3_2d(U.cur1, U.cur0, U.skip1 ? below1 : -__builtin_inf());
                    n0 = U.act0 ? v0 + (double)(em0 * LOG2E) : -__builtin_inf();
                    n1 = U.act1 ? v1 + (double)(em1 * LOG2E) : -__builtin_inf();
                    s0 = (float)(n0 * (double)LN2); s1 = (float)(n1 * (double)LN2);
                }
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s0), r_al, so0[u] + (so0[u] < 0x80000000u ? rowo : 0u), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s1), r_al, so1[u] + (so1[u] < 0x80000000u ? rowo : 0u), 0, 0);
                U.cur0 = n0; U.cur1 = n1;
            }
        }
    }
    // ---- log p(l|x) = lse(alpha_{Tb-1}(S-1), alpha_{Tb-1}(S-2)); loss
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
        const CtcChain& U = us[u];
        double mine = -__builtin_inf();
        if (U.st0 && (U.r0 == U.S - 1 || U.r0 == U.S - 2)) mine = lse2_2d(mine, U.cur0);
        if (U.st1 && (U.r0 + 1 == U.S - 1 || U.r0 + 1 == U.S - 2)) mine = lse2_2d(mine, U.cur1);
        __syncthreads();
        fin[tid] = mine;
        __syncthreads();
        if (tid < 64) {
            double tot = -__builtin_inf();
#pragma unroll
            for (int k = 0; k < 4; ++k) tot = lse2_2d(tot, fin[k * 64 + tid]);
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) tot = lse2_2d(tot, __shfl_xor(tot, o2));
            if (tid == 0 && U.real) {
                const float llv = U.ok ? (float)(tot * (double)LN2) : 0.f;
                c.ll[U.b] = llv;
                c.loss[U.b] = U.ok ? -llv : 0.f;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: the leader
// team / nteams: this team's index among the worker teams of the launch (two per workgroup; both run the same number of rounds
// and barriers).  One utterance per team and round, frames 16 at a time from the end of the sequence.
template <int H>
__device__ __forceinline__ void ctc_leader(const CtcFlow& c, float* lds, const int worker, const int nworkers) {
    const int T = c.T, B = c.B;
    const int* lengths = c.lengths;
    const int team = threadIdx.x >> 8, tid = threadIdx.x & 255, w = tid >> 6, lane = tid & 63;
    const int nteams = nworkers * 2, tm = worker * 2 + team;
    float* lp = lds + (size_t)team * CF_LEAD_TEAM_FLOATS;                         // [16][CF_RP] log p rows of the chunk
    float* occ = lp + 16 * CF_RP;                                                // [16][CF_RP] posterior per label
    float* dl = occ + 16 * CF_RP;                                                // [16][CF_RP] dlogits rows: the GEMM's operand
    double2 (*edge)[4][16] = reinterpret_cast<double2 (*)[4][16]>(dl + 16 * CF_RP);
    float* al = dl + 16 * CF_RP + 2 * 4 * 16 * 4;                                // [16][CF_AP] alpha rows of the chunk
    const int C = c.C, blank = C - 1;
    const int nch = (T + 15) / 16;
    constexpr int NTW = H / 64;                                                  // 16-unit tiles of dZ_top per wave
    const auto r_top = __builtin_amdgcn_make_buffer_rsrc(c.dztop, 0, (unsigned)((size_t)T * B * H * 4), 0x00020000);
    for (int b0 = worker * 2; b0 < B; b0 += nteams) {        // (workgroup-uniform: both teams or neither)
        CtcChain U;
        ctc_chain_init<1>(U, c, b0 + team, w, lane);
        const float llb = c.ll[U.b];
        const bool nopath = llb == NEG_INF;
        // this lane's two owned states in alpha's (forward) coordinates
        const int sx0 = U.st0 ? U.S - 1 - U.r0 : 0, sx1 = U.st1 ? U.S - 2 - U.r0 : 0;
        int par = 0;
        bool first = true;
        for (int ch = nch - 1; ch >= 0; --ch) {
            const int t0 = ch * 16;
            // ---- the chunk's log p rows and alpha rows -> LDS, occupancy cleared
            for (int idx = tid; idx < 16 * C; idx += 256) {
                const int q = idx / C, cc = idx - q * C, t = min(t0 + q, T - 1);
                lp[q * CF_RP + cc] = c.logp[((size_t)t * B + U.b) * C + cc];
                occ[q * CF_RP + cc] = 0.f;
            }
            {
                const int qn = max(0, min(16, U.Tb - t0));
                const float* arow = c.alpha + ((size_t)U.b * T + t0) * c.smax;
                for (int idx = tid; idx < qn * c.smax; idx += 256) {
                    const int q = idx / c.smax, sx = idx - q * c.smax;
                    al[q * CF_AP + sx] = arow[idx];
                }
            }
            __syncthreads();
            if (!first) ctc_chain_refresh(U, edge, par, w, lane); else __syncthreads();
            // ---- beta over the chunk's frames, last first (ctc_alpha_beta3_kernel's step in reversed coordinates).  What the staged
            // call would have stored as beta_t(s) stays in registers, rounded the same way, and meets alpha_t(s) at once: the
            // posterior exp(alpha + beta - log p(l|x)) of the owned states (ctc_grad_kernel's arithmetic) -- a wave's blanks are
            // summed in registers, the labels meet in LDS atomics.  (Rolled: see ctc_follower.)
            const int qhi = U.ok ? min(15, U.Tb - 1 - t0) : -1;
#pragma unroll 1
            for (int q = qhi; q >= 0; --q) {
                const int f = t0 + q;
                const float em0 = lp[q * CF_RP + blank], em1 = lp[q * CF_RP + U.lab1];
                float be0, be1;
                if (f == U.Tb - 1) {
                    U.cur0 = (U.act0 && U.r0 < 2) ? (double)(em0 * LOG2E) : -__builtin_inf();
                    U.cur1 = (U.act1 && U.r0 + 1 < 2) ? (double)(em1 * LOG2E) : -__builtin_inf();
                    be0 = U.r0 < 2 ? 0.f : NEG_INF;
                    be1 = U.r0 + 1 < 2 ? 0.f : NEG_INF;
                    first = false;
                } else {
                    const double below1 = ctc_from_lane_below(U.cur1);
                    const double v0 = lse2_2d(U.cur0, below1);
                    const double v1 = lse3_2d(U.cur1, U.cur0, U.skip1 ? below1 : -__builtin_inf());
                    U.cur0 = U.act0 ? v0 + (double)(em0 * LOG2E) : -__builtin_inf();
                    U.cur1 = U.act1 ? v1 + (double)(em1 * LOG2E) : -__builtin_inf();
                    be0 = (float)(v0 * (double)LN2);
                    be1 = (float)(v1 * (double)LN2);
                }
                if (!nopath) {
                    const float a0 = U.st0 ? al[q * CF_AP + sx0] : NEG_INF, a1 = U.st1 ? al[q * CF_AP + sx1] : NEG_INF;
                    float p0 = U.st0 ? expf((float)((double)a0 + (double)be0 - (double)llb)) : 0.f;
                    const float p1 = U.st1 ? expf((float)((double)a1 + (double)be1 - (double)llb)) : 0.f;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) p0 += __shfl_xor(p0, o);
                    if (lane == 0 && p0 > 0.f) atomicAdd(&occ[q * CF_RP + blank], p0);
                    if (p1 > 0.f) atomicAdd(&occ[q * CF_RP + U.lab1], p1);
                }
            }
            __syncthreads();
            // ---- dlogits rows (ctc_grad_kernel's cases) -> memory and LDS
            for (int idx = tid; idx < 16 * C; idx += 256) {
                const int q = idx / C, cc = idx - q * C, t = t0 + q;
                float g = 0.f;
                if (U.ok && t < lengths[U.b] && t < T) g = nopath ? expf(lp[q * CF_RP + cc]) : expf(lp[q * CF_RP + cc]) - occ[q * CF_RP + cc];
                dl[q * CF_RP + cc] = g;
                if (t < T && U.real) c.dlogits[((size_t)t * B + U.b) * C + cc] = g;
            }
            __syncthreads();
            // ---- dZ_top[t0 + j][b][units] = dlogits[16 x C] . W_o^T, operand-swapped (weights on the A port): a lane's accumulator is
            // four consecutive units of one frame, one 16-byte store
            {
                f32x4 acc[NTW];
#pragma unroll
                for (int n = 0; n < NTW; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const int j = lane & 15, kq = lane >> 4;
                for (int kb = 0; kb * 16 < C; ++kb) {
                    const float4 bv = *reinterpret_cast<const float4*>(dl + j * CF_RP + kb * 16 + 4 * kq);
                    float4 aw[NTW];
#pragma unroll
                    for (int n = 0; n < NTW; ++n)
                        aw[n] = *reinterpret_cast<const float4*>(c.wo + (size_t)((w * NTW + n) * 16 + j) * C + kb * 16 + 4 * kq);
#pragma unroll
                    for (int n = 0; n < NTW; ++n) {
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[n].x, bv.x, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[n].y, bv.y, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[n].z, bv.z, acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[n].w, bv.w, acc[n], 0, 0, 0);
                    }
                }
                const int t = t0 + j;
                const unsigned base = (t < T && U.real) ? (unsigned)((((size_t)t * B + U.b) * H + 4 * kq) * 4) : 0x80000000u;
#pragma unroll
                for (int n = 0; n < NTW; ++n) {
                    const u32x4_f v = {__float_as_uint(acc[n][0]), __float_as_uint(acc[n][1]), __float_as_uint(acc[n][2]), __float_as_uint(acc[n][3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, r_top, base + (base < 0x80000000u ? (unsigned)((w * NTW + n) * 64) : 0u), 0, 16);      // sc1
                }
            }
        }
    }
}

}  // namespace amdspeech
