// Device helpers shared by the CTC kernels (ctc.hip) and the CTC head fused into the dataflow LSTM kernels (ctc_flow.h):
// the log2-domain log-sum-exps, the DPP neighbour shift of ctc_alpha_beta3_kernel, the row log-softmax.  One definition, so
// that the fused head and the staged call produce the same bits from the same logits.
#pragma once
#include "common.h"

namespace amdspeech {

#define NEG_INF (-__builtin_inff())

struct CtcLayout { size_t logp, alpha, beta, ext, slen, valid, ll, total; int smax; };  // byte offsets

static CtcLayout ctc_layout(int T, int B, int C, int U) {
    CtcLayout o;
    o.smax = 2 * U + 1;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t r = off; off += align_up(bytes, 256); return r; };
    o.logp = take((size_t)T * B * C * 4);
    o.alpha = take((size_t)B * T * o.smax * 4);
    o.beta = take((size_t)B * T * o.smax * 4);
    o.ext = take((size_t)B * o.smax * 4);
    o.slen = take((size_t)B * 4);
    o.valid = take((size_t)B * 4);
    o.ll = take((size_t)B * 4);
    o.total = off;
    return o;
}


// The CTC head fused into the dataflow LSTM kernels (ctc_flow.h): what both launches of a training step are told about it
// EVERYTHING a role reads is in here, and the struct is 64-byte aligned at the end of the kernel's argument struct: hipcc loads a
// kernel's arguments at function entry in 8- / 16-dword pieces, and a piece holding one field a role reads and another the
// recurrence reads is ONE register tuple alive in both regions -- under the role's scalar pressure it is spilled where it is
// defined, and the recurrence then re-reads its fields lane move by lane move inside its steady loop (seen: 20 -> 100 per step).
struct alignas(64) CtcFlow {
    int on;                    // 0: this launch has no fused head
    int C, smax;               // labels (<= 80, multiple of 16), pitch of the extended targets (<= 384)
    int nfw;                   // forward: follower workgroups per spare XCD (behind the x-product workers)
    int T, B;                  // the launch's frames and batch rows
    const int* lengths;        // [B]
    unsigned* err;             // the launch's error word (bit 32: the follower gave up waiting)
    unsigned long long limit;  // wall_clock64 ticks a wait may last
    const float* ztp;          // forward: packed panels [T][bp][H] of the top layer's masked output (sentinel pre-filled)
    const float* wo;           // W_o [H][C]
    const float* wo_pack;      // forward: W_o as MFMA B fragments, [H/16][5][64] float4 (ctc_pack_wo_kernel)
    const float* bo;           // b_o [C]
    float* logits;             // [T][B][C]
    float* logp;               // [T][B][C]   (CTC workspace)
    float* alpha;              // [B][T][smax]
    float* ll;                 // [B]
    float* loss;               // [B]
    float* dlogits;            // backward: [T][B][C]
    float* dztop;              // backward: [T][B][H], sentinel pre-filled, written through
    const int* ext; const int* slen; const int* valid;
};

constexpr int CF_NTC = 5, CF_RP = 84;      // N tiles of the output layer (C <= 80); LDS row pitch in floats (16-byte aligned rows)
constexpr int CF_FOLLOW_TEAM_FLOATS = 4 * 16 * CF_RP + 16 * CF_RP + 2 * 4 * 16 * 4 + 256 * 2;      // partial sums, log p rows, edge, fin
constexpr int CF_AP = 388;                 // pitch of the leader's alpha rows in LDS (>= 384 extended states)
constexpr int CF_LEAD_TEAM_FLOATS = 3 * 16 * CF_RP + 2 * 4 * 16 * 4 + 16 * CF_AP;                    // log p, occupancy, dlogits, edge, alpha rows

#ifndef CTC_DIAG
#define CTC_DIAG 0      // dev ablations of ctc_alpha_beta2_kernel: 1 no alpha/beta stores, 2 no emission gathers, 3 no LDS exchange / barrier, 4 no transcendentals
#endif
__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m == NEG_INF) return NEG_INF;
    return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

// The alpha/beta recursions run in the log2 domain: one v_exp_f32 / v_log_f32 (1 ulp, quarter rate)
// per term instead of the ~20-instruction expf/logf expansions -- the chain of T dependent
// log-sum-exps is the whole cost of this kernel.
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
__device__ __forceinline__ float lse3_2(float a, float b, float c) {
#if defined(CTC_DIAG) && CTC_DIAG == 4
    return fmaxf(a, fmaxf(b, c)) + 0.3f;
#endif
    // branch-free (the frame loop is a chain of these)
    const float mm = fmaxf(fmaxf(a, fmaxf(b, c)), -1e30f);     // (all three at -inf: mm = -1e30, exp2(-inf) = 0, log2(0) = -inf)
    return mm + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - mm) + __builtin_amdgcn_exp2f(b - mm) +
                                      __builtin_amdgcn_exp2f(c - mm));
}

__device__ __forceinline__ float lse2_2(float a, float b) {          // = lse3_2(a, b, -inf), bit for bit (the third term adds 0)
#if defined(CTC_DIAG) && CTC_DIAG == 4
    return fmaxf(a, b) + 0.3f;
#endif
    const float mm = fmaxf(fmaxf(a, b), -1e30f);
    return mm + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - mm) + __builtin_amdgcn_exp2f(b - mm));
}
__device__ __forceinline__ float ctc_from_lane_below(float v) {       // lane i <- lane i - 1; lane 0 <- -inf
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(NEG_INF), __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
// Round 4: the recursion STATE is float64.  After 1001 frames |alpha| ~ 10^3 (log2 units), where a float's ulp is 6e-5: every
// log-sum-exp rounded its result by that much, a thousand times over, and dlogits = softmax - exp(alpha + beta - log p) ended
// up 2-3e-3 of its maximum away from the float64 oracle (TensorFlow's op is float32 too, but the gradient is what is trained
// on).  gfx950 adds and compares doubles at the float rate; the transcendentals stay v_exp_f32 / v_log_f32 on the DIFFERENCES
// to the maximum, which are small numbers -- what a float loses there is 1e-7 of a term, not 6e-5 of the sum.
__device__ __forceinline__ double lse2_2d(double a, double b) {
    const double mm = fmax(fmax(a, b), -1e30);
    return mm + (double)__builtin_amdgcn_logf(__builtin_amdgcn_exp2f((float)(a - mm)) + __builtin_amdgcn_exp2f((float)(b - mm)));
}
__device__ __forceinline__ double lse3_2d(double a, double b, double c) {
    const double mm = fmax(fmax(a, fmax(b, c)), -1e30);
    return mm + (double)__builtin_amdgcn_logf(__builtin_amdgcn_exp2f((float)(a - mm)) + __builtin_amdgcn_exp2f((float)(b - mm)) +
                                              __builtin_amdgcn_exp2f((float)(c - mm)));
}
__device__ __forceinline__ double ctc_from_lane_below(double v) {     // (two 32-bit DPP moves)
    const long long bits = __double_as_longlong(v), ninf = __double_as_longlong(-__builtin_inf());
    const int lo = __builtin_amdgcn_update_dpp((int)ninf, (int)bits, 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(ninf >> 32), (int)(bits >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// log-softmax of one row of C logits by ONE wave (lane c and c + 64, ...): the arithmetic of log_softmax_kernel, shared with the
// fused head.  Returns the row's log-sum-exp; y = x - lse is left to the caller.
__device__ __forceinline__ float ctc_row_lse(const float* xr, int C, int lane) {
    float m = NEG_INF;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, xr[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += expf(xr[c] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    return m + logf(sum);
}

}  // namespace amdspeech
