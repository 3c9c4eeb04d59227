// Stacked-LSTM forward and BPTT for gfx950 (replaces BasicLSTMCell + DropoutWrapper +
// MultiRNNCell + dynamic_rnn, /root/reference/models/AcousticModel.py:223-237,266-298).
//
// Design (MI355X-first):
//  * The recurrence is latency bound: per frame and layer the dependent product is
//    only [B, 2H] x [2H, 4H].  All L layers advance together along the anti-diagonal
//    d = t + l (wavefront pipelining), so the dependent chain is T+L-1 short kernels,
//    not T*L; each launch is cut at the h all-gather seam (a kernel boundary costs
//    ~1.5 us on this chip, less than any in-kernel grid barrier).
//  * A workgroup owns a slice of hidden units for ALL four gates, so the gate
//    non-linearities, the cell update, length masking and dropout are fused behind
//    the MFMAs and nothing but h/c/gates ever goes back to HBM.
//  * [x_t ; h_{t-1}] . K uses v_mfma_f32_16x16x4_f32 (exact f32).  The 2H-long K axis
//    is split across the 4 waves of a workgroup (one per SIMD), reduced through LDS.
//  * Weights are repacked once per optimiser step into MFMA B-fragment order: one
//    fully coalesced 1 KiB float4 load per wave feeds four MFMAs; the slices stay
//    L2/MALL resident across the T launches (24 MB total for 3x512).
//  * BPTT runs the mirrored diagonal: dh_t = dG_{t+1} . W_hh^T (+ dG^{l+1}_t . W_ih^T
//    from the layer above) fused with the gate-gradient math; the weight gradients
//    dK = [Z ; Hprev]^T . dG are time-independent and go to the big split-K GEMM.
#include "common.h"
#include <stdlib.h>

namespace amdspeech {

// ------------------------------------------------------------------ workspace
struct LstmLayout {
    size_t wp, wq, z, hs, cs, gates, dg, dztop, dz0, dc, total;  // float offsets
};

static LstmLayout lstm_layout(const amdspeech_lstm_desc* d) {
    const size_t T = d->T, B = d->B, H = d->H, L = d->L;
    const size_t tbh = T * B * H;
    LstmLayout o;
    size_t off = 0;
    auto take = [&](size_t n) { size_t r = off; off += (n + 63) / 64 * 64; return r; };
    o.wp = take(L * 2 * H * 4 * H);
    o.wq = take(L * 2 * H * 4 * H);
    o.z = take((L + 1) * tbh);
    o.hs = take(L * (T + 1) * B * H);
    o.cs = take(L * (T + 1) * B * H);
    o.gates = take(L * tbh * 4);
    o.dg = take(L * tbh * 4);
    o.dztop = take(tbh);
    o.dz0 = take(tbh);
    o.dc = take(L * 2 * B * H);
    o.total = off;
    return o;
}

static int check_desc(const amdspeech_lstm_desc* d) {
    AS_CHECK_ARG(d != nullptr, "lstm: null descriptor");
    AS_CHECK_ARG(d->T > 0 && d->B > 0 && d->H > 0 && d->L > 0, "lstm: bad shape T=%d B=%d H=%d L=%d",
                 d->T, d->B, d->H, d->L);
    AS_CHECK_ARG(d->H % 16 == 0, "lstm: hidden size %d must be a multiple of 16", d->H);
    AS_CHECK_ARG(d->keep_in > 0.f && d->keep_in <= 1.f && d->keep_out > 0.f && d->keep_out <= 1.f,
                 "lstm: keep probabilities must be in (0,1]");
    AS_CHECK_ARG((size_t)d->T * d->B * d->H < (1ull << 32), "lstm: T*B*H too large for the dropout counter");
    return AMDSPEECH_OK;
}

// ------------------------------------------------------------------- dropout
struct DropCfg { float keep_in, keep_out; uint64_t seed; int L; };

// Multiplier of inter-layer tensor Z_lp (lp = 0..L): input mask of layer lp (if it
// exists) times output mask of layer lp-1 (if it exists), each mask/keep.
__device__ __forceinline__ float zmult(const DropCfg& c, int lp, uint32_t idx) {
    float m = 1.0f;
    if (c.keep_in < 1.0f && lp < c.L)
        m *= (uniform01(c.seed, 2u * lp, idx) < c.keep_in) ? (1.0f / c.keep_in) : 0.0f;
    if (c.keep_out < 1.0f && lp >= 1)
        m *= (uniform01(c.seed, 2u * (lp - 1) + 1u, idx) < c.keep_out) ? (1.0f / c.keep_out) : 0.0f;
    return m;
}

__global__ void apply_zmult_kernel(float* x, long n, DropCfg c, int lp) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= zmult(c, lp, (uint32_t)i);
}

// ------------------------------------------------------------ weight packing
// Forward B-fragments.  Workgroup ub owns UW units x 4 gates = 4*UW columns,
// local column c = g*UW + u, N-tile nt = c/16, j = c%16.  For K-block kb (16 rows
// of K) lane (j, kq) holds rows kb*16 + 4*kq + m, m = 0..3, as one float4:
//   Wp[(((l*NUB + ub)*NKB + kb)*NT + nt)*256 + lane*4 + m]
__global__ void pack_fwd_kernel(const float* __restrict__ kernels, long kstride, float* __restrict__ wp,
                                int H, int L, int UW) {
    const int NT = UW / 4, NKB = 2 * H / 16, NUB = H / UW;
    const long total = (long)L * 2 * H * 4 * H;
    long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    int m = o & 3, lane = (o >> 2) & 63;
    long r = o >> 8;
    int nt = r % NT; r /= NT;
    int kb = r % NKB; r /= NKB;
    int ub = r % NUB; int l = r / NUB;
    int j = lane & 15, kq = lane >> 4;
    int c = nt * 16 + j, g = c / UW, u = c % UW;
    int k = kb * 16 + 4 * kq + m;
    wp[o] = kernels[l * kstride + (long)k * 4 * H + g * H + ub * UW + u];
}

// Backward B-fragments = K^T: row block rb (16 rows of K = 16 input units), K-block
// kb (16 gate columns):  Wq[((l*(2H/16) + rb)*(4H/16) + kb)*256 + lane*4 + m]
//   = K_l[rb*16 + (lane&15)][kb*16 + 4*(lane>>4) + m]
__global__ void pack_bwd_kernel(const float* __restrict__ kernels, long kstride, float* __restrict__ wq,
                                int H, int L) {
    const int NRB = 2 * H / 16, NKB = 4 * H / 16;
    const long total = (long)L * 2 * H * 4 * H;
    long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    int m = o & 3, lane = (o >> 2) & 63;
    long r = o >> 8;
    int kb = r % NKB; r /= NKB;
    int rb = r % NRB; int l = r / NRB;
    int row = rb * 16 + (lane & 15), col = kb * 16 + 4 * (lane >> 4) + m;
    wq[o] = kernels[l * kstride + (long)row * 4 * H + col];
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------- forward step
struct FwdArgs {
    const float* wp; const float* bias; long bias_stride;
    float* z; float* hs; float* cs; float* gates; const int* lengths;
    int T, B, H, L, d;
    DropCfg drop;
    int dbg;   // dev-only timing experiments (AMDSPEECH_DBG): 1 = A from one hot line, 2 = B from one hot line
    unsigned long long* trace; int trace_d;   // dev-only: per-wave s_memtime stamps for diagonal trace_d
};

template <int UW, int NW, int UN, bool DB>   // units/workgroup, waves/workgroup, K-blocks per load burst, double buffer
__global__ __launch_bounds__(NW * 64) void lstm_fwd_step(FwdArgs a) {
    constexpr int NT = UW / 4, MT = 2;
    const int l = blockIdx.y;
    const int t = a.d - l;
    if (t < 0 || t >= a.T) return;
    const int ub = blockIdx.x, mb = blockIdx.z;
    const int T = a.T, B = a.B, H = a.H;
    const int nkb = 2 * H / 16, nkb_x = H / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;

    const float* x = a.z + ((size_t)l * T + t) * B * H;            // Z_l[t]
    const float* hp = a.hs + ((size_t)l * (T + 1) + t) * B * H;    // h_{t-1}
    const float* wp = a.wp + ((size_t)(l * (H / UW) + ub) * nkb) * (NT * 256) + lane * 4;
    const bool tracing = a.trace != nullptr && a.d == a.trace_d;
    unsigned long long* tr = a.trace + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW + wave) * 8;
#define STAMP(i) do { if (tracing && lane == 0) { tr[i] = __builtin_amdgcn_s_memtime(); if (i == 0) tr[7] = wall_clock64(); if (i == 3) tr[6] = wall_clock64(); } } while (0)
    STAMP(0);

    // ---- epilogue operands: issue their loads first so they land under the MFMA phase
    const float* bias = a.bias + l * a.bias_stride;
    const float* cprev = a.cs + ((size_t)l * (T + 1) + t) * B * H;
    const int pidx = threadIdx.x % (32 * UW);     // (batch row, unit) pair of this thread
    const int pbl = pidx / UW, pu = pidx % UW;
    const int pb = mb * 32 + pbl, punit = ub * UW + pu;
    const bool pok = threadIdx.x < 32 * UW && pb < B;
    const int pbc = min(pb, B - 1);               // clamped: unconditional loads, no branches
    float e_bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) e_bias[g] = bias[g * H + punit];
    const float e_cp = cprev[(size_t)pbc * H + punit];
    const float e_hp = hp[(size_t)pbc * H + punit];
    const int e_len = a.lengths[pbc];

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    size_t rowoff[MT]; bool rok[MT]; (void)rok;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        // rows past B are clamped (loads stay unconditional: a predicated load makes hipcc
        // branch + wait per load); their results are never stored
        const int r = min(mb * 32 + i * 16 + li, B - 1);
        rok[i] = true;
        rowoff[i] = (size_t)r * H + 4 * kq;
    }
    const int kb0 = wave * nkb / NW, kb1 = (wave + 1) * nkb / NW;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    auto load_batch = [&](int kbs, float4 (&av)[UN][MT], float4 (&bv)[UN][NT]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool kok = kbs + u < kb1;
            const int kb = min(kbs + u, kb1 - 1);      // clamped address, data zeroed by select
            const int kba = (a.dbg & 1) ? kb0 : kb, kbb = (a.dbg & 2) ? kb0 : kb;
            const bool isx = kba < nkb_x;
            const float* src = (isx ? x : hp) + (isx ? kba : kba - nkb_x) * 16;
#pragma unroll
            for (int i = 0; i < MT; ++i) av[u][i] = *reinterpret_cast<const float4*>(src + rowoff[i]);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float4 w = *reinterpret_cast<const float4*>(wp + (size_t)(kbb * NT + j) * 256);
                bv[u][j] = kok ? w : zero4;
            }
        }
    };
    auto mma_batch = [&](const float4 (&av)[UN][MT], const float4 (&bv)[UN][NT]) {
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i].x, bv[u][j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i].y, bv[u][j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i].z, bv[u][j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i].w, bv[u][j].w, acc[i][j], 0, 0, 0);
                }
    };
    if (!DB) {
        // one register set: a burst of UN*(MT+NT) loads, then its MFMAs; other waves of the
        // CU cover the latency (thread-level parallelism)
        float4 a0[UN][MT], b0[UN][NT];
        for (int kb = kb0; kb < kb1; kb += UN) {
            load_batch(kb, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma_batch(a0, b0);
        }
    } else {
        // software pipeline, two register sets; the steady-state body has no branches so
        // hipcc keeps the next batch's loads in flight under this batch's MFMAs
        float4 a0[UN][MT], b0[UN][NT], a1[UN][MT], b1[UN][NT];
        const int nb = (kb1 - kb0 + UN - 1) / UN;
        int i = 0;
        // sched_barrier: keep each burst of loads together and ahead of the MFMAs (memory-level
        // parallelism is what bounds this kernel: every operand comes from MALL/HBM, ~1 us away)
        load_batch(kb0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        for (; i + 2 < nb; i += 2) {
            load_batch(kb0 + (i + 1) * UN, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mma_batch(a0, b0);
            load_batch(kb0 + (i + 2) * UN, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma_batch(a1, b1);
        }
        if (nb - i == 2) {
            load_batch(kb0 + (i + 1) * UN, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mma_batch(a0, b0);
            mma_batch(a1, b1);
        } else if (nb - i == 1) {
            mma_batch(a0, b0);
        }
    }

    __shared__ __attribute__((aligned(16))) float red[NW][MT * NT][256];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            *reinterpret_cast<f32x4*>(&red[wave][i * NT + j][lane * 4]) = acc[i][j];
    STAMP(1);
    __syncthreads();
    STAMP(2);

    if (!pok) return;
    float* gates = a.gates + ((size_t)l * T + t) * B * 4 * H;
    float* cnext = a.cs + ((size_t)l * (T + 1) + t + 1) * B * H;
    float* hnext = a.hs + ((size_t)l * (T + 1) + t + 1) * B * H;
    float* zout = a.z + ((size_t)(l + 1) * T + t) * B * H;
    const int mt = pbl >> 4, i = pbl & 15;
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = g * UW + pu, nt = c >> 4, j = c & 15;
        const int e = ((i >> 2) * 16 + j) * 4 + (i & 3);
        const int tl = mt * NT + nt;
        float sacc = e_bias[g];
#pragma unroll
        for (int w = 0; w < NW; ++w) sacc += red[w][tl][e];
        pre[g] = sacc;
    }
    const float gi = sigmoidf_(pre[0]);
    const float gj = tanhf(pre[1]);
    const float gf = sigmoidf_(pre[2] + 1.0f);   // forget_bias = 1.0, added at run time
    const float go = sigmoidf_(pre[3]);
    const size_t e = (size_t)pb * H + punit;
    const float cn = e_cp * gf + gi * gj;
    const float hn = tanhf(cn) * go;
    const bool live = t < e_len;
    float* gr = gates + (size_t)pb * 4 * H + punit;
    gr[0] = gi; gr[H] = gj; gr[2 * H] = gf; gr[3 * H] = go;
    cnext[e] = live ? cn : e_cp;
    hnext[e] = live ? hn : e_hp;
    zout[e] = live ? hn * zmult(a.drop, l + 1, (uint32_t)((size_t)t * B * H + e)) : 0.0f;
    STAMP(3);
#undef STAMP
}

// ------------------------------------------------------------ backward step
struct BwdArgs {
    const float* wq; const float* cs; const float* gates; float* dg; const float* dztop; float* dc;
    const int* lengths;
    int T, B, H, L, d;
    DropCfg drop;
};

template <int NW, int UN, bool DB>    // waves per workgroup, virtual K-blocks per load burst, double buffer
__global__ __launch_bounds__(NW * 64) void lstm_bwd_step(BwdArgs a) {
    const int l = blockIdx.y;
    const int T = a.T, B = a.B, H = a.H, L = a.L;
    const int t = (T - 1) - (a.d - (L - 1 - l));
    if (t < 0 || t >= T) return;
    const int ub = blockIdx.x, mb = blockIdx.z;      // 16 units x 16 batch rows
    const int nkb = 4 * H / 16, nrb = 2 * H / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int row = min(mb * 16 + li, B - 1);    // clamped: loads stay unconditional
    const bool has_rec = t + 1 < T, has_up = l + 1 < L;

    // ---- epilogue operands first: their latency hides under the MFMA phase
    const int bl = (threadIdx.x & 255) >> 4, u = threadIdx.x & 15;
    const int b = mb * 16 + bl;
    const int unit = ub * 16 + u;
    const bool pok = threadIdx.x < 256 && b < B;
    const int bc = min(b, B - 1);                 // clamped: unconditional loads, no branches
    const size_t bec = (size_t)bc * H + unit;
    const size_t be = (size_t)b * H + unit;
    float* dcb = a.dc + (size_t)l * 2 * B * H;
    const float* gr = a.gates + ((size_t)l * T + t) * B * 4 * H + (size_t)bc * 4 * H + unit;
    const float gi = gr[0], gj = gr[H], gf = gr[2 * H], go = gr[3 * H];
    const float c = a.cs[((size_t)l * (T + 1) + t + 1) * B * H + bec];
    const float cp = a.cs[((size_t)l * (T + 1) + t) * B * H + bec];
    const float dcin_raw = dcb[(size_t)((t + 1) & 1) * B * H + bec];   // garbage at t = T-1, selected away
    const float dtop = a.dztop[(size_t)t * B * H + bec];
    const int len = a.lengths[bc];
    const float dcin = has_rec ? dcin_raw : 0.0f;

    // Two product streams share the loop: s=0 "rec" dG_l[t+1].W_hh^T, s=1 "up" dG_{l+1}[t].W_ih^T.
    const float *a_src0, *a_src1, *b_src0, *b_src1;   // (no arrays: a runtime index would go to scratch)
    a_src0 = a.dg + ((size_t)l * T + (t + 1)) * B * 4 * H + (size_t)row * 4 * H + 4 * kq;
    a_src1 = a.dg + ((size_t)(l + 1) * T + t) * B * 4 * H + (size_t)row * 4 * H + 4 * kq;
    b_src0 = a.wq + ((size_t)(l * nrb + H / 16 + ub) * nkb) * 256 + lane * 4;
    b_src1 = a.wq + ((size_t)((l + 1) * nrb + ub) * nkb) * 256 + lane * 4;
    const int nsrc = (has_rec ? 1 : 0) + (has_up ? 1 : 0);
    const int kb0 = wave * nkb / NW, kb1 = (wave + 1) * nkb / NW;
    const int nv = (kb1 - kb0) * nsrc;             // virtual blocks: both -> alternate rec/up
    const int only = has_rec ? 0 : 1;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // two independent MFMA chains
    auto load_batch = [&](int vs, float4 (&av)[UN], float4 (&bv)[UN]) {
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const bool ok = vs + q < nv;
            const int v = min(vs + q, nv - 1);          // clamped address, data zeroed by select
            const int sidx = nsrc == 2 ? (v & 1) : only;
            const int kb = kb0 + (nsrc == 2 ? (v >> 1) : v);
            av[q] = *reinterpret_cast<const float4*>((sidx ? a_src1 : a_src0) + kb * 16);
            const float4 w = *reinterpret_cast<const float4*>((sidx ? b_src1 : b_src0) + (size_t)kb * 256);
            bv[q] = ok ? w : zero4;
        }
    };
    auto mma_batch = [&](const float4 (&av)[UN], const float4 (&bv)[UN]) {
#pragma unroll
        for (int q = 0; q < UN; ++q) {      // vs is a multiple of UN (even) -> parity of v == parity of q
            acc[q & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q].x, bv[q].x, acc[q & 1], 0, 0, 0);
            acc[q & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q].y, bv[q].y, acc[q & 1], 0, 0, 0);
            acc[q & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q].z, bv[q].z, acc[q & 1], 0, 0, 0);
            acc[q & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q].w, bv[q].w, acc[q & 1], 0, 0, 0);
        }
    };
    if (!DB) {
        float4 a0[UN], b0[UN];
        for (int v = 0; v < nv; v += UN) {
            load_batch(v, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma_batch(a0, b0);
        }
    } else if (nv > 0) {
        float4 a0[UN], b0[UN], a1[UN], b1[UN];
        const int nb = (nv + UN - 1) / UN;
        int i = 0;
        load_batch(0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        for (; i + 2 < nb; i += 2) {             // branch-free steady state (see lstm_fwd_step)
            load_batch((i + 1) * UN, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mma_batch(a0, b0);
            load_batch((i + 2) * UN, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma_batch(a1, b1);
        }
        if (nb - i == 2) {
            load_batch((i + 1) * UN, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mma_batch(a0, b0);
            mma_batch(a1, b1);
        } else if (nb - i == 1) {
            mma_batch(a0, b0);
        }
    }
    f32x4 acc_r, acc_u;
    if (nsrc == 2) { acc_r = acc[0]; acc_u = acc[1]; }
    else if (has_rec) { acc_r = acc[0] + acc[1]; acc_u = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    else { acc_u = acc[0] + acc[1]; acc_r = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    __shared__ __attribute__((aligned(16))) float red[NW][2][256];
    *reinterpret_cast<f32x4*>(&red[wave][0][lane * 4]) = acc_r;
    *reinterpret_cast<f32x4*>(&red[wave][1][lane * 4]) = acc_u;
    __syncthreads();

    if (!pok) return;
    const int e = ((bl >> 2) * 16 + u) * 4 + (bl & 3);
    float drec = 0.f, dsum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { drec += red[w][0][e]; dsum += red[w][1][e]; }
    const float dup = has_up ? dsum : dtop;
    const float dh = drec + dup * zmult(a.drop, l + 1, (uint32_t)((size_t)t * B * H + be));
    const bool live = t < len;
    const float tc = tanhf(c);
    const float dct = dcin + dh * go * (1.0f - tc * tc);
    float dgi = dct * gj * gi * (1.0f - gi);
    float dgj = dct * gi * (1.0f - gj * gj);
    float dgf = dct * cp * gf * (1.0f - gf);
    float dgo = dh * tc * go * (1.0f - go);
    float dcout = dct * gf;
    if (!live) { dgi = dgj = dgf = dgo = 0.0f; dcout = 0.0f; }
    float* dgw = a.dg + ((size_t)l * T + t) * B * 4 * H + (size_t)b * 4 * H + unit;
    dgw[0] = dgi; dgw[H] = dgj; dgw[2 * H] = dgf; dgw[3 * H] = dgo;
    dcb[(size_t)(t & 1) * B * H + be] = dcout;
}

// ---------------------------------------------------------------- profiling
static bool g_prof_on = false;
static hipEvent_t g_prof_ev[2][2];
static int g_prof_launches[2] = {0, 0};
static bool g_prof_valid[2] = {false, false};

static void prof_begin(int which, hipStream_t s) {
    if (g_prof_on) (void)hipEventRecord(g_prof_ev[which][0], s);
}
static void prof_end(int which, hipStream_t s, int launches) {
    if (!g_prof_on) return;
    (void)hipEventRecord(g_prof_ev[which][1], s);
    g_prof_launches[which] = launches;
    g_prof_valid[which] = true;
}

// --------------------------------------------------------------- host side
static int pick_uw(const amdspeech_lstm_desc* d) {
    if (getenv("AMDSPEECH_UW")) return atoi(getenv("AMDSPEECH_UW"));
    // 8 units (two 16-column N tiles) per workgroup halves the redundant re-reads of the
    // [B, 2H] activation panel; fall back to 4 when that would leave most CUs without work.
    const long wgs8 = (long)d->L * (d->H / 8) * ceil_div(d->B, 32);
    return (d->H % 8 == 0 && wgs8 >= 96) ? 8 : 4;
}

int lstm_fwd(hipStream_t s, const amdspeech_lstm_desc* d, float* ws, const float* kernels, long kstride,
             const float* biases, long bstride, const int* lengths, const float* h0, const float* c0) {
    if (int rc = check_desc(d)) return rc;
    AS_CHECK_ARG(ws && kernels && biases && lengths, "lstm_fwd: null pointer");
    AS_CHECK_ARG(((uintptr_t)ws % 256) == 0, "lstm_fwd: workspace must be 256-byte aligned");
    const LstmLayout lo = lstm_layout(d);
    const int T = d->T, B = d->B, H = d->H, L = d->L;
    const int uw = pick_uw(d);
    const long wtotal = (long)L * 2 * H * 4 * H;
    hipLaunchKernelGGL(pack_fwd_kernel, dim3(ceil_div(wtotal, 256)), dim3(256), 0, s, kernels, kstride,
                       ws + lo.wp, H, L, uw);
    AS_CHECK_LAUNCH();
    const size_t bh = (size_t)B * H;
    for (int l = 0; l < L; ++l) {
        float* hs0 = ws + lo.hs + (size_t)l * (T + 1) * bh;
        float* cs0 = ws + lo.cs + (size_t)l * (T + 1) * bh;
        if (h0) AS_CHECK_HIP(hipMemcpyAsync(hs0, h0 + l * bh, bh * 4, hipMemcpyDeviceToDevice, s));
        else AS_CHECK_HIP(hipMemsetAsync(hs0, 0, bh * 4, s));
        if (c0) AS_CHECK_HIP(hipMemcpyAsync(cs0, c0 + l * bh, bh * 4, hipMemcpyDeviceToDevice, s));
        else AS_CHECK_HIP(hipMemsetAsync(cs0, 0, bh * 4, s));
    }
    DropCfg dc{d->keep_in, d->keep_out, d->seed, L};
    if (d->keep_in < 1.0f) {
        const long n = (long)T * bh;
        hipLaunchKernelGGL(apply_zmult_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, ws + lo.z, n, dc, 0);
        AS_CHECK_LAUNCH();
    }
    FwdArgs a;
    a.wp = ws + lo.wp; a.bias = biases; a.bias_stride = bstride;
    a.z = ws + lo.z; a.hs = ws + lo.hs; a.cs = ws + lo.cs; a.gates = ws + lo.gates; a.lengths = lengths;
    a.T = T; a.B = B; a.H = H; a.L = L; a.drop = dc;
    a.dbg = getenv("AMDSPEECH_DBG") ? atoi(getenv("AMDSPEECH_DBG")) : 0;
    a.trace = nullptr; a.trace_d = -1;
    if (getenv("AMDSPEECH_TRACE_PTR")) {      // dev-only: address of a device buffer, see tools/trace_step.py
        a.trace = reinterpret_cast<unsigned long long*>(strtoull(getenv("AMDSPEECH_TRACE_PTR"), nullptr, 0));
        a.trace_d = getenv("AMDSPEECH_TRACE_D") ? atoi(getenv("AMDSPEECH_TRACE_D")) : T / 2;
    }
    static const int fwd_nw = getenv("AMDSPEECH_FWD_NW") ? atoi(getenv("AMDSPEECH_FWD_NW")) : 8;
    static const int fwd_un = getenv("AMDSPEECH_FWD_UN") ? atoi(getenv("AMDSPEECH_FWD_UN")) : 8;
    dim3 grid(H / uw, L, ceil_div(B, 32)), block(fwd_nw * 64);
    void (*kern)(FwdArgs) = nullptr;
    static const int fwd_db = getenv("AMDSPEECH_FWD_DB") ? atoi(getenv("AMDSPEECH_FWD_DB")) : 0;
#define FWD_CASE(U, W, N, D) if (uw == U && fwd_nw == W && fwd_un == N && fwd_db == D) kern = lstm_fwd_step<U, W, N, D != 0>;
    FWD_CASE(4, 4, 8, 1) FWD_CASE(4, 8, 8, 0) FWD_CASE(4, 8, 4, 1) FWD_CASE(4, 16, 4, 0) FWD_CASE(4, 8, 16, 0)
    FWD_CASE(8, 4, 4, 1) FWD_CASE(8, 8, 4, 1) FWD_CASE(8, 8, 8, 0) FWD_CASE(8, 16, 4, 0) FWD_CASE(8, 4, 8, 0)
#undef FWD_CASE
    AS_CHECK_ARG(kern != nullptr, "lstm_fwd: no kernel variant for UW=%d NW=%d UN=%d", uw, fwd_nw, fwd_un);
    // Ask for > half of a CU's 160 KiB LDS so that the dispatcher cannot stack two of these
    // MFMA-bound workgroups on one CU while other CUs sit idle (measured: it does otherwise).
    static const int fwd_lds = getenv("AMDSPEECH_FWD_LDS") ? atoi(getenv("AMDSPEECH_FWD_LDS")) : 0;
    if (fwd_lds > 0) AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, fwd_lds));
    prof_begin(0, s);
    for (int dd = 0; dd < T + L - 1; ++dd) {
        a.d = dd;
        hipLaunchKernelGGL(kern, grid, block, fwd_lds, s, a);
    }
    prof_end(0, s, T + L - 1);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

int lstm_bwd(hipStream_t s, const amdspeech_lstm_desc* d, float* ws, const float* kernels, long kstride,
             float* dkernels, float* dbiases, long bstride, const int* lengths) {
    if (int rc = check_desc(d)) return rc;
    AS_CHECK_ARG(ws && kernels && dkernels && dbiases && lengths, "lstm_bwd: null pointer");
    const LstmLayout lo = lstm_layout(d);
    const int T = d->T, B = d->B, H = d->H, L = d->L;
    const long wtotal = (long)L * 2 * H * 4 * H;
    hipLaunchKernelGGL(pack_bwd_kernel, dim3(ceil_div(wtotal, 256)), dim3(256), 0, s, kernels, kstride,
                       ws + lo.wq, H, L);
    AS_CHECK_LAUNCH();
    DropCfg dc{d->keep_in, d->keep_out, d->seed, L};
    BwdArgs a;
    a.wq = ws + lo.wq; a.cs = ws + lo.cs; a.gates = ws + lo.gates; a.dg = ws + lo.dg;
    a.dztop = ws + lo.dztop; a.dc = ws + lo.dc; a.lengths = lengths;
    a.T = T; a.B = B; a.H = H; a.L = L; a.drop = dc;
    static const int bwd_nw = getenv("AMDSPEECH_BWD_NW") ? atoi(getenv("AMDSPEECH_BWD_NW")) : 4;
    static const int bwd_un = getenv("AMDSPEECH_BWD_UN") ? atoi(getenv("AMDSPEECH_BWD_UN")) : 8;
    dim3 grid(H / 16, L, ceil_div(B, 16)), block(bwd_nw * 64);
    void (*kern)(BwdArgs) = nullptr;
    static const int bwd_db = getenv("AMDSPEECH_BWD_DB") ? atoi(getenv("AMDSPEECH_BWD_DB")) : 1;
#define BWD_CASE(W, N, D) if (bwd_nw == W && bwd_un == N && bwd_db == D) kern = lstm_bwd_step<W, N, D != 0>;
    BWD_CASE(4, 8, 1) BWD_CASE(8, 8, 1) BWD_CASE(8, 16, 0) BWD_CASE(16, 8, 0) BWD_CASE(16, 16, 0) BWD_CASE(8, 32, 0)
#undef BWD_CASE
    AS_CHECK_ARG(kern != nullptr, "lstm_bwd: no kernel variant for NW=%d UN=%d", bwd_nw, bwd_un);
    prof_begin(1, s);
    for (int dd = 0; dd < T + L - 1; ++dd) {
        a.d = dd;
        hipLaunchKernelGGL(kern, grid, block, 0, s, a);
    }
    prof_end(1, s, T + L - 1);
    AS_CHECK_LAUNCH();
    // Time-independent weight gradients: dK_l += [Z_l ; Hprev_l]^T . dG_l, db_l += colsum(dG_l)
    const int TB = T * B;
    for (int l = 0; l < L; ++l) {
        const float* dg = ws + lo.dg + (size_t)l * TB * 4 * H;
        const float* zl = ws + lo.z + (size_t)l * TB * H;
        const float* hp = ws + lo.hs + (size_t)l * (T + 1) * B * H;   // slots 0..T-1 = h_{t-1}
        float* dk = dkernels + l * kstride;
        if (int rc = gemm_f32(s, true, false, H, 4 * H, TB, zl, H, dg, 4 * H, dk, 4 * H, nullptr, true)) return rc;
        if (int rc = gemm_f32(s, true, false, H, 4 * H, TB, hp, H, dg, 4 * H, dk + (size_t)H * 4 * H, 4 * H,
                              nullptr, true)) return rc;
        if (int rc = colsum_accumulate(s, dg, TB, 4 * H, 4 * H, dbiases + l * bstride)) return rc;
    }
    // dZ_0 = dG_0 . K_0[0:H,:]^T  (then the layer-0 input dropout mask)
    if (int rc = gemm_f32(s, false, true, TB, H, 4 * H, ws + lo.dg, 4 * H, kernels, 4 * H, ws + lo.dz0, H,
                          nullptr, false)) return rc;
    if (d->keep_in < 1.0f) {
        const long n = (long)TB * H;
        hipLaunchKernelGGL(apply_zmult_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, ws + lo.dz0, n, dc, 0);
        AS_CHECK_LAUNCH();
    }
    return AMDSPEECH_OK;
}

}  // namespace amdspeech

// ------------------------------------------------------------------- C ABI
using namespace amdspeech;

extern "C" int amdspeech_profile_enable(int on) {
    if (on && !g_prof_on) {
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) AS_CHECK_HIP(hipEventCreate(&g_prof_ev[i][j]));
    }
    if (!on && g_prof_on) {
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) (void)hipEventDestroy(g_prof_ev[i][j]);
        g_prof_valid[0] = g_prof_valid[1] = false;
    }
    g_prof_on = on != 0;
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_profile_get(int which, float* elapsed_ms, int* launches) {
    AS_CHECK_ARG(which == 0 || which == 1, "profile_get: which must be 0 or 1");
    AS_CHECK_ARG(elapsed_ms && launches, "profile_get: null pointer");
    AS_CHECK_ARG(g_prof_on && g_prof_valid[which], "profile_get: nothing recorded (enable profiling first)");
    AS_CHECK_HIP(hipEventSynchronize(g_prof_ev[which][1]));
    AS_CHECK_HIP(hipEventElapsedTime(elapsed_ms, g_prof_ev[which][0], g_prof_ev[which][1]));
    *launches = g_prof_launches[which];
    return AMDSPEECH_OK;
}

extern "C" size_t amdspeech_lstm_workspace_bytes(const amdspeech_lstm_desc* d) {
    if (check_desc(d)) return 0;
    return lstm_layout(d).total * sizeof(float);
}

extern "C" void* amdspeech_lstm_ws_ptr(const amdspeech_lstm_desc* d, void* ws, int which) {
    if (check_desc(d) || !ws) return nullptr;
    const LstmLayout lo = lstm_layout(d);
    float* w = static_cast<float*>(ws);
    const size_t tbh = (size_t)d->T * d->B * d->H;
    switch (which) {
        case AMDSPEECH_LSTM_WS_Z0: return w + lo.z;
        case AMDSPEECH_LSTM_WS_ZTOP: return w + lo.z + (size_t)d->L * tbh;
        case AMDSPEECH_LSTM_WS_DZTOP: return w + lo.dztop;
        case AMDSPEECH_LSTM_WS_DZ0: return w + lo.dz0;
        case AMDSPEECH_LSTM_WS_HFINAL: return w + lo.hs + (size_t)d->T * d->B * d->H;
        case AMDSPEECH_LSTM_WS_CFINAL: return w + lo.cs + (size_t)d->T * d->B * d->H;
        default: set_error("lstm_ws_ptr: unknown region %d", which); return nullptr;
    }
}

extern "C" int amdspeech_lstm_fwd(void* stream, const amdspeech_lstm_desc* d, void* ws, const float* kernels,
                                  long kernel_stride, const float* biases, long bias_stride,
                                  const int* lengths, const float* h0, const float* c0) {
    return lstm_fwd(static_cast<hipStream_t>(stream), d, static_cast<float*>(ws), kernels, kernel_stride,
                    biases, bias_stride, lengths, h0, c0);
}

extern "C" int amdspeech_lstm_bwd(void* stream, const amdspeech_lstm_desc* d, void* ws, const float* kernels,
                                  long kernel_stride, float* dkernels, float* dbiases, long bias_stride,
                                  const int* lengths) {
    return lstm_bwd(static_cast<hipStream_t>(stream), d, static_cast<float*>(ws), kernels, kernel_stride,
                    dkernels, dbiases, bias_stride, lengths);
}
