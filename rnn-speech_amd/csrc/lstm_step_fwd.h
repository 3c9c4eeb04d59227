// Part of lstm.hip -- the launch-per-diagonal forward kernel (lstm_fwd_step): the fallback family, DESIGN.md 4.2b.
// Not a standalone header: lstm.hip includes its kernel families in a fixed order, inside namespace amdspeech, after the helpers
// (layout, dropout multipliers, packs) they use.  Tuning macros (#ifndef ...) keep their defaults here; rnn-speech_amd/build.py
// passes overrides for development builds (AMDSPEECH_CXXFLAGS).

// ------------------------------------------------------------- forward step
struct FwdArgs {
    const float* wp; const float* bias; long bias_stride;
    float* z; float* hs; float* cs; float* gates; const int* lengths;
    const float* xp0; float* xp; float* hp;      // packed A-operand panels (see packed_off)
    int T, B, H, L, d, mt0;
    int hoist, l0;   // hoist != 0: ONE layer (l0) per launch at frame t = d; the x half of the product was done by a GEMM
                     // whose result (bias included) waits in gates[l][t] and is replaced there by the activated gates
    DropCfg drop;
    int dbg;   // dev builds only (-DAMDSPEECH_DEVTRACE): timing experiments selected by AMDSPEECH_DBG
    unsigned long long* trace; int trace_d;   // dev builds only: per-wave s_memtime stamps for diagonal trace_d
};
#ifdef AMDSPEECH_DEVTRACE
#define DEV_DBG(a, bit) ((a).dbg & (bit))
#else
#define DEV_DBG(a, bit) 0
#endif

template <int UW, int NW, int UN, bool DB, int MT>   // units/WG, waves/WG, K-blocks per load burst, double buffer, 16-row M tiles/WG
__global__ __launch_bounds__(NW * 64) void lstm_fwd_step(FwdArgs a) {
    constexpr int NT = UW / 4;
    const int l = a.hoist ? a.l0 : blockIdx.y;
    const int t = a.hoist ? a.d : a.d - l;
    if (t < 0 || t >= a.T) return;
    const int ub = blockIdx.x;
    const int tile0 = a.mt0 + blockIdx.z * MT;      // first 16-row batch tile of this workgroup
    const int T = a.T, B = a.B, H = a.H;
    const int nkb = 2 * H / 16, nkb_x = H / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;

    const float* hp = a.hs + ((size_t)l * (T + 1) + t) * B * H;    // h_{t-1}, row-major (epilogue carry-through)
    const int nmt = (B + 15) / 16;
    const size_t bph = (size_t)nmt * 16 * H;
    const int slot = a.d & 1;                                      // produced by the previous diagonal
    const float* xa = (l == 0 ? a.xp0 + (size_t)t * bph : a.xp + ((size_t)l * 2 + slot) * bph) + lane * 4;
    const float* ha = a.hp + ((size_t)l * 2 + slot) * bph + lane * 4;
    const float* wp = a.wp + ((size_t)(l * (H / UW) + ub) * nkb) * (NT * 256) + lane * 4;
#ifdef AMDSPEECH_DEVTRACE
    const bool tracing = a.trace != nullptr && a.d == a.trace_d;
    unsigned long long* tr = a.trace + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * NW + wave) * 16;
#define STAMP(i) do { if (tracing && lane == 0) { tr[i] = __builtin_amdgcn_s_memtime(); if (i == 0) tr[7] = wall_clock64(); if (i == 3) tr[6] = wall_clock64(); } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
    STAMP(0);

    // ---- epilogue operands (bias, previous state, length)
    const float* bias = a.bias + l * a.bias_stride;
    const float* cprev = a.cs + ((size_t)l * (T + 1) + t) * B * H;
    const int pidx = threadIdx.x % (16 * MT * UW);     // (batch row, unit) pair of this thread
    const int pbl = pidx / UW, pu = pidx % UW;
    const int pb = tile0 * 16 + pbl, punit = ub * UW + pu;
    const bool pok = threadIdx.x < 16 * MT * UW && pb < B;
    const int pbc = min(pb, B - 1);               // clamped: unconditional loads, no branches
    // Issued BEFORE the operand bursts (measured: issuing them behind the burst costs 3 us per launch --
    // they then retire last in the in-order vmcnt queue and the epilogue waits for the whole burst).
    float e_bias[4];
    {
        const float* pre = a.gates + ((size_t)l * T + t) * B * 4 * H + (size_t)pbc * 4 * H;      // hoisted: x.W_ih + bias
#pragma unroll
        for (int g = 0; g < 4; ++g) e_bias[g] = a.hoist ? pre[g * H + punit] : bias[g * H + punit];
    }
    const float e_cp = cprev[(size_t)pbc * H + punit];
    const float e_hp = hp[(size_t)pbc * H + punit];
    const int e_len = a.lengths[pbc];

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    size_t tileoff[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        // M tiles past the batch are clamped (loads stay unconditional: a predicated load makes
        // hipcc branch + wait per load); their results are never stored
        tileoff[i] = (size_t)min(tile0 + i, nmt - 1) * (H / 16) * 256;
    }
    (void)li; (void)kq;
    const int kfirst = a.hoist ? nkb_x : 0;                  // hoisted: only the h rows of K are contracted here
    const int kb0 = kfirst + wave * (nkb - kfirst) / NW, kb1 = kfirst + (wave + 1) * (nkb - kfirst) / NW;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    auto load_batch = [&](int kbs, float4 (&av)[UN][MT], float4 (&bv)[UN][NT]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (DEV_DBG(a, 8)) {   // dev-only: MFMAs without loads
#pragma unroll
                for (int i = 0; i < MT; ++i) av[u][i] = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
                for (int j = 0; j < NT; ++j) bv[u][j] = make_float4(0.5f, 0.25f, 0.125f, 1.f);
                continue;
            }
            const bool kok = kbs + u < kb1;
            const int kb = min(kbs + u, kb1 - 1);      // clamped address, data zeroed by select
            const int kba = DEV_DBG(a, 1) ? kb0 : kb, kbb = DEV_DBG(a, 2) ? kb0 : kb;
            const bool isx = kba < nkb_x;
            const float* src = (isx ? xa : ha) + (size_t)(isx ? kba : kba - nkb_x) * 256;
#pragma unroll
            for (int i = 0; i < MT; ++i) av[u][i] = *reinterpret_cast<const float4*>(src + tileoff[i]);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float4 w = *reinterpret_cast<const float4*>(wp + (size_t)(kbb * NT + j) * 256);
                bv[u][j] = kok ? w : zero4;
            }
        }
    };
    auto mma_batch = [&](const float4 (&av)[UN][MT], const float4 (&bv)[UN][NT]) {
#ifdef AMDSPEECH_DEVTRACE
        if (DEV_DBG(a, 4)) {   // dev-only: loads without MFMAs; with bit 16 also stamp each K-block's arrival
#pragma unroll
            for (int u = 0; u < UN; ++u) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j][0] += av[u][i].x * bv[u][j].x + av[u][i].w * bv[u][j].w;
                if (DEV_DBG(a, 16) && tracing && u < 8) {
                    asm volatile("" :: "v"(acc[0][0][0]));
                    const unsigned long long now = __builtin_amdgcn_s_memtime();
                    if (lane == 0) tr[8 + u] = now;     // second 8 slots of a 16-slot record
                }
            }
            return;
        }
#endif
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
        for (int kb = kb0; kb < kb1; kb += UN) {      // (a wave's K range may be empty for small H)
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
        if (nb > 0) load_batch(kb0, a0, b0);          // (a wave's K range may be empty for small H)
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
    const float hv = live ? hn : e_hp;
    const float zv = live ? hn * zmult(a.drop, l + 1, (uint32_t)((size_t)t * B * H + e)) : 0.0f;
    cnext[e] = live ? cn : e_cp;
    hnext[e] = hv;
    zout[e] = zv;
    // packed copies for the next diagonal's MFMA A operands
    const size_t po = packed_off(pb, punit, H);
    a.hp[((size_t)l * 2 + (slot ^ 1)) * bph + po] = hv;
    if (l + 1 < a.L) a.xp[((size_t)(l + 1) * 2 + (slot ^ 1)) * bph + po] = zv;
    STAMP(3);
#undef STAMP
}


