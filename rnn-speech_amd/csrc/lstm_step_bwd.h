// Part of lstm.hip -- the launch-per-diagonal backward kernel (lstm_bwd_step), DESIGN.md 4.2b.
// Not a standalone header: lstm.hip includes its kernel families in a fixed order, inside namespace amdspeech, after the helpers
// (layout, dropout multipliers, packs) they use.  Tuning macros (#ifndef ...) keep their defaults here; rnn-speech_amd/build.py
// passes overrides for development builds (AMDSPEECH_CXXFLAGS).

// ------------------------------------------------------------ backward step
struct BwdArgs {
    const float* wq; const float* cs; const float* gates; float* dg; const float* dztop; float* dc;
    float* dgp;                                   // packed dG ring [L][2][bp*4H]
    const int* lengths;
    int T, B, H, L, d, mt0;
    int hoist, l0;   // hoist != 0: ONE layer (l0) per launch at frame t = T-1-d; the gradient from the layer above was formed by
                     // a GEMM and waits in dztop (like the top layer's), so only the recurrent product is left here
    DropCfg drop;
};

template <int NW, int UN, bool DB>    // waves per workgroup, virtual K-blocks per load burst, double buffer
__global__ __launch_bounds__(NW * 64) void lstm_bwd_step(BwdArgs a) {
    const int l = a.hoist ? a.l0 : blockIdx.y;
    const int T = a.T, B = a.B, H = a.H, L = a.L;
    const int t = a.hoist ? (T - 1) - a.d : (T - 1) - (a.d - (L - 1 - l));
    if (t < 0 || t >= T) return;
    const int ub = blockIdx.x, mb = a.mt0 + blockIdx.z;      // 16 units x 16 batch rows
    const int nkb = 4 * H / 16, nrb = 2 * H / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nmt = (B + 15) / 16;
    const size_t bpg = (size_t)nmt * 16 * 4 * H;
    const int slot = a.d & 1;
    const bool has_rec = t + 1 < T, has_up = !a.hoist && l + 1 < L;

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
    a_src0 = a.dgp + ((size_t)l * 2 + slot) * bpg + (size_t)mb * nkb * 256 + lane * 4;        // dG_l[t+1]
    a_src1 = a.dgp + ((size_t)(l + 1) * 2 + slot) * bpg + (size_t)mb * nkb * 256 + lane * 4;  // dG_{l+1}[t]
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
            av[q] = *reinterpret_cast<const float4*>((sidx ? a_src1 : a_src0) + (size_t)kb * 256);
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
    // packed copy for the next diagonal (this layer's recurrent stream, the layer below's "up" stream)
    float* dgpw = a.dgp + ((size_t)l * 2 + (slot ^ 1)) * bpg;
    dgpw[packed_off(b, unit, 4 * H)] = dgi;
    dgpw[packed_off(b, H + unit, 4 * H)] = dgj;
    dgpw[packed_off(b, 2 * H + unit, 4 * H)] = dgf;
    dgpw[packed_off(b, 3 * H + unit, 4 * H)] = dgo;
}

