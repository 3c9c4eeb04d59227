// Part of lstm.hip -- lstm_bwd_big (H = 1024: one launch per layer), DESIGN.md 4.2c.
// Not a standalone header: lstm.hip includes its kernel families in a fixed order, inside namespace amdspeech, after the helpers
// (layout, dropout multipliers, packs) they use.  Tuning macros (#ifndef ...) keep their defaults here; rnn-speech_amd/build.py
// passes overrides for development builds (AMDSPEECH_CXXFLAGS).

// ------------------------------------------------- backward, H = 1024: one launch per LAYER, 64 workgroups per batch tile
// The counterpart of lstm_fwd_big: a batch tile's group is the 64 workgroups of an XCD pair, W_hh^T stays in registers for the
// whole sequence (128 VGPRs per wave), the workgroup multiplies dG tiles (LDS -> MFMA A operand) and hands 16x16 partial tiles
// of dh to the workgroups that own those units.  The gradient from the layer above is NOT formed here: lstm_bwd hoists
// dX_{l-1} = dG_l.W_ih^T into one GEMM per layer (into the dztop buffer).
// Rounds 2-3 contracted a workgroup's own 64 gate columns against ALL 1024 output units and handed every one of the group's
// 64 workgroups a partial tile: half of those cross to the other XCD of the pair, so the whole exchange went through memory --
// 64 KiB out and 64 KiB in per workgroup and step, 36.5 GB per layer launch at 4.6 TB/s, 74 % L2 misses (round 3's counters),
// and the hop (write-through store, sc1 load: 2-3 us) sat on the loop-carried path behind ALL the MFMAs: 8.0 us per step.
// Round 4 cuts the product the other way across the pair (6.2 us per step): a workgroup on XCD x of the pair forms the outputs of ITS XCD's 512
// units (32 tiles) from 128 gate columns -- its own dG tile and the tile of its partner (the same ticket on the other XCD).
//   * what crosses XCDs is the INPUT: one 4 KiB dG tile per workgroup and step (16x less than the partials), and it crosses
//     WHILE the own-tile half of the MFMAs runs;
//   * the partial tiles (32 per workgroup) go to the 32 workgroups of the SAME XCD: plain stores, non-temporal loads, served
//     by that XCD's L2 like the rings of lstm_bwd_flow2 (0.95 us per hand-off, 2 MiB of ring per XCD: L2-resident).
// Same registers (W_hh^T[128 gate columns, 64 units] per wave = 128 VGPRs), same MFMA count.  Tags as everywhere: the least
// significant mantissa bit of every exchanged word carries the parity of the slot's use count (two slots each).
struct BigBwdArgs {
    const float* wq; const float* cs; const float* gates; float* dg; const float* dup;      // dup: dZ_top or the hoisted dX [T][B][H]
    float* pring;                  // [2 slots][nmt][2 XCDs][32 consumers][32 producers][256], zeroed before the launch
    float* xring;                  // [2 slots][nmt][64 unit blocks][1024]: dG tiles in MFMA A-fragment order, zeroed before the launch
    const int* lengths; unsigned* err; unsigned* tickets;
    int T, B, H, L, layer;
    DropCfg drop;
    unsigned long long limit;
    // lstm_bwd_big1 only: dG leaves the kernel as the row-major bf16 operand copy [T*B][4H] of the layer's batched products
    // (gemm_bf16p.hip; its transposed copy is made from it) and as its column sums (the bias gradient, accumulated); no f32 dG is written
    unsigned short* dgb; float* dbias;
};
#ifndef BIG_XGATHER_AT
#define BIG_XGATHER_AT 1         // the partner tile's first load goes out after this many quarters (0..3) of the own-tile MFMAs
#endif
#ifndef BIG_QL
#define BIG_QL 2                 // round 6: the product cut a second time INSIDE each XCD of the pair (1: round 4's kernel) -- see lstm_bwd_big
#endif
#ifndef BIG_SETTLE_ALL
#define BIG_SETTLE_ALL 1         // an explicit (free) vmcnt(0) behind the settle: see the step
#endif

// Round 6 cuts it a second time, INSIDE each XCD (BIG_QL = 2; what lstm_bwd_flow2's Q = 2 does inside a group): the local neighbours j and
// j ^ 1 share a K slice of FOUR dG tiles -- their own two and the two of their partners on the other XCD -- and each forms half of
// this XCD's 32 output tiles from it: 2 tiles x 4 K blocks = the same 128 weight VGPRs and 128 MFMAs per wave, 16 instead of 32
// partial tiles out and in per workgroup and step (the P ring's 8.4 MB per time step and layer halves).  The tiles arrive in the
// order of their distance: own (LDS) -> the neighbour's (a second, PLAINLY stored copy through this XCD's L2, ~1 us: under the
// own-tile MFMAs of both wave sets) -> the two remote ones (write-through, sc1, 2-3 us: under the own + neighbour MFMAs); one LDS
// barrier in front of each foreign batch.
template <int PR>             // PR: 0 exact f32, 1 bf16x3, 2 bf16 products
__global__ __launch_bounds__(512) void lstm_bwd_big(BigBwdArgs a) {
    constexpr bool BF3 = PR != 0;
    constexpr int H = 1024, NKB = 4 * H / 16, NRB = 2 * H / 16, NW = 8, NP = 32;      // NP: workgroups (= tiles) per XCD
    constexpr int QL = BIG_QL, NTW = 4 / QL, NKP = 2 * QL, NPR = NP / QL;              // tiles per wave, K blocks per workgroup, producers per consumer
    __shared__ __attribute__((aligned(16))) float a_lds[NKP][1024];          // [own | (neighbour) | partner | (partner's neighbour)][4 m][4 kq][16 i][4 g]: dG tiles as MFMA A fragments
    __shared__ __attribute__((aligned(16))) float red[NW][256];              // partial sums of dh
    __shared__ unsigned s_ticket;
    const int T = a.T, B = a.B, l = a.layer;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nmt = (B + 15) / 16;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xF;
    if (threadIdx.x == 0) s_ticket = atomicAdd(a.tickets + xcc, 1u);
    __syncthreads();
    const int mb = (int)(xcc >> 1), x = (int)(xcc & 1u), j = (int)s_ticket;
    if (mb >= nmt || j >= NP) return;
    const int ub = x * NP + j, pub = (1 - x) * NP + j;      // this workgroup's unit block (epilogue, own dG tile) and its partner's
    const int half = QL == 2 ? (j & 1) : 0;                  // (QL = 2) which half of this XCD's output tiles this workgroup forms
    const unsigned long long t_begin = wall_clock64();

    // W_hh^T fragments: output tile nt = x*32 + half*16 + wave*NTW + n; K block p = the gate columns of unit block kub(p), gate g.
    // Order of the K blocks = order of arrival: own, (local neighbour), partner, (partner's neighbour)
    auto kub = [&](int p) { return QL == 2 ? ((p & 2) ? (1 - x) * NP : x * NP) + (j ^ (p & 1)) : (p ? pub : ub); };
    f32x4 wt[NTW][NKP][4];
    {
        const float* base = a.wq + (size_t)l * NRB * NKB * 256 + lane * 4;
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int p = 0; p < NKP; ++p)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    wt[n][p][g] = *reinterpret_cast<const f32x4*>(base + ((size_t)(H / 16 + x * NP + half * 16 + wave * NTW + n) * NKB + g * (H / 16) + kub(p)) * 256);
    }
    u32x4_f wth[BF3 ? NTW : 1][NKP][2], wtl[BF3 ? NTW : 1][NKP][2];      // split precision: [tile][K block][gate pair]
    if (BF3) {
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int p = 0; p < NKP; ++p)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const float xx[8] = {wt[n][p][2 * sp][0], wt[n][p][2 * sp][1], wt[n][p][2 * sp][2], wt[n][p][2 * sp][3],
                                         wt[n][p][2 * sp + 1][0], wt[n][p][2 * sp + 1][1], wt[n][p][2 * sp + 1][2], wt[n][p][2 * sp + 1][3]};
                    flow_bf3_split(xx, wth[n][p][sp], wtl[n][p][sp]);
                }
    }
    const int bl = (threadIdx.x & 255) >> 4, u = threadIdx.x & 15;
    const int b = mb * 16 + bl, unit = ub * 16 + u;
    const bool epi = wave < 4;
    const bool pok = b < B;
    const int bc = min(b, B - 1);
    const size_t bec = (size_t)bc * H + unit;
    const int len = a.lengths[bc];
    float dcin = 0.0f;
    const int e = ((bl >> 2) * 16 + u) * 4 + (bl & 3);
    const int a_slot = (((u & 3) * 4 + (u >> 2)) * 16 + bl) * 4;

    // P ring of this XCD: [slot][mb][x][consumer][producer][256]
    constexpr unsigned PSLOT = (unsigned)NP * NP * 1024u;                    // bytes per (slot, mb, x)
    const unsigned pslot_stride = (unsigned)nmt * 2u * PSLOT;
    const auto rp = __builtin_amdgcn_make_buffer_rsrc(a.pring, 0, 2u * pslot_stride, 0x00020000);
    const unsigned pbase = (unsigned)(mb * 2 + x) * PSLOT;
    const unsigned gather_off = pbase + (unsigned)(((j * NPR + wave * NTW) * 256 + lane * 4) * 4);        // + q KiB: producer (pair) wave*NTW + q
    const unsigned store_off = pbase + (unsigned)((((half * 16 + wave * NTW) * NPR + j / QL) * 256 + lane * 4) * 4);       // + n*NPR KiB: consumer half*16 + wave*NTW + n
    // (QL = 2) the upper half of this (slot, mb, x) share of the P ring is free: the PLAIN copies of this XCD's dG tiles for their local
    // neighbours live there, [32][1024] floats
    const unsigned lx_base = pbase + (unsigned)NP * NPR * 1024u;
    const unsigned lx_store_off = lx_base + (unsigned)(j * 4096 + a_slot * 4);
    const unsigned lx_load_off = lx_base + (unsigned)((j ^ 1) * 4096 + (wave * 64 + lane) * 8);
    // X ring: [slot][mb][unit block][1024 floats]
    const unsigned xslot_stride = (unsigned)nmt * 64u * 4096u;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(a.xring, 0, 2u * xslot_stride, 0x00020000);
    const unsigned x_store_off = (unsigned)((mb * 64 + ub) * 4096 + a_slot * 4);                          // this thread's four gates (epilogue threads)
    const unsigned x_load_off = (unsigned)((mb * 64 + pub) * 4096 + (wave * 64 + lane) * 8);              // this lane's 8 bytes of the partner tile
    const unsigned x_load_off2 = (unsigned)((mb * 64 + (pub ^ 1)) * 4096 + (wave * 64 + lane) * 8);       // (QL = 2) ... and of the partner's neighbour's
    bool dead = false;
    u32x4_f gt[NTW];
    auto issue = [&](int slot) {
#pragma unroll
        for (int q = 0; q < NTW; ++q)
            gt[q] = __builtin_amdgcn_raw_buffer_load_b128(rp, gather_off + (unsigned)(q * 1024), (unsigned)slot * pslot_stride, 2);      // nt: this XCD's L2
    };
    auto settle = [&](int slot, unsigned par) {      // (first check straight-line, the retry loop behind it: see lstm_fwd_flow2)
        bool again = false;
#pragma unroll
        for (int q = 0; q < NTW; ++q) again = again || flow_untagged(gt[q], par);
        if (__any(again) && !dead) {
            while (true) {
                if (wall_clock64() - t_begin > a.limit) { dead = true; if (lane == 0) atomicOr(a.err, 2u); break; }
                issue(slot);
                again = false;
#pragma unroll
                for (int q = 0; q < NTW; ++q) again = again || flow_untagged(gt[q], par);
                if (!__any(again)) break;
            }
        }
    };
    u32x2_f gx, gx2 = {0u, 0u}, gl = {0u, 0u};      // this lane's 8 bytes of the partner's dG tile (, of the partner's neighbour's, of the local neighbour's)
    auto issue_x = [&](int slot) {
        gx = __builtin_amdgcn_raw_buffer_load_b64(rx, x_load_off, (unsigned)slot * xslot_stride, 16);      // sc1: written by the other XCD
        if (QL == 2) gx2 = __builtin_amdgcn_raw_buffer_load_b64(rx, x_load_off2, (unsigned)slot * xslot_stride, 16);
    };
    auto issue_l = [&](int slot) {
        gl = __builtin_amdgcn_raw_buffer_load_b64(rp, lx_load_off, (unsigned)slot * pslot_stride, 2);      // nt: this XCD's L2
    };
    auto stale2 = [](const u32x2_f v, unsigned par) { return (((v[0] ^ par) | (v[1] ^ par)) & 1u) != 0u; };
    auto settle_x = [&](int slot, unsigned par) {
        bool again = stale2(gx, par) || (QL == 2 && stale2(gx2, par));
        if (__any(again) && !dead) {
            while (true) {
                if (wall_clock64() - t_begin > a.limit) { dead = true; if (lane == 0) atomicOr(a.err, 2u); break; }
                issue_x(slot);
                again = stale2(gx, par) || (QL == 2 && stale2(gx2, par));
                if (!__any(again)) break;
            }
        }
    };
    auto settle_l = [&](int slot, unsigned par) {
        bool again = stale2(gl, par);
        if (__any(again) && !dead) {
            while (true) {
                if (wall_clock64() - t_begin > a.limit) { dead = true; if (lane == 0) atomicOr(a.err, 2u); break; }
                issue_l(slot);
                again = stale2(gl, par);
                if (!__any(again)) break;
            }
        }
    };
    auto parity = [&](int t) -> unsigned { return ((((unsigned)(T - 1 - t)) >> 1) & 1u) ^ 1u; };
    auto ftanh = [](float xv) {
        const float x2 = xv * xv;
        const float small = xv * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f - 0.053968254f * x2)));
        const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * xv));
        return fabsf(xv) < 0.25f ? small : big;
    };
    auto mma_half = [&](f32x4 (&acc)[NTW], const f32x4 (&av)[4], const int p, auto mid) __attribute__((always_inline)) {
        if (BF3) {
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const float xx[8] = {av[0][2 * sp], av[1][2 * sp], av[2][2 * sp], av[3][2 * sp],
                                     av[0][2 * sp + 1], av[1][2 * sp + 1], av[2][2 * sp + 1], av[3][2 * sp + 1]};
                u32x4_f ah, al;
                flow_bf3_split(xx, ah, al);
#pragma unroll
                for (int n = 0; n < NTW; ++n) acc[n] = flow_bf_mma<PR>(acc[n], ah, al, wth[BF3 ? n : 0][p][sp], wtl[BF3 ? n : 0][p][sp]);
                if (sp == 0) mid();
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g == BIG_XGATHER_AT) mid();
#pragma unroll
                for (int n = 0; n < NTW; ++n) {
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][g], wt[n][p][g][0], acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][g], wt[n][p][g][1], acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2][g], wt[n][p][g][2], acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3][g], wt[n][p][g][3], acc[n], 0, 0, 0);
                }
            }
        }
    };
    const auto rdg = __builtin_amdgcn_make_buffer_rsrc(a.dg + (size_t)l * T * B * 4 * H, 0, (unsigned)((size_t)T * B * 4 * H * 4), 0x00020000);
    FLOW_WEIGHTS_RESIDENT();
    for (int t = T - 1; t >= 0; --t) {
        const unsigned par = parity(t);
        // forward stash and the gradient arriving from above for this frame (needed after the gather)
        const float* gr = a.gates + ((size_t)l * T + t) * B * 4 * H + (size_t)bc * 4 * H + unit;
        const float gi = gr[0], gj = gr[H], gf = gr[2 * H], go = gr[3 * H];
        const float c = a.cs[((size_t)l * (T + 1) + t + 1) * B * H + bec];
        const float cp = a.cs[((size_t)l * (T + 1) + t) * B * H + bec];
        const float dup = a.dup[(size_t)t * B * H + bec];
        // ---- the partial tiles of step t+1 addressed to this workgroup (gather issued at the end of step t+1)
        f32x4 sr = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (t + 1 < T) {
            settle((t + 1) & 1, parity(t + 1));
#pragma unroll
            for (int q = 0; q < NTW; ++q)
                sr += (f32x4){__uint_as_float(gt[q][0]), __uint_as_float(gt[q][1]), __uint_as_float(gt[q][2]), __uint_as_float(gt[q][3])};
        }
        *reinterpret_cast<f32x4*>(&red[wave][lane * 4]) = sr;
#if BIG_SETTLE_ALL
        // (the gathered tiles were the youngest memory operations in flight, so this waits for nothing -- but it tells hipcc that
        //  the stash loads above have landed in EVERY wave: waves 4-7 never use theirs, and the "still pending" state they carried
        //  to the merge behind the epilogue made the A-fragment reads behind B2 wait for vmcnt(0) -- at run time, in waves 0-3,
        //  for the write-through store of the tile to the partner XCD they had just issued)
        __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
        lds_barrier();
        if (epi) {
            float dh = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) dh += red[w][e];
            dh += dup * zmult(a.drop, l + 1, (uint32_t)((size_t)t * B * H + bec));
            const bool live = pok && t < len;
            const float tc = ftanh(c);
            const float dct = dcin + dh * go * (1.0f - tc * tc);
            f32x4 dgv;
            dgv[0] = dct * gj * gi * (1.0f - gi);
            dgv[1] = dct * gi * (1.0f - gj * gj);
            dgv[2] = dct * cp * gf * (1.0f - gf);
            dgv[3] = dh * tc * go * (1.0f - go);
            float dcout = dct * gf;
            if (!live) { dgv = (f32x4){0.f, 0.f, 0.f, 0.f}; dcout = 0.0f; }
            // the tile's way to the partner starts HERE, before anything else of the step: write-through, tagged
            if (t > 0) {
                __builtin_amdgcn_raw_buffer_store_b128(flow_tag(dgv, par), rx, x_store_off + (unsigned)(t & 1) * xslot_stride, 0, 16);
                if (QL == 2)      // ... and a plain copy for the local neighbour (the L2's copy of a write-through line follows late: round 4's forward experiment)
                    __builtin_amdgcn_raw_buffer_store_b128(flow_tag(dgv, par), rp, lx_store_off + (unsigned)(t & 1) * pslot_stride, 0, 0);
            }
            *reinterpret_cast<f32x4*>(&a_lds[0][a_slot]) = dgv;
            dcin = dcout;
        }
        lds_barrier();
        f32x4 av[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<const f32x4*>(&a_lds[0][(m * 64 + lane) * 4]);
        if (!epi && pok) {
            // row-major dG[t] for the weight-gradient GEMMs and the hoisted down product (they run after this kernel)
            const int g = u >> 2, q4 = u & 3;
            u32x4_f row;
#pragma unroll
            for (int m = 0; m < 4; ++m) row[m] = __float_as_uint(a_lds[0][((m * 4 + q4) * 16 + bl) * 4 + g]);
            __builtin_amdgcn_raw_buffer_store_b128(row, rdg, (unsigned)((((size_t)t * B + b) * 4 * H + g * H + ub * 16 + q4 * 4) * 4), 0, 0);
        }
        if (t > 0) {
            f32x4 acc[NTW];
#pragma unroll
            for (int n = 0; n < NTW; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // ---- own tile (the others are on their way: the nearest is requested first -- loads retire in order)
            mma_half(acc, av, 0, [&]() __attribute__((always_inline)) {      // (part-way through: see BIG_XGATHER_AT)
                __builtin_amdgcn_sched_barrier(0);
                if (QL == 2) issue_l(t & 1);
                issue_x(t & 1);
                __builtin_amdgcn_sched_barrier(0);
            });
            __builtin_amdgcn_sched_barrier(0);
            if (QL == 2) {
                // ---- the local neighbour's tile (through this XCD's L2): 8 bytes per lane -> LDS -> everybody's A fragments
                settle_l(t & 1, par);
                *reinterpret_cast<u32x2_f*>(&a_lds[1][(wave * 64 + lane) * 2]) = gl;
                lds_barrier();
#pragma unroll
                for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<const f32x4*>(&a_lds[1][(m * 64 + lane) * 4]);
                mma_half(acc, av, 1, []() {});
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- the tile(s) from the other XCD
            settle_x(t & 1, par);
            *reinterpret_cast<u32x2_f*>(&a_lds[QL][(wave * 64 + lane) * 2]) = gx;
            if (QL == 2) *reinterpret_cast<u32x2_f*>(&a_lds[3][(wave * 64 + lane) * 2]) = gx2;
            lds_barrier();
#pragma unroll
            for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<const f32x4*>(&a_lds[QL][(m * 64 + lane) * 4]);
            mma_half(acc, av, QL, []() {});
            if (QL == 2) {
#pragma unroll
                for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<const f32x4*>(&a_lds[3][(m * 64 + lane) * 4]);
                mma_half(acc, av, 3, []() {});
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NTW; ++n)      // (slot offset in voffset, not soffset: see store_tiles in lstm_bwd_flow2)
                __builtin_amdgcn_raw_buffer_store_b128(flow_tag(acc[n], par), rp,
                                                       store_off + (unsigned)(n * NPR * 1024) + (unsigned)(t & 1) * pslot_stride, 0, 0);
            issue(t & 1);            // the next step's operand: most of it is there when the stash loads above have come back
        }
    }
}


// ------------------------------------------------- backward, H = 1024 in plain bf16: a batch tile's group on ONE XCD (round 5)
// The counterpart of lstm_fwd_big1.  As bf16 W_hh^T fits the registers of 32 CUs, so a batch tile's group is the 32 workgroups of ONE
// XCD, and two stacks of one shape (amdspeech_lstm_bwd_pair) run their layers side by side, stack 0 on XCDs 0 - 3, stack 1 on 4 - 7.
// Workgroup j owns the unit blocks 2j, 2j + 1 (all eight waves run the epilogue: 16 rows x 32 units).  The product is cut in TWO
// directions, Q parts of the output units x 32/Q slices of the gate columns:
//   * the Q workgroups ks*Q .. ks*Q + Q - 1 share K slice ks = the gate columns of THEIR 2Q unit blocks, and workgroup (ks, nq)
//     contracts that slice against the output tiles of part nq (64/Q tiles; wave w: 8/Q of them) -- Q * 128 gate columns x 1024/Q
//     units of W_hh^T as bf16 = the same 128 VGPRs per wave, 32 MFMAs (16x16x32 bf16) per wave and step, whatever Q;
//   * what the Q workgroups of a slice exchange is the INPUT: every workgroup stores its two dG tiles (8 KiB, tagged f32, in the
//     A-fragment order of the LDS image) to the X ring and reads the 2(Q - 1) tiles of the others while its own blocks' MFMAs run;
//   * the partial tiles (64/Q per workgroup) go to the workgroups that own those units, which gather 2 x 32/Q of them: 64/Q KiB out and
//     in per workgroup and step.
// Everything stays inside the XCD: plain stores, non-temporal loads through its L2, parity tags in the least significant mantissa bit,
// two slots per ring (a workgroup stores the tiles of step t - 2 only after it has gathered the partial tiles of step t - 1, which
// its slice partners form after they have read its tiles of step t).
// Q is the lever.  Q = 1 (no X ring: every workgroup hands all 64 tiles out) moves 4 MB of partial tiles through the XCD's L2 per step
// and THAT is the step: 5.8 us for two stacks side by side (29.4 ms of backward recurrence per configs[4] step), 3.3 us with half of
// the tiles left out (wrong results; with the MFMAs left out instead: 5.6) -- against 3.95 on the XCD pairs of lstm_bwd_big.  Q = 2:
// 23.1 ms; Q = 4: 17.7 ms = 3.55 us per step for BOTH stacks (40 KiB in, 24 KiB out per workgroup and step; Q = 8 would read more
// than it saves: 64 in, 16 out).  One stack alone on four XCDs: 16.95 ms at configs[2]'s shape where lstm_bwd_big<2> takes 19.7 on
// all eight -- so this is the backward kernel of precision 2 at 1024 units whenever its batched products run on the bf16 operand
// copies (AMDSPEECH_BIG1=0: the XCD pairs).
// (Also measured, and removed: BOTH stacks on the XCD pairs with two workgroups per CU -- as bf16 a wave's weights are 64 VGPRs and
//  lstm_fwd_big / lstm_bwd_big <2> compile to 128 registers with one or two scratch accesses per step, so a launch of 512 workgroups
//  puts a workgroup of either stack on every CU.  Alone that build runs at 16.0 / 19.0 ms of forward / backward recurrence per
//  configs[2] step (13.2 / 19.8 at 256 registers); side by side the two stacks take 34.8 / 39.6 ms -- more than one after the other.  A
//  step is not idle while it waits: every poll round of a workgroup re-reads its whole operand (64 KiB forward) from memory, and
//  twice the pollers saturate that path.)
#ifndef BIG1_Q
#define BIG1_Q 4
#endif
struct BigBwd1Args {
    BigBwdArgs b[2];
    int n;                         // stacks in this launch: 1, or 2 (stack 1 on the XCDs from 4 up)
};
template <int Q>
__global__ __launch_bounds__(512) void lstm_bwd_big1(BigBwd1Args a1) {
    constexpr int H = 1024, NKB = 4 * H / 16, NRB = 2 * H / 16, NW = 8, NP = 32;      // NP: workgroups per XCD
    constexpr int NKS = NP / Q;            // K slices = producers of an output tile
    constexpr int NB = 2 * Q;              // unit blocks of a K slice (local block 0, 1: this workgroup's own)
    constexpr int TPW = 64 / Q / NW;       // output tiles per wave
    constexpr int PPW = NKS / NW;          // producers a wave gathers (x 2 tiles)
    static_assert(Q == 1 || Q == 2 || Q == 4, "lstm_bwd_big1: 1, 2 or 4 parts of the output units");
    __shared__ __attribute__((aligned(16))) float a_lds[NB][1024];           // [local unit block][4 m][4 kq][16 i][4 g]: dG tiles as MFMA A fragments
    __shared__ __attribute__((aligned(16))) float red[NW][2][256];           // partial sums of dh
    __shared__ unsigned s_ticket;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xF;
    const bool second = a1.n == 2 && xcc >= 4u;
    const BigBwdArgs a = second ? a1.b[1] : a1.b[0];
    const int T = a.T, B = a.B, l = a.layer;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nmt = (B + 15) / 16;
    if (threadIdx.x == 0) s_ticket = atomicAdd(a.tickets + xcc, 1u);
    __syncthreads();
    const int mb = (int)xcc - (second ? 4 : 0), j = __builtin_amdgcn_readfirstlane((int)s_ticket);
    if (mb >= nmt || j >= NP) return;
    const int ks = j / Q, nq = j % Q;
    const unsigned long long t_begin = wall_clock64();

    // W_hh^T fragments as bf16: output tile nt = nq*(64/Q) + wave*TPW + n, K = the gate columns of the slice's unit block
    // 2Q ks + (2 nq + lb) % 2Q (local block lb: 0, 1 are this workgroup's own), gate pair sp
    u32x4_f wth[TPW][NB][2];
    {
        const float* base = a.wq + (size_t)l * NRB * NKB * 256 + lane * 4;
#pragma unroll
        for (int n = 0; n < TPW; ++n)
#pragma unroll
            for (int lb = 0; lb < NB; ++lb)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const int ub_k = NB * ks + (2 * nq + lb) % NB;
                    const size_t row = (size_t)(H / 16 + nq * (64 / Q) + wave * TPW + n) * NKB;
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(base + (row + (2 * sp) * (H / 16) + ub_k) * 256);
                    const f32x4 w1 = *reinterpret_cast<const f32x4*>(base + (row + (2 * sp + 1) * (H / 16) + ub_k) * 256);
                    const float xx[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
                    u32x4_f lo_unused;
                    flow_bf3_split(xx, wth[n][lb][sp], lo_unused);
                }
    }
    const int hb = threadIdx.x >> 8;                                        // which unit block of the pair this thread's element is in
    const int bl = (threadIdx.x & 255) >> 4, u = threadIdx.x & 15;
    const int ub = 2 * j + hb;
    const int b = mb * 16 + bl, unit = ub * 16 + u;
    const bool pok = b < B;
    const int bc = min(b, B - 1);
    const size_t bec = (size_t)bc * H + unit;
    const int len = a.lengths[bc];
    float dcin = 0.0f;
    const int e = ((bl >> 2) * 16 + u) * 4 + (bl & 3);
    const int a_slot = (((u & 3) * 4 + (u >> 2)) * 16 + bl) * 4;

    // P ring of this XCD: [slot][mb][output tile = consumer*2 + unit block of its pair][producer slice][256]
    constexpr unsigned PSLOT = 64u * (unsigned)NKS * 1024u;                 // bytes per (slot, mb)
    const unsigned pslot_stride = (unsigned)nmt * PSLOT;
    const auto rp = __builtin_amdgcn_make_buffer_rsrc(a.pring, 0, 2u * pslot_stride, 0x00020000);
    const unsigned pbase = (unsigned)mb * PSLOT;
    const unsigned gather_off = pbase + (unsigned)((((j * 2) * NKS + wave * PPW) * 256 + lane * 4) * 4);            // + h*NKS KiB + q KiB: slice wave*PPW + q
    const unsigned store_off = pbase + (unsigned)((((nq * (64 / Q) + wave * TPW) * NKS + ks) * 256 + lane * 4) * 4);      // + n*NKS KiB: tile + n
    // X ring: [slot][mb][unit block][1024 floats], a tile in the order of the LDS image
    const unsigned xslot_stride = (unsigned)nmt * 64u * 4096u;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(a.xring, 0, Q > 1 ? 2u * xslot_stride : 0u, 0x00020000);
    const unsigned x_store_off = (unsigned)((mb * 64 + ub) * 4096 + a_slot * 4);
    unsigned x_load_off[Q > 1 ? Q - 1 : 1];      // load r: local blocks 2 + 2r (threads 0-255), 3 + 2r (threads 256-511); 16 bytes per thread
#pragma unroll
    for (int r = 0; r < Q - 1; ++r)
        x_load_off[r] = (unsigned)((mb * 64 + NB * ks + (2 * nq + 2 + 2 * r + hb) % NB) * 4096 + (threadIdx.x & 255) * 16);
    bool dead = false;
    u32x4_f gt[2][PPW];
    auto issue = [&](int slot) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < PPW; ++q)
                gt[h][q] = __builtin_amdgcn_raw_buffer_load_b128(rp, gather_off + (unsigned)(h * NKS * 1024 + q * 1024), (unsigned)slot * pslot_stride, 2);      // nt: this XCD's L2
    };
    auto settle = [&](int slot, unsigned par) {
        bool again = false;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < PPW; ++q) again = again || flow_untagged(gt[h][q], par);
        if (__any(again) && !dead) {
            while (true) {
                if (wall_clock64() - t_begin > a.limit) { dead = true; if (lane == 0) atomicOr(a.err, 2u); break; }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < PPW; ++q)
                        if (__any(flow_untagged(gt[h][q], par)))      // (only the slices whose tiles still carry the old tag)
                            gt[h][q] = __builtin_amdgcn_raw_buffer_load_b128(rp, gather_off + (unsigned)(h * NKS * 1024 + q * 1024), (unsigned)slot * pslot_stride, 2);
                again = false;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < PPW; ++q) again = again || flow_untagged(gt[h][q], par);
                if (!__any(again)) break;
            }
        }
    };
    u32x4_f gx[Q > 1 ? Q - 1 : 1];               // this thread's 16 bytes of the slice partners' dG tiles
    auto issue_x = [&](int slot) {
#pragma unroll
        for (int r = 0; r < Q - 1; ++r) gx[r] = __builtin_amdgcn_raw_buffer_load_b128(rx, x_load_off[r], (unsigned)slot * xslot_stride, 2);
    };
    auto settle_x = [&](int slot, unsigned par) {
        bool again = false;
#pragma unroll
        for (int r = 0; r < Q - 1; ++r) again = again || flow_untagged(gx[r], par);
        if (__any(again) && !dead) {
            while (true) {
                if (wall_clock64() - t_begin > a.limit) { dead = true; if (lane == 0) atomicOr(a.err, 2u); break; }
#pragma unroll
                for (int r = 0; r < Q - 1; ++r)
                    if (__any(flow_untagged(gx[r], par))) gx[r] = __builtin_amdgcn_raw_buffer_load_b128(rx, x_load_off[r], (unsigned)slot * xslot_stride, 2);
                again = false;
#pragma unroll
                for (int r = 0; r < Q - 1; ++r) again = again || flow_untagged(gx[r], par);
                if (!__any(again)) break;
            }
        }
    };
    auto parity = [&](int t) -> unsigned { return ((((unsigned)(T - 1 - t)) >> 1) & 1u) ^ 1u; };
    auto ftanh = [](float xv) {
        const float x2 = xv * xv;
        const float small = xv * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f - 0.053968254f * x2)));
        const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * xv));
        return fabsf(xv) < 0.25f ? small : big;
    };
    f32x4 acc[TPW];
    auto mma_block = [&](const int lb) __attribute__((always_inline)) {      // the slice's local unit block lb against this wave's tiles
        f32x4 av[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<const f32x4*>(&a_lds[lb][(m * 64 + lane) * 4]);
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const float xx[8] = {av[0][2 * sp], av[1][2 * sp], av[2][2 * sp], av[3][2 * sp],
                                 av[0][2 * sp + 1], av[1][2 * sp + 1], av[2][2 * sp + 1], av[3][2 * sp + 1]};
            u32x4_f ah, al;
            flow_bf3_split(xx, ah, al);
#pragma unroll
            for (int n = 0; n < TPW; ++n) acc[n] = flow_bf_mma<2>(acc[n], ah, al, wth[n][lb][sp], wth[n][lb][sp]);
        }
    };
    // dG out: the bf16 row-major operand copy (8 bytes per thread and step), column sums in registers
    const size_t TB = (size_t)T * B;
    const auto rdgb = __builtin_amdgcn_make_buffer_rsrc(a.dgb, 0, (unsigned)(TB * 4 * H * 2), 0x00020000);
    f32x4 csum = (f32x4){0.f, 0.f, 0.f, 0.f};
    FLOW_WEIGHTS_RESIDENT();
    for (int t = T - 1; t >= 0; --t) {
        const unsigned par = parity(t);
        // forward stash and the gradient arriving from above for this frame (needed after the gather)
        const float* gr = a.gates + ((size_t)l * T + t) * B * 4 * H + (size_t)bc * 4 * H + unit;
        const float gi = gr[0], gj = gr[H], gf = gr[2 * H], go = gr[3 * H];
        const float c = a.cs[((size_t)l * (T + 1) + t + 1) * B * H + bec];
        const float cp = a.cs[((size_t)l * (T + 1) + t) * B * H + bec];
        const float dup = a.dup[(size_t)t * B * H + bec];
        // ---- the partial tiles of step t+1 addressed to this workgroup (gather issued at the end of step t+1)
        f32x4 sr[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        if (t + 1 < T) {
            settle((t + 1) & 1, parity(t + 1));
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < PPW; ++q)
                    sr[h] += (f32x4){__uint_as_float(gt[h][q][0]), __uint_as_float(gt[h][q][1]), __uint_as_float(gt[h][q][2]), __uint_as_float(gt[h][q][3])};
        }
        *reinterpret_cast<f32x4*>(&red[wave][0][lane * 4]) = sr[0];
        *reinterpret_cast<f32x4*>(&red[wave][1][lane * 4]) = sr[1];
        lds_barrier();
        {
            float dh = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) dh += red[w][hb][e];
            dh += dup * zmult(a.drop, l + 1, (uint32_t)((size_t)t * B * H + bec));
            const bool live = pok && t < len;
            const float tc = ftanh(c);
            const float dct = dcin + dh * go * (1.0f - tc * tc);
            f32x4 dgv;
            dgv[0] = dct * gj * gi * (1.0f - gi);
            dgv[1] = dct * gi * (1.0f - gj * gj);
            dgv[2] = dct * cp * gf * (1.0f - gf);
            dgv[3] = dh * tc * go * (1.0f - go);
            float dcout = dct * gf;
            if (!live) { dgv = (f32x4){0.f, 0.f, 0.f, 0.f}; dcout = 0.0f; }
            // the tile's way to the slice partners starts HERE, before anything else of the step
            if (Q > 1 && t > 0) __builtin_amdgcn_raw_buffer_store_b128(flow_tag(dgv, par), rx, x_store_off + (unsigned)(t & 1) * xslot_stride, 0, 0);
            *reinterpret_cast<f32x4*>(&a_lds[hb][a_slot]) = dgv;
            dcin = dcout;
            csum += dgv;
        }
        lds_barrier();
        {
            // dG[t] for the batched products that run after this kernel, as their bf16 operand: four units of a row.  (The transposed
            // copy too -- four rows of a column, 8 bytes per thread -- was measured: 32-byte pieces 128 bytes apart cost the step
            // 0.9 us, 17.7 -> 22.0 ms per configs[4] step, more than the pass over the bf16 copy that makes it afterwards.)
            auto pack2 = [](float x0, float x1) -> unsigned {
                const flow_f32x2 v = {x0, x1};
                return __builtin_bit_cast(unsigned, __builtin_convertvector(v, flow_bf16x2));
            };
            if (pok) {
                const int g = u >> 2, q4 = u & 3;
                u32x2_f row;
                row[0] = pack2(a_lds[hb][((0 * 4 + q4) * 16 + bl) * 4 + g], a_lds[hb][((1 * 4 + q4) * 16 + bl) * 4 + g]);
                row[1] = pack2(a_lds[hb][((2 * 4 + q4) * 16 + bl) * 4 + g], a_lds[hb][((3 * 4 + q4) * 16 + bl) * 4 + g]);
                __builtin_amdgcn_raw_buffer_store_b64(row, rdgb, (unsigned)((((size_t)t * B + b) * 4 * H + g * H + ub * 16 + q4 * 4) * 2), 0, 0);
            }
        }
        if (t > 0) {
#pragma unroll
            for (int n = 0; n < TPW; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // ---- this workgroup's own unit blocks (the partners' tiles are on their way)
            mma_block(0);
            if (Q > 1) {
                __builtin_amdgcn_sched_barrier(0);
                issue_x(t & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            mma_block(1);
            if (Q > 1) {
                __builtin_amdgcn_sched_barrier(0);
                // ---- the slice partners' tiles: 16 bytes per thread and load -> LDS -> everybody's A fragments
                settle_x(t & 1, par);
#pragma unroll
                for (int r = 0; r < Q - 1; ++r) *reinterpret_cast<u32x4_f*>(&a_lds[2 + 2 * r + hb][(threadIdx.x & 255) * 4]) = gx[r];
                lds_barrier();
#pragma unroll
                for (int lb = 2; lb < NB; ++lb) mma_block(lb);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < TPW; ++n)      // (slot offset in voffset, not soffset: see store_tiles in lstm_bwd_flow2)
                __builtin_amdgcn_raw_buffer_store_b128(flow_tag(acc[n], par), rp,
                                                       store_off + (unsigned)(n * NKS * 1024) + (unsigned)(t & 1) * pslot_stride, 0, 0);
            issue(t & 1);            // the next step's operand: most of it is there when the stash loads above have come back
        }
    }
    if (pok) {      // the bias gradient: this thread's (row, unit) over all frames; 16 rows x the batch tiles meet in every address
#pragma unroll
        for (int g = 0; g < 4; ++g) atomicAdd(a.dbias + g * H + unit, csum[g]);
    }
}
