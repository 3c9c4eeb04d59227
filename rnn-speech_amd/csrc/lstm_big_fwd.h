// Part of lstm.hip -- lstm_fwd_big (H = 1024: one weight-stationary launch per layer), DESIGN.md 4.2c.
// Not a standalone header: lstm.hip includes its kernel families in a fixed order, inside namespace amdspeech, after the helpers
// (layout, dropout multipliers, packs) they use.  Tuning macros (#ifndef ...) keep their defaults here; rnn-speech_amd/build.py
// passes overrides for development builds (AMDSPEECH_CXXFLAGS).

// ------------------------------------------------- forward, H = 1024: one launch per LAYER, 64 workgroups per batch tile
// The h half of a 1024-wide layer's kernel is 16 MB: it fits the registers of 64 CUs, i.e. TWO XCDs.  The x half does not fit
// beside it, so it is hoisted: one GEMM per layer forms x.W_ih + b for all T frames (into `gates`, see lstm_fwd), and this
// kernel keeps W_hh on chip for the whole sequence and runs the recurrence of ONE layer:
//   * group = batch tile mb = the 64 workgroups of XCDs (2mb, 2mb+1); workgroup ub owns 16 units x 4 gates and, per wave,
//     a K slice of 128 rows of W_hh (128 VGPRs);
//   * the loop-carried panel h_{t-1} [16 x 1024] travels through a 2-slot ring in MEMORY (the group spans two XCDs whose L2s
//     are not coherent: write-through stores, sc1 loads); every workgroup contributes its 16x16 tile and reads the whole
//     64 KiB panel.  As in lstm_bwd_flow2 the flag is the least significant mantissa bit of every word (parity of the
//     slot's use count), so nothing has to be reset or counted;
//   * step: settle h_{t-1} -> 128 MFMAs per wave -> K-split partial sums to LDS -> barrier -> waves 0-3: epilogue (adds
//     the hoisted row, gates, c, h; BPTT stash; h tile out) -> barrier.
#ifndef BIG_POLL_DELAY
#define BIG_POLL_DELAY 16
#endif
struct BigFwdArgs {
    const float* wp; float* z; float* hs; float* cs; float* gates; const int* lengths;
    float* hring;                  // [2 slots][nmt][H/16][256]: packed h panels of this layer (slot 0 = initial state, tagged)
    unsigned* err; unsigned* tickets;
    int T, B, H, L, layer;
    DropCfg drop;
    unsigned long long limit;
};

__global__ void tag_panel_kernel(float* p, size_t n, unsigned par) {      // host-packed initial state: give every word its tag
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = __uint_as_float((__float_as_uint(p[i]) & ~1u) | par);
}

template <int PR>             // PR: 0 exact f32, 1 bf16x3, 2 bf16 products (desc.precision), fragments split in registers as in lstm_fwd_flow2
__global__ __launch_bounds__(512) void lstm_fwd_big(BigFwdArgs a) {
    constexpr bool BF3 = PR != 0;
    constexpr int H = 1024, UW = 16, NT = 4, NKBX = H / 16, KBW = 8;        // KBW: 16-row K blocks per wave (8 waves x 8 = 64)
    __shared__ __attribute__((aligned(16))) float part[8][NT][256];          // K-split partial sums
    __shared__ unsigned s_ticket;
    const int T = a.T, B = a.B, l = a.layer;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nmt = (B + 15) / 16;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xF;
    if (threadIdx.x == 0) s_ticket = atomicAdd(a.tickets + xcc, 1u);
    __syncthreads();
    const int mb = (int)(xcc >> 1), ub = (int)((xcc & 1u) * 32u + s_ticket);
    if (mb >= nmt || s_ticket >= 32u) return;
    const unsigned long long t_begin = wall_clock64();

    // ---- this wave's W_hh fragments (forward pack, UW = 16: K blocks NKBX.. are the h rows) -> registers, once
    float4 wv[KBW][NT];
    {
        const float* wp = a.wp + ((size_t)(l * (H / UW) + ub) * (2 * NKBX)) * (NT * 256) + lane * 4;
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                wv[kb][j] = *reinterpret_cast<const float4*>(wp + (size_t)((NKBX + wave * KBW + kb) * NT + j) * 256);
    }
    u32x4_f whi[BF3 ? KBW / 2 : 1][NT], wlo[BF3 ? KBW / 2 : 1][NT];      // split precision: bf16 hi / lo pairs (same register count)
    if (BF3) {
#pragma unroll
        for (int jb = 0; jb < KBW / 2; ++jb)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float x[8] = {wv[2 * jb][j].x, wv[2 * jb][j].y, wv[2 * jb][j].z, wv[2 * jb][j].w,
                                    wv[2 * jb + 1][j].x, wv[2 * jb + 1][j].y, wv[2 * jb + 1][j].z, wv[2 * jb + 1][j].w};
                flow_bf3_split(x, whi[jb][j], wlo[jb][j]);
            }
    }
    // ---- epilogue identity of threads 0..255: one (batch row, unit) pair for the whole sequence
    const int pbl = (threadIdx.x & 255) >> 4, pu = threadIdx.x & 15;
    const int pb = mb * 16 + pbl, punit = ub * UW + pu;
    const bool epi = wave < 4;
    const bool pok = pb < B;
    const int pbc = min(pb, B - 1);
    const int e_len = a.lengths[pbc];
    const size_t e = (size_t)pbc * H + punit;
    float c_prev = a.cs[((size_t)l * (T + 1)) * B * H + e];
    float h_prev = a.hs[((size_t)l * (T + 1)) * B * H + e];
    const int ee = ((pbl >> 2) * 16 + pu) * 4 + (pbl & 3);      // this element inside a 16x16 accumulator tile
    const size_t po = packed_off(pb, punit, H);                  // ... and inside a packed [rows, H] panel

    const size_t slot_floats = (size_t)nmt * 16 * H;
    const auto rh = __builtin_amdgcn_make_buffer_rsrc(a.hring, 0, (unsigned)(2 * slot_floats * 4), 0x00020000);
    const unsigned lane_off = (unsigned)((((size_t)mb * NKBX + wave * KBW) * 256 + lane * 4) * 4);
    bool dead = false;
    u32x4_f av[KBW];
    // (Round 4, measured and removed: a wave's K slice is the h tiles of eight unit blocks, and unit blocks 0-31 / 32-63 are produced on
    // the first / second XCD of the pair, so half of a workgroup's waves read tiles written on THEIR XCD.  Reading those through the
    // XCD's L2 at once -- non-temporal loads of the write-through tiles: 7.15 instead of 5.87 us per step, the L2's copy follows late
    // and the early polls only add retry rounds; or from a second, plainly stored copy of the ring: 5.87 us, no gain -- the near wave's
    // MFMAs do start earlier, but the step still ends with the far wave's, which start when the far tiles arrive either way.)
    auto issue = [&](int slot) {
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb)
            av[kb] = __builtin_amdgcn_raw_buffer_load_b128(rh, lane_off + (unsigned)(kb * 1024), (unsigned)((size_t)slot * slot_floats * 4), 16);   // sc1
    };
    auto settle = [&](int slot, unsigned par) {      // (first check straight-line, the retry loop behind it: see lstm_fwd_flow2)
        bool again = false;
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) again = again || flow_untagged(av[kb], par);
        if (__any(again) && !dead) {
            while (true) {
                if (wall_clock64() - t_begin > a.limit) { dead = true; if (lane == 0) atomicOr(a.err, 1u); break; }
                issue(slot);
                again = false;
#pragma unroll
                for (int kb = 0; kb < KBW; ++kb) again = again || flow_untagged(av[kb], par);
                if (!__any(again)) break;
            }
        }
    };
    auto fsig = [](float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); };
    auto ftanh = [](float x) {
        const float x2 = x * x;
        const float small = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f - 0.053968254f * x2)));
        const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
        return fabsf(x) < 0.25f ? small : big;
    };
#if BIG_WEIGHTS_RESIDENT
    FLOW_WEIGHTS_RESIDENT();      // BIGRES
#endif
    for (int t = 0; t < T; ++t) {
        // the hoisted row of this step (x.W_ih + b), needed after the MFMAs
        float* gr = a.gates + ((size_t)l * T + t) * B * 4 * H + (size_t)pbc * 4 * H + punit;
        float xg[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) xg[g] = gr[g * H];
        // h_{t-1}: slot t & 1, use count t >> 1 (slot 0 starts with the tagged initial state, slot 1 zeroed).  Every poll is a
        // round trip to memory (~2 us): the first one goes out BIG_POLL_DELAY x 64 clocks after the step's last barrier, when
        // the tiles the other workgroups stored a moment ago have had time to get there
        if (t > 0) {
#pragma unroll 1
            for (int i = 0; i < BIG_POLL_DELAY; ++i) __builtin_amdgcn_s_sleep(1);
        }
        issue(t & 1);
        settle(t & 1, ((unsigned)(t >> 1) & 1u) ^ 1u);
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (BF3) {
#pragma unroll
            for (int jb = 0; jb < KBW / 2; ++jb) {
                // (the ring words carry the slot's parity in their last mantissa bit: 1 ulp, far below the bf16 split's own error)
                const float x[8] = {__uint_as_float(av[2 * jb][0]), __uint_as_float(av[2 * jb][1]), __uint_as_float(av[2 * jb][2]),
                                    __uint_as_float(av[2 * jb][3]), __uint_as_float(av[2 * jb + 1][0]), __uint_as_float(av[2 * jb + 1][1]),
                                    __uint_as_float(av[2 * jb + 1][2]), __uint_as_float(av[2 * jb + 1][3])};
                u32x4_f ah, al;
                flow_bf3_split(x, ah, al);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = flow_bf_mma<PR>(acc[j], ah, al, whi[jb][j], wlo[jb][j]);
            }
        } else {
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av[kb][0]), wv[kb][j].x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av[kb][1]), wv[kb][j].y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av[kb][2]), wv[kb][j].z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av[kb][3]), wv[kb][j].w, acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(&part[wave][j][lane * 4]) = acc[j];
        lds_barrier();
        if (epi) {
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float sacc = xg[g];
#pragma unroll
                for (int w = 0; w < 8; ++w) sacc += part[w][g][ee];
                pre[g] = sacc;
            }
            const float gi = fsig(pre[0]);
            const float gj = ftanh(pre[1]);
            const float gf = fsig(pre[2] + 1.0f);        // forget_bias = 1.0, added at run time
            const float go = fsig(pre[3]);
            const float cn = c_prev * gf + gi * gj;
            const float hn = ftanh(cn) * go;
            const bool live = pok && t < e_len;
            const float hv = live ? hn : (pok ? h_prev : 0.0f);        // (padding rows carry zeros)
            const float cv = live ? cn : c_prev;
            const float zv = live ? hn * zmult(a.drop, l + 1, (uint32_t)((size_t)t * B * H + e)) : 0.0f;
            // the loop-carried hand-off first: this element of h_t, tagged, write-through (the group spans two XCDs)
            const unsigned par = ((unsigned)((t + 1) >> 1) & 1u) ^ 1u;
            __hip_atomic_store(reinterpret_cast<unsigned*>(a.hring) + (size_t)((t + 1) & 1) * slot_floats + po,
                               (__float_as_uint(hv) & ~1u) | par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pok) {
                gr[0] = gi; gr[H] = gj; gr[2 * H] = gf; gr[3 * H] = go;
                a.cs[((size_t)l * (T + 1) + t + 1) * B * H + e] = cv;
                a.hs[((size_t)l * (T + 1) + t + 1) * B * H + e] = hv;
                a.z[((size_t)(l + 1) * T + t) * B * H + e] = zv;
            }
            c_prev = cv; h_prev = hv;
        }
        lds_barrier();                                    // part[] is free again
    }
}


// ------------------------------------------------- forward, H = 1024 in plain bf16: a batch tile's group on ONE XCD (round 5)
// As bf16 the h half of a 1024-wide layer's kernel is 8 MB: it fits the registers of 32 CUs.  A batch tile's group is then the 32
// workgroups of ONE XCD (lstm_fwd_big: the 64 of an XCD pair):
//   * workgroup ub owns 32 units x 4 gates = eight 16-column N tiles (the packs of the two 16-unit blocks 2 ub, 2 ub + 1), a wave a K
//     slice of 128 rows of W_hh as bf16 fragments of v_mfma_f32_16x16x32_bf16 -- 4 K pairs x 8 tiles x 4 VGPRs = the same 128
//     registers; 32 MFMAs per wave and step;
//   * the loop-carried panel h_{t-1} never leaves the XCD: plain stores, non-temporal loads served by its L2 (0.95 us per hand-off
//     against 2 - 3 us through memory for the XCD pair), the same parity tags, the same f32 words (split in registers);
//   * all eight waves run the epilogue (512 elements per workgroup and step);
//   * four batch tiles use four XCDs: TWO stacks of the same shape (a bidirectional model's two directions, amdspeech_lstm_fwd_pair)
//     run their layers side by side in ONE launch, stack 0 on XCDs 0 - 3, stack 1 on XCDs 4 - 7.  (Measured first: two launches on
//     two streams, one of them placed on the upper XCDs.  They do not overlap -- a workgroup of the second launch that is dealt to an
//     XCD the first one fills waits for a CU there, and the dispatcher hands out workgroups in order: 109.5 ms per configs[4] step
//     against 108.0 one after the other.)
struct BigFwd1Args {
    BigFwdArgs b[2];
    int n;                         // stacks in this launch: 1, or 2 (stack 1 on the XCDs from 4 up)
};
constexpr int BIG1_LDS_BYTES = 8 * 8 * 256 * 4;
__global__ __launch_bounds__(512) void lstm_fwd_big1(BigFwd1Args a1) {
    constexpr int H = 1024, NKBX = H / 16, KBW = 8, NTL = 8;      // K blocks (16 rows) per wave; N tiles per workgroup
    extern __shared__ __attribute__((aligned(16))) float part1[];      // [8 waves][NTL][256] K-split partial sums (64 KiB)
    __shared__ unsigned s_ticket;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xF;
    const bool second = a1.n == 2 && xcc >= 4u;
    const BigFwdArgs a = second ? a1.b[1] : a1.b[0];
    const int T = a.T, B = a.B, l = a.layer;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nmt = (B + 15) / 16;
    if (threadIdx.x == 0) s_ticket = atomicAdd(a.tickets + xcc, 1u);
    __syncthreads();
    const int mb = (int)xcc - (second ? 4 : 0), ub = __builtin_amdgcn_readfirstlane((int)s_ticket);
    if (mb >= nmt || ub >= 32) return;
    const unsigned long long t_begin = wall_clock64();

    // ---- this wave's W_hh fragments: K pairs (16-row blocks 2 jb, 2 jb + 1 of its slice) x the eight N tiles, as bf16 (hi only)
    u32x4_f whi[KBW / 2][NTL];
    {
#pragma unroll
        for (int n = 0; n < NTL; ++n) {
            const float* wp = a.wp + ((size_t)(l * (H / 16) + 2 * ub + (n >> 2)) * (2 * NKBX)) * (4 * 256) + lane * 4;
#pragma unroll
            for (int jb = 0; jb < KBW / 2; ++jb) {
                const float4 w0 = *reinterpret_cast<const float4*>(wp + (size_t)((NKBX + wave * KBW + 2 * jb) * 4 + (n & 3)) * 256);
                const float4 w1 = *reinterpret_cast<const float4*>(wp + (size_t)((NKBX + wave * KBW + 2 * jb + 1) * 4 + (n & 3)) * 256);
                const float x[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                u32x4_f lo_unused;
                flow_bf3_split(x, whi[jb][n], lo_unused);
            }
        }
    }
    // ---- epilogue identity: one (batch row, unit) pair per thread for the whole sequence
    const int hb = threadIdx.x >> 8;                               // which of the workgroup's two 16-unit blocks
    const int pbl = (threadIdx.x & 255) >> 4, pu = threadIdx.x & 15;
    const int pb = mb * 16 + pbl, punit = ub * 32 + hb * 16 + pu;
    const bool pok = pb < B;
    const int pbc = min(pb, B - 1);
    const int e_len = a.lengths[pbc];
    const size_t e = (size_t)pbc * H + punit;
    float c_prev = a.cs[((size_t)l * (T + 1)) * B * H + e];
    float h_prev = a.hs[((size_t)l * (T + 1)) * B * H + e];
    const int ee = ((pbl >> 2) * 16 + pu) * 4 + (pbl & 3);      // this element inside a 16x16 accumulator tile
    const size_t po = packed_off(pb, punit, H);                  // ... and inside a packed [rows, H] panel

    const size_t slot_floats = (size_t)nmt * 16 * H;
    const auto rh = __builtin_amdgcn_make_buffer_rsrc(a.hring, 0, (unsigned)(2 * slot_floats * 4), 0x00020000);
    const unsigned lane_off = (unsigned)((((size_t)mb * NKBX + wave * KBW) * 256 + lane * 4) * 4);
    bool dead = false;
    u32x4_f av[KBW];
    auto issue = [&](int slot) {
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb)
            av[kb] = __builtin_amdgcn_raw_buffer_load_b128(rh, lane_off + (unsigned)(kb * 1024), (unsigned)((size_t)slot * slot_floats * 4), 2);   // nt: this XCD's L2
    };
    auto settle = [&](int slot, unsigned par) {
        bool again = false;
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) again = again || flow_untagged(av[kb], par);
        if (__any(again) && !dead) {
            while (true) {
                if (wall_clock64() - t_begin > a.limit) { dead = true; if (lane == 0) atomicOr(a.err, 1u); break; }
                issue(slot);
                again = false;
#pragma unroll
                for (int kb = 0; kb < KBW; ++kb) again = again || flow_untagged(av[kb], par);
                if (!__any(again)) break;
            }
        }
    };
    auto fsig = [](float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); };
    auto ftanh = [](float x) {
        const float x2 = x * x;
        const float small = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f - 0.053968254f * x2)));
        const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
        return fabsf(x) < 0.25f ? small : big;
    };
    FLOW_WEIGHTS_RESIDENT();
    for (int t = 0; t < T; ++t) {
        // the hoisted row of this step (x.W_ih + b), needed after the MFMAs
        float* gr = a.gates + ((size_t)l * T + t) * B * 4 * H + (size_t)pbc * 4 * H + punit;
        float xg[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) xg[g] = gr[g * H];
        if (t > 0) {
#pragma unroll 1
            for (int i = 0; i < FLOW_POLL_DELAY; ++i) __builtin_amdgcn_s_sleep(1);
        }
        issue(t & 1);
        settle(t & 1, ((unsigned)(t >> 1) & 1u) ^ 1u);
        f32x4 acc[NTL];
#pragma unroll
        for (int n = 0; n < NTL; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jb = 0; jb < KBW / 2; ++jb) {
            const float x[8] = {__uint_as_float(av[2 * jb][0]), __uint_as_float(av[2 * jb][1]), __uint_as_float(av[2 * jb][2]),
                                __uint_as_float(av[2 * jb][3]), __uint_as_float(av[2 * jb + 1][0]), __uint_as_float(av[2 * jb + 1][1]),
                                __uint_as_float(av[2 * jb + 1][2]), __uint_as_float(av[2 * jb + 1][3])};
            u32x4_f ah, al;
            flow_bf3_split(x, ah, al);
#pragma unroll
            for (int n = 0; n < NTL; ++n) acc[n] = flow_bf_mma<2>(acc[n], ah, al, whi[jb][n], whi[jb][n]);
        }
#pragma unroll
        for (int n = 0; n < NTL; ++n) *reinterpret_cast<f32x4*>(part1 + ((size_t)(wave * NTL + n) * 256 + lane * 4)) = acc[n];
        lds_barrier();
        {
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float sacc = xg[g];
#pragma unroll
                for (int w = 0; w < 8; ++w) sacc += part1[(size_t)(w * NTL + hb * 4 + g) * 256 + ee];
                pre[g] = sacc;
            }
            const float gi = fsig(pre[0]);
            const float gj = ftanh(pre[1]);
            const float gf = fsig(pre[2] + 1.0f);        // forget_bias = 1.0, added at run time
            const float go = fsig(pre[3]);
            const float cn = c_prev * gf + gi * gj;
            const float hn = ftanh(cn) * go;
            const bool live = pok && t < e_len;
            const float hv = live ? hn : (pok ? h_prev : 0.0f);        // (padding rows carry zeros)
            const float cv = live ? cn : c_prev;
            const float zv = live ? hn * zmult(a.drop, l + 1, (uint32_t)((size_t)t * B * H + e)) : 0.0f;
            // the loop-carried hand-off first: this element of h_t, tagged; it only has to reach this XCD's L2
            const unsigned par = ((unsigned)((t + 1) >> 1) & 1u) ^ 1u;
            __hip_atomic_store(reinterpret_cast<unsigned*>(a.hring) + (size_t)((t + 1) & 1) * slot_floats + po,
                               (__float_as_uint(hv) & ~1u) | par, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (pok) {
                gr[0] = gi; gr[H] = gj; gr[2 * H] = gf; gr[3 * H] = go;
                a.cs[((size_t)l * (T + 1) + t + 1) * B * H + e] = cv;
                a.hs[((size_t)l * (T + 1) + t + 1) * B * H + e] = hv;
                a.z[((size_t)(l + 1) * T + t) * B * H + e] = zv;
            }
            c_prev = cv; h_prev = hv;
        }
        lds_barrier();                                    // part1[] is free again
    }
}
