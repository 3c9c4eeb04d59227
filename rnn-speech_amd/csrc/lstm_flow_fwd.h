// Part of lstm.hip -- lstm_fwd_flow2 (H <= 512: the whole sequence in one launch) and its x-product workers, DESIGN.md 4.2.
// Not a standalone header: lstm.hip includes its kernel families in a fixed order, inside namespace amdspeech, after the helpers
// (layout, dropout multipliers, packs) they use.  Tuning macros (#ifndef ...) keep their defaults here; rnn-speech_amd/build.py
// passes overrides for development builds (AMDSPEECH_CXXFLAGS).

// ------------------------------------------------- forward dataflow kernel, lockstep form
// (Round 1's lstm_fwd_flow, removed in round 4, specialised its waves -- four ran the x half of step t+1 while four waited for h_t
// and ran the h half; the x waves' MFMA burst sat on the same SIMDs as the h waves' polls and held them back ~0.6 us per step,
// DESIGN.md 4.2.)  Here all eight waves run the SAME phase, like lstm_bwd_flow2: every wave owns a K slice (H/128 blocks of 16 rows) of BOTH halves,
//   [settle h_{t-1}] [h MFMAs into the accumulators that already hold the x half] [partials -> LDS] B1
//   [waves 0-3: epilogue(t), h tile out | waves 4-7: the stores of step t-1, x prefetch] B2
//   [x MFMAs of step t+1 into fresh accumulators; the loads of h_t go out part-way through them] ...
// so the hand-off of h_t travels under the x MFMAs, the x half never leaves the registers, and one LDS reduction per step is left.
#ifndef FWD2_GATHER_AT
#define FWD2_GATHER_AT 1          // the loads of h_t are issued after this many of the K blocks of the x half (clamped to the last one)
#endif
#ifndef FWD2_WORKER_LAG
#define FWD2_WORKER_LAG 4         // x-product workers above the bottom layer: frames they stay behind the layer below (see fwd_x_worker)
#endif

// ---- x-product workers of lstm_fwd_flow2 (round 5) ---------------------------------------------------------------------------
// The x half of a layer's product, x_t . W_ih, is not loop-carried: x_t is the (masked) output of the layer below, complete
// long before this layer needs it.  cfg2 places its six recurrence groups on six XCDs; the waves of the other two take MV of the
// KB K blocks every recurrence wave owns of the x half (K rows, ALL 64 gate columns of the workgroup) and hand the group a
// pre-multiplied [16 rows x 64 gate columns] tile per workgroup and frame.  One worker WAVE = one role (layer, batch tile, unit
// block, part): the eight K blocks {w*KB + KB-1-part : w = 0..7} x 4 N tiles of W_ih stay in its registers for the whole
// sequence (128 VGPRs), the operand is the SAME fragment-major panel the recurrence waves read (xp0 for the bottom layer, the
// sentinel-polled xph[l][t] above it: 8 KiB per frame), 128 MFMAs per frame, no LDS, no barrier.  The result goes out
// write-through as four 1 KiB stores in accumulator order, every word tagged with the LAUNCH's parity in its least significant
// mantissa bit (the panel is written exactly once per launch: the previous launch left the other parity, nothing is re-filled);
// the epilogue threads of the recurrence group fetch their 16 bytes two steps ahead and add them to the bias in front of B1.
// Nothing throttles a worker but its operand: the layer above then trails the layer below by the few frames the hand-off takes.
template <int KB, int MV>
__device__ __forceinline__ void fwd_x_worker(const FlowArgs& a, const int role, const int lane, const unsigned long long t_begin) {
    constexpr int H = 128 * KB, NKBX = H / 16, NU = H / 16, NT = 4;
    const int T = a.T, nmt = (a.B + 15) / 16;
    int r = __builtin_amdgcn_readfirstlane(role);
    const int part = r % MV; r /= MV;
    const int ub = r % NU; r /= NU;
    const int mb = r % nmt;
    const int l = r / nmt;
    if (l >= a.L) return;
    const size_t bph = (size_t)nmt * 16 * H;
    float4 w[8][NT];
    {
        const float* wp = a.wp + ((size_t)(l * NU + ub) * (2 * NKBX)) * (NT * 256) + lane * 4;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv)
#pragma unroll
            for (int j = 0; j < NT; ++j) w[wv][j] = *reinterpret_cast<const float4*>(wp + (size_t)((wv * KB + KB - 1 - part) * NT + j) * 256);
    }
    const float* xsrc = l == 0 ? a.xp0 : a.xph + (size_t)l * T * bph;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xsrc), 0, (unsigned)((size_t)T * bph * 4), 0x00020000);
    const size_t fs = (size_t)a.L * nmt * NU * MV * 1024;                              // floats per frame of xwp
    const auto ro = __builtin_amdgcn_make_buffer_rsrc(a.xwp, 0, (unsigned)((size_t)T * fs * 4), 0x00020000);
    const unsigned lane_off = (unsigned)((((size_t)mb * NKBX + (KB - 1 - part)) * 256 + lane * 4) * 4);      // + wv*KB KiB: K block of recurrence wave wv
    const unsigned out_off = (unsigned)((((((size_t)l * nmt + mb) * NU + ub) * MV + part) * 1024 + lane * 4) * 4);
    const unsigned par = a.xw_par & 1u;
    bool dead = false;
    u32x4_f xa[8] = {}, xb[8] = {};
    // EVERY load of the frame loop is inline assembly and every wait an explicit s_waitcnt (the pattern of gemm_tile_tn_direct):
    // left to hipcc, the retry paths below turn the waits in front of the MFMAs into vmcnt(0) -- a wait for
    // the probe issued a moment earlier, i.e. a round trip to memory in series with every frame's MFMAs (first version: 5.6 us per
    // step).  The frame loop issues, per frame and in this order: 1 probe, 4 tile stores, 8 panel loads -- always, with clamped frame
    // indices at the end of the sequence -- so "this frame's panel and probe have landed" is vmcnt(12) everywhere.
    // (plain lambdas: clang does not capture a variable that a GENERIC lambda names only in an asm operand)
    auto load_l2 = [&](u32x4_f& dst, unsigned vo, unsigned so) { asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(dst) : "v"(vo), "s"(rx), "s"(so)); };
    auto load_mem = [&](u32x4_f& dst, unsigned vo, unsigned so) { asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen sc1" : "+v"(dst) : "v"(vo), "s"(rx), "s"(so)); };
    auto issue = [&](auto bottom, u32x4_f (&buf)[8], int t) {
        const unsigned base = (unsigned)((size_t)(t < T ? t : T - 1) * bph * 4);
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) {       // (bottom layer: xp0 is complete and read through this XCD's L2; above it: sc1, served by memory)
            const unsigned vo = lane_off + (unsigned)(wv * KB * 1024);
            if (decltype(bottom)::value) load_l2(buf[wv], vo, base); else load_mem(buf[wv], vo, base);
        }
    };
    FLOW_WEIGHTS_RESIDENT();
    // A worker above the bottom layer stays FWD2_WORKER_LAG frames behind the layer below ON PURPOSE.  Next to its producer it would
    // find every operand panel missing, and a frame would cost a poll's round trip to memory (~2.5 us) PLUS its MFMAs (1.8 us, 3.6
    // with the partner wave of its SIMD streaming too) -- more than a recurrence step.  Behind a gate -- ONE 16-byte probe of the
    // frame LAG ahead, requested in front of the previous frame's MFMAs -- the panels two frames ahead are always there (the 32
    // workgroups of the group below run in lockstep, a step apart at most) and a frame costs its MFMAs.  The panels are still
    // checked; the layer above trails the layer below by LAG + ~3 frames.
    u32x4_f pr = (u32x4_f){0u, 0u, 0u, 0u};
    auto probe = [&](int t) {
        const int tp = t + FWD2_WORKER_LAG < T ? t + FWD2_WORKER_LAG : T - 1;
        load_mem(pr, lane_off, (unsigned)((size_t)tp * bph * 4));
    };
    // at most 12 / 0 younger operations may still be in flight: the probe and the panel have landed
    auto landed12 = [&](u32x4_f (&buf)[8]) {
        asm volatile("s_waitcnt vmcnt(12)" : "+v"(pr), "+v"(buf[0]), "+v"(buf[1]), "+v"(buf[2]), "+v"(buf[3]), "+v"(buf[4]), "+v"(buf[5]), "+v"(buf[6]), "+v"(buf[7]));
    };
    auto landed0 = [&](u32x4_f (&buf)[8]) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pr), "+v"(buf[0]), "+v"(buf[1]), "+v"(buf[2]), "+v"(buf[3]), "+v"(buf[4]), "+v"(buf[5]), "+v"(buf[6]), "+v"(buf[7]));
    };
    auto pending_any = [&](const u32x4_f (&buf)[8]) -> bool {      // branch-free (a chain of || became eight saveexec branches)
        unsigned bad = 0u;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) bad |= (unsigned)flow_pending(buf[wv]);
        return bad != 0u;
    };
#if defined(AMDSPEECH_DEVTRACE) && AMDSPEECH_DEVTRACE == 5      // tools/trace_fwd2.py: the worker of layer 1 (if any), batch tile 0, unit block 3, part 0
    const bool wtracing = a.trace != nullptr && l == a.trace_layer && ub == 3 && mb == 0 && part == 0 && lane == 0;
#define FXWSTAMP(i) do { if (wtracing && t >= 500 && t < 508) a.trace[128 + (t - 500) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define FXWSTAMP(i) do { } while (0)
#endif
    auto work = [&](auto bottom, int t, u32x4_f (&buf)[8]) __attribute__((always_inline)) {
        FXWSTAMP(0);
        landed12(buf);
        FXWSTAMP(1);
        if (!decltype(bottom)::value) {
            if (__any(flow_pending(pr) || pending_any(buf)) && !dead) {      // the gate is shut, or (never seen) a panel behind it is missing
                while (true) {
                    if (wall_clock64() - t_begin > a.limit) { dead = true; if (lane == 0) atomicOr(a.err, 8u); break; }
                    __builtin_amdgcn_s_sleep(4);
                    probe(t);
                    issue(bottom, buf, t);
                    landed0(buf);
                    if (!__any(flow_pending(pr) || pending_any(buf))) break;
                }
            }
        }
        FXWSTAMP(2);
        probe(t + 1);            // (the bottom layer's workers too: one order of operations, one wait count)
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int wv = 0; wv < 8; ++wv)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(buf[wv][0]), w[wv][j].x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(buf[wv][1]), w[wv][j].y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(buf[wv][2]), w[wv][j].z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(buf[wv][3]), w[wv][j].w, acc[j], 0, 0, 0);
            }
        // element lane*4 + i of the 16x16 tile: its four gates as one 16-byte word at slot i*64 + lane (a 1 KiB run per store)
        // (the frame offset in voffset, not in an SGPR soffset: the gfx950 store hazard noted at lstm_bwd_flow2's store_tiles)
        const unsigned fo = out_off + (unsigned)((size_t)t * fs * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_buffer_store_b128(flow_tag((f32x4){acc[0][i], acc[1][i], acc[2][i], acc[3][i]}, par), ro,
                                                   fo + (unsigned)(i * 1024), 0, 16);      // sc1: through to memory
        __builtin_amdgcn_sched_barrier(0);
        FXWSTAMP(3);
        issue(bottom, buf, t + 2);      // (two register sets: the operand panels are requested two frames ahead; past the end: the last frame again)
        FXWSTAMP(4);
    };
#undef FXWSTAMP
    auto run = [&](auto bottom) __attribute__((always_inline)) {
        probe(0);
        if (!decltype(bottom)::value) {                // the first panels are requested once the gate of frame 0 is open
            landed0(xa);
            if (__any(flow_pending(pr)) && !dead) {
                while (true) {
                    if (wall_clock64() - t_begin > a.limit) { dead = true; if (lane == 0) atomicOr(a.err, 8u); break; }
                    __builtin_amdgcn_s_sleep(4);
                    probe(0);
                    landed0(xa);
                    if (!__any(flow_pending(pr))) break;
                }
            }
        }
        issue(bottom, xa, 0);
        landed0(xa);
        // (frame 0: its probe and panel have landed, only the second panel is in flight; from frame 1 on the order above holds)
        issue(bottom, xb, 1);
        for (int t = 0; t < T; t += 2) {
            work(bottom, t, xa);
            if (t + 1 < T) work(bottom, t + 1, xb);
        }
        landed0(xa);
        landed0(xb);
    };
    if (l == 0) run(std::true_type{}); else run(std::false_type{});
}

#define FLOW_G(T, p) ((T __attribute__((address_space(1)))*)(p))      // a pointer into global memory, said so (see lstm_bwd_flow2)
template <typename Args>
__device__ __forceinline__ Args flow_args_again() {      // (scalar loads: 16-byte pieces through a pointer in the constant address space)
    typedef unsigned args_u4 __attribute__((ext_vector_type(4)));
    typedef const args_u4 __attribute__((address_space(4))) * cptr;
    static_assert(sizeof(Args) % 16 == 0, "argument struct: a whole number of 16-byte pieces");
    unsigned long long p = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    asm("" : "+s"(p));      // (not volatile: nothing but the value must be opaque)
    const cptr q = (cptr)p;
    args_u4 buf[sizeof(Args) / 16];
#pragma unroll
    for (unsigned i = 0; i < sizeof(Args) / 16; ++i) buf[i] = q[i];
    Args r;
    __builtin_memcpy(&r, buf, sizeof(Args));
    return r;
}
template <int H>
__device__ __forceinline__ void ctc_follower_call(const CtcFlow& c, int wg, int nwg) {
    extern __shared__ __attribute__((aligned(16))) float cf_lds[];
    if (c.B <= nwg * 2) ctc_follower<H, 1>(c, cf_lds, wg, nwg);
    else ctc_follower<H, 2>(c, cf_lds, wg, nwg);
}

template <int KB, int PR, int MV, bool CF = false>   // KB: 16-row K blocks per wave and half (H / 128); PR: 0 exact f32, 1 bf16x3, 2 bf16 products (KB even);
                                    // MV: K blocks per wave of the x half that the x-product workers of the spare XCDs form (0: none)
                                    // CF: the instantiation that carries the fused CTC head's follower (ctc_flow.h).  A separate one: the
                                    // role's scalar-register pressure costs the recurrence loops of the SAME function lane moves per
                                    // step (register allocation is per function), which launches without a head must not pay
__global__ __launch_bounds__(512) void lstm_fwd_flow2(FlowArgs a_in) {
    constexpr bool BF3 = PR != 0;
    static_assert(MV >= 0 && MV < KB && (MV == 0 || PR == 0), "x-product workers: exact f32 only, and one K block of the x half stays");
    constexpr int KX = KB - MV;       // K blocks of the x half this wave multiplies itself
    constexpr int MVA = MV > 0 ? MV : 1;
    constexpr int UW = 16, NT = 4, H = 128 * KB, NKBX = H / 16, NW = 8;
    __shared__ __attribute__((aligned(16))) float red_[1][NW][256][NT];   // K-split partial sums (x + h halves together), the four gates of an element adjacent
    __shared__ __attribute__((aligned(16))) float outbox[2][8][256];         // epilogue results on their way to the stores
    __shared__ unsigned s_ticket;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xF;
    if (threadIdx.x == 0) s_ticket = atomicAdd(a_in.tickets + xcc, 1u);
    __syncthreads();
    const int grp = (int)xcc, ub = __builtin_amdgcn_readfirstlane((int)s_ticket);      // (both wave-uniform, said so: the role dispatch below is then made of real branches)
    if (grp >= a_in.L * ((a_in.B + 15) / 16)) {             // an XCD without a recurrence group
        const int first = a_in.L * ((a_in.B + 15) / 16);
        if (MV > 0 && ub < a_in.w_wpx && wave < a_in.w_wpw)
            fwd_x_worker<KB, MVA>(a_in, (((grp - first) * a_in.w_wpx + ub) * a_in.w_wpw + wave), lane, wall_clock64());
        else if (CF && a_in.cf_on && ub >= a_in.w_wpx && ub < a_in.w_wpx + a_in.cf_nfw) {
            // the CTC head's forward half (ctc_flow.h): output Linear + log-softmax + alpha, 16 frames behind the top layer
            if constexpr (CF) ctc_follower_call<H>(a_in.cf, (grp - first) * a_in.cf_nfw + (ub - a_in.w_wpx), (8 - first) * a_in.cf_nfw);
        }
        return;
    }
    // (CF: the recurrence's own copy of the arguments, loaded behind the role dispatch: see lstm_bwd_flow2)
    const FlowArgs a = CF ? flow_args_again<FlowArgs>() : a_in;
    const int T = a.T, B = a.B;
    const int nmt = (B + 15) / 16;
    if (ub >= H / UW) return;                               // spare workgroups of a narrow layer
    const int l = grp / nmt, mb = grp % nmt;
    const size_t bph = (size_t)nmt * 16 * H;
    const unsigned long long t_begin = wall_clock64();
    const unsigned long long c_begin = __builtin_readcyclecounter();

    // ---- this wave's weight fragments: K blocks wave*KB .. +KB of the x rows and of the h rows -> registers, once
    // (with x-product workers: only the first KX of the wave's KB x blocks -- the workers hold the others)
    float4 wx[KX][NT], wh[KB][NT];
    {
        const float __attribute__((address_space(1)))* wp = FLOW_G(const float, a.wp) + ((size_t)(l * (H / UW) + ub) * (2 * NKBX)) * (NT * 256) + lane * 4;      // (FLOW_G: see lstm_bwd_flow2)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (kb < KX) { const f32x4 v = *(const f32x4 __attribute__((address_space(1)))*)(wp + (size_t)((wave * KB + kb) * NT + j) * 256); wx[kb][j] = make_float4(v[0], v[1], v[2], v[3]); }
                { const f32x4 v = *(const f32x4 __attribute__((address_space(1)))*)(wp + (size_t)((NKBX + wave * KB + kb) * NT + j) * 256); wh[kb][j] = make_float4(v[0], v[1], v[2], v[3]); }
            }
    }
    // split-precision mode: the weight fragments as bf16 hi / lo pairs (same register count), built once
    constexpr int KP = BF3 ? KB / 2 : 1;
    u32x4_f wxh[KP][NT], wxl[KP][NT], whh[KP][NT], whl[KP][NT];
    if constexpr (BF3) {
#pragma unroll
        for (int jb = 0; jb < KB / 2; ++jb)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float xx[8] = {wx[2 * jb][j].x, wx[2 * jb][j].y, wx[2 * jb][j].z, wx[2 * jb][j].w,
                                     wx[2 * jb + 1][j].x, wx[2 * jb + 1][j].y, wx[2 * jb + 1][j].z, wx[2 * jb + 1][j].w};
                flow_bf3_split(xx, wxh[jb][j], wxl[jb][j]);
                const float xh[8] = {wh[2 * jb][j].x, wh[2 * jb][j].y, wh[2 * jb][j].z, wh[2 * jb][j].w,
                                     wh[2 * jb + 1][j].x, wh[2 * jb + 1][j].y, wh[2 * jb + 1][j].z, wh[2 * jb + 1][j].w};
                flow_bf3_split(xh, whh[jb][j], whl[jb][j]);
            }
    }
    // ---- epilogue identity of threads 0..255: one (batch row, unit) pair for the whole sequence
    const int pbl = (threadIdx.x & 255) >> 4, pu = threadIdx.x & 15;
    const int pb = mb * 16 + pbl, punit = ub * UW + pu;
    const bool epi = threadIdx.x < 256;
    const bool pok = pb < B;
    const int pbc = min(pb, B - 1);
    const float __attribute__((address_space(1)))* bias = FLOW_G(const float, a.bias) + l * a.bias_stride;
    float e_bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) e_bias[g] = bias[g * H + punit];
    const int e_len = FLOW_G(const int, a.lengths)[pbc];
    const size_t e = (size_t)pbc * H + punit;
    float c_prev = FLOW_G(float, a.cs)[((size_t)l * (T + 1)) * B * H + e];
    float h_prev = FLOW_G(float, a.hs)[((size_t)l * (T + 1)) * B * H + e];
    const size_t po = packed_off(pb, punit, H);
    const int ee = ((pbl >> 2) * 16 + pu) * 4 + (pbl & 3);     // this element inside a 16x16 accumulator tile

    const float* xsrc = l == 0 ? a.xp0 : a.xph + (size_t)l * T * bph;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xsrc), 0, (unsigned)((size_t)T * bph * 4), 0x00020000);
    const auto rh = __builtin_amdgcn_make_buffer_rsrc(a.hph + (size_t)l * (T + 1) * bph, 0, (unsigned)((size_t)(T + 1) * bph * 4), 0x00020000);
    const unsigned lane_off = (unsigned)((((size_t)mb * NKBX + wave * KB) * 256 + lane * 4) * 4);
    bool dead = false;
    using Local = std::integral_constant<int, 2>;       // nt: served by this XCD's L2
    using Remote = std::integral_constant<int, 16>;     // sc1: served by memory
    u32x4_f hv[KB] = {}, xa[KX] = {}, xb[KX] = {};      // h_{t-1}; x[s] for even s (xa) and odd s (xb), fetched two steps ahead
    // Round 5: the loads of the time loop are INLINE ASSEMBLY and its waits explicit (FWD2_ASM_LOADS; the pattern of
    // gemm_tile_tn_direct).  gfx9 retires loads in order on one counter and hipcc counts exactly only through straight-line code:
    // with the retry loops of the polled operands in the loop, rounds 2 - 4 waited for h_t with a vmcnt(3..0) ladder -- i.e. also
    // for the x panel (and now the workers' tiles) requested from MEMORY right behind the gather -- and kept a second ladder inside
    // the h MFMA stream (the h phase ran 2.16 us where the x phase ran 1.76).  A step now issues, in this order and unconditionally
    // (clamped frame indices at the end of the sequence): the KB loads of h_t part-way through the x half, the KX loads of the x
    // panel three steps ahead, the MV loads of the workers' tiles two steps ahead; the ONE wait of the step is vmcnt(KX + MV) at its
    // top -- h_t has landed, whatever was requested behind it is still in flight.  Everything else the step reads was requested
    // before h_t.  A retry (sentinel / old tag seen) re-requests and waits for vmcnt(0): fewer operations in flight than the count
    // assumes is always safe.  (Plain lambdas: clang does not capture a variable a generic lambda names only in an asm operand.)
    // (H = 512 WITHOUT workers -- AMDSPEECH_FLOW_FWD_WORKERS=0, the split precisions, no spare XCD -- keeps the loop of rounds 2 - 4:
    //  with a fourth x block per wave in registers the pinned buffers do not fit 256 VGPRs, six spills)
#ifndef FWD2_ASM_LOADS
#define FWD2_ASM_LOADS 1
#endif
    constexpr bool ASM = FWD2_ASM_LOADS != 0 && (KB < 4 || MV > 0);
    auto ld_l2 = [&](u32x4_f& dst, decltype(rx) rsrc, unsigned vo, unsigned so) __attribute__((always_inline)) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen nt" : "+v"(dst) : "v"(vo), "s"(rsrc), "s"(so));
    };
    auto ld_mem = [&](u32x4_f& dst, decltype(rx) rsrc, unsigned vo, unsigned so) __attribute__((always_inline)) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen sc1" : "+v"(dst) : "v"(vo), "s"(rsrc), "s"(so));
    };
    auto pin = [&](u32x4_f& r) __attribute__((always_inline)) { asm volatile("" : "+v"(r)); };      // orders the uses of r behind the asm statements in front of it
    auto issue = [&](auto pol, auto& buf, decltype(rx) rsrc, unsigned base) __attribute__((always_inline)) {
        constexpr int NB = (int)(sizeof(buf) / sizeof(buf[0]));
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            if constexpr (ASM) {
                if (decltype(pol)::value == 2) ld_l2(buf[kb], rsrc, lane_off, base + (unsigned)(kb * 1024));
                else ld_mem(buf[kb], rsrc, lane_off, base + (unsigned)(kb * 1024));
            } else {
                buf[kb] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, base + (unsigned)(kb * 1024), decltype(pol)::value);
            }
        }
    };
    auto wait_all = [&](auto& buf) __attribute__((always_inline)) {          // everything this wave has requested has landed
        constexpr int NB = (int)(sizeof(buf) / sizeof(buf[0]));
        if constexpr (ASM) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) pin(buf[kb]);
        }
    };
    // the first check of a polled operand as straight-line code, the retry loop behind it
    auto settle = [&](auto pol, auto& buf, decltype(rx) rsrc, unsigned base) __attribute__((always_inline)) {
        constexpr int NB = (int)(sizeof(buf) / sizeof(buf[0]));
        unsigned again = 0u;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) again |= (unsigned)flow_pending(buf[kb]);
        if (__any(again != 0u) && !dead) {
            while (true) {
                if (wall_clock64() - t_begin > a.limit) { dead = true; if (lane == 0) __hip_atomic_fetch_or(FLOW_G(unsigned, a.err), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                issue(pol, buf, rsrc, base);
                wait_all(buf);
                again = 0u;
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) again |= (unsigned)flow_pending(buf[kb]);
                if (!__any(again != 0u)) break;
            }
        }
    };
    // ---- the x-product workers' tiles: this thread's element (its four gates) of frame t, MV parts, fetched two steps ahead
    // by EVERY wave (waves 4-7 never use theirs: a load in one role only would make hipcc's wait counts inexact at the merge,
    // and the wait for h_t would then cover it -- DESIGN.md 4.2 item 3); tagged with the launch's parity
    const size_t wfs = (size_t)a.L * nmt * (H / UW) * MVA * 1024;                       // floats per frame of xwp
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(a.xwp, 0, MV > 0 ? (unsigned)((size_t)T * wfs * 4) : 0u, 0x00020000);
    const unsigned w_off = (unsigned)((((((size_t)l * nmt + mb) * (H / UW) + ub) * MVA) * 1024 +
                                       ((pbl & 3) * 64 + (pbl >> 2) * 16 + pu) * 4) * 4);      // + part KiB*4
    const unsigned w_par = a.xw_par & 1u;
    u32x4_f wa[MVA] = {}, wb[MVA] = {};            // frames of even (wa) and odd (wb) index
    auto wissue = [&](u32x4_f (&buf)[MVA], int sidx) __attribute__((always_inline)) {
        if (MV > 0) {
#pragma unroll
            for (int p = 0; p < MVA; ++p) {
                if constexpr (ASM) ld_mem(buf[p], rw, w_off + (unsigned)(p * 4096), (unsigned)((size_t)sidx * wfs * 4));
                else buf[p] = __builtin_amdgcn_raw_buffer_load_b128(rw, w_off + (unsigned)(p * 4096), (unsigned)((size_t)sidx * wfs * 4), 16);
            }
        }
    };
    // bias + the workers' share of the x half, checked (first check straight-line, like settle); in front of B1, off the epilogue
    auto wsettle = [&](u32x4_f (&buf)[MVA], int sidx) __attribute__((always_inline)) -> f32x4 {
        f32x4 pre = (f32x4){e_bias[0], e_bias[1], e_bias[2], e_bias[3]};      // gate g of unit pu is column g*16 + pu: N tile g
        if (MV > 0) {
            unsigned again = 0u;
#pragma unroll
            for (int p = 0; p < MVA; ++p) again |= (unsigned)flow_untagged(buf[p], w_par);
            if (__any(again != 0u) && !dead) {
                while (true) {
                    if (wall_clock64() - t_begin > a.limit) { dead = true; if (lane == 0) __hip_atomic_fetch_or(FLOW_G(unsigned, a.err), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    wissue(buf, sidx);
                    wait_all(buf);
                    again = 0u;
#pragma unroll
                    for (int p = 0; p < MVA; ++p) again |= (unsigned)flow_untagged(buf[p], w_par);
                    if (!__any(again != 0u)) break;
                }
            }
#pragma unroll
            for (int p = 0; p < MVA; ++p)
                pre += (f32x4){__uint_as_float(buf[p][0]), __uint_as_float(buf[p][1]), __uint_as_float(buf[p][2]), __uint_as_float(buf[p][3])};
        }
        return pre;
    };
    // (xpol: Local for the bottom layer -- xp0 is complete, read through this XCD's L2 -- Remote above it.  A compile-time tag, and
    //  the whole time loop exists once per tag: a run-time branch around two asm loads of one buffer ends in a phi, i.e. in register
    //  COPIES of loads still in flight)
    auto xissue = [&](auto xpol, u32x4_f (&buf)[KX], int sidx) __attribute__((always_inline)) {
        issue(xpol, buf, rx, (unsigned)((size_t)sidx * bph * 4));
    };
    f32x4 acc[NT];
#ifndef FWD2_EARLY_XCHECK
#define FWD2_EARLY_XCHECK 0        // (1: measured equal or slower)
#endif
#ifndef FWD2_RR_ACC
#define FWD2_RR_ACC 0             // 1: the four accumulators take turns (no MFMA depends on the one in front of it)
#endif
    auto mma_block = [&](const u32x4_f& v, const float4 (&w)[NT]) __attribute__((always_inline)) {
#if FWD2_RR_ACC
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v[0]), w[j].x, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v[1]), w[j].y, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v[2]), w[j].z, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v[3]), w[j].w, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v[0]), w[j].x, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v[1]), w[j].y, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v[2]), w[j].z, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v[3]), w[j].w, acc[j], 0, 0, 0);
        }
#endif
    };
    // one half of the product: the wave's KB blocks of operand `v` against the matching weight fragments
    auto half_product = [&](const auto& v, const auto& w, const u32x4_f (&wh_)[KP][NT], const u32x4_f (&wl_)[KP][NT],
                            auto between) __attribute__((always_inline)) {
        constexpr int NB = (int)(sizeof(v) / sizeof(v[0]));       // KB for the h half, KX for the x half
        if constexpr (BF3) {
#pragma unroll
            for (int jb = 0; jb < KB / 2; ++jb) {
                between(2 * jb);
                between(2 * jb + 1);      // (a pair of K blocks per MFMA group: BOTH indices pass -- at H = 256 the gather point is block 1)
                const float x[8] = {__uint_as_float(v[2 * jb][0]), __uint_as_float(v[2 * jb][1]), __uint_as_float(v[2 * jb][2]),
                                    __uint_as_float(v[2 * jb][3]), __uint_as_float(v[2 * jb + 1][0]), __uint_as_float(v[2 * jb + 1][1]),
                                    __uint_as_float(v[2 * jb + 1][2]), __uint_as_float(v[2 * jb + 1][3])};
                u32x4_f ah, al;
                flow_bf3_split(x, ah, al);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = flow_bf_mma<PR>(acc[j], ah, al, wh_[jb][j], wl_[jb][j]);
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
                between(kb);
                mma_block(v[kb], w[kb]);
            }
        }
    };
    auto fsig = [](float x) __attribute__((always_inline)) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); };
    auto ftanh = [](float x) __attribute__((always_inline)) {
        const float x2 = x * x;
        const float small = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f - 0.053968254f * x2)));
        const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
        return fabsf(x) < 0.25f ? small : big;
    };
    // the epilogue's results of step t (thread tid-256 stores what epilogue thread tid computed): x hand-off to the layer above
    // through memory (write-through), then the BPTT stash (read by later kernels only)
    auto stores = [&](int t) __attribute__((always_inline)) {
        const int sl = threadIdx.x - 256;
        const float (&ob)[8][256] = outbox[t & 1];
        // (the output-dropout multiplier is formed HERE, in the store waves' window: its two hashes -- ~35 integer operations -- sat in
        //  the epilogue, i.e. on the loop-carried path, for a value only the layer above and the backward pass read)
        const float zv = ob[6][sl] * zmult(a.drop, l + 1, (uint32_t)((size_t)t * B * H + e));
        if (l + 1 < a.L || (CF && a.cf_on))      // (the top layer's panels, slot [L]: read by the fused CTC head's follower)
            __hip_atomic_store(FLOW_G(float, a.xph) + ((size_t)(l + 1) * T + t) * bph + po, zv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pb < B) {
            float __attribute__((address_space(1)))* gr = FLOW_G(float, a.gates) + ((size_t)l * T + t) * B * 4 * H + (size_t)pb * 4 * H + punit;
            gr[0] = ob[0][sl]; gr[H] = ob[1][sl]; gr[2 * H] = ob[2][sl]; gr[3 * H] = ob[3][sl];
            FLOW_G(float, a.cs)[((size_t)l * (T + 1) + t + 1) * B * H + e] = ob[4][sl];
            FLOW_G(float, a.hs)[((size_t)l * (T + 1) + t + 1) * B * H + e] = ob[5][sl];
            FLOW_G(float, a.z)[((size_t)(l + 1) * T + t) * B * H + e] = zv;
        }
    };
#if defined(AMDSPEECH_DEVTRACE) && AMDSPEECH_DEVTRACE == 5      // tools/trace_fwd2.py: layer 1 (if any), unit block 3, waves 0 and 5
    const bool tracing = a.trace != nullptr && l == a.trace_layer && ub == 3 && mb == 0 && (wave == 0 || wave == 5) && lane == 0;
#define F2STAMP(i) do { if (tracing && t >= 500 && t < 508) a.trace[((t - 500) * 2 + (wave ? 1 : 0)) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define F2STAMP(i) do { } while (0)
#endif
    // One step; xnext holds x[t+1] (its products end this step), and is refilled with x[t+3].
    // Measured with tools/trace_fwd2.py (5.08 us per step run alone): x phase 2.06 us (the two waves of a SIMD run their 64 MFMAs
    // one after the other, 0.92 us each, + 0.3 us for a layer >= 1 whose prefetch met the sentinel), settle of h_t 0.28, h phase
    // 2.16, B1 0.12, epilogue 0.44, B2 0.04.  Tried and kept out (same box, +-0.05 ms per sequence = no gain or worse): every load
    // unconditional (three x buffers, clamped index, a straight-line first check: exact vmcnt(7..4) waits, but 24 register-pair
    // copies per step), polled operands copied into fresh registers once settled (the vmcnt ladders then guard nothing younger;
    // the waiting just moves into the copies -- VALU does not issue beside the partner's MFMA burst), s_setprio for waves 0-3,
    // round-robin instead of chained accumulators, the four gates of an element adjacent in the LDS reduction (kept: fewer reads).
    // the epilogue of step t (threads 0..255): K-split reduction, gates, state, the h hand-off, results into the outbox
    auto epilogue = [&](int t, const float (&rd)[NW][256][NT], f32x4 pre) __attribute__((always_inline)) {      // pre: bias (+ the workers' tiles)
#pragma unroll
        for (int w = 0; w < NW; ++w) pre += *reinterpret_cast<const f32x4*>(&rd[w][ee][0]);
        const float gi = fsig(pre[0]);
        const float gj = ftanh(pre[1]);
        const float gf = fsig(pre[2] + 1.0f);        // forget_bias = 1.0, added at run time
        const float go = fsig(pre[3]);
        const float cn = c_prev * gf + gi * gj;
        const float hn = ftanh(cn) * go;
        const bool live = pok && t < e_len;
        const float hval = live ? hn : (pok ? h_prev : 0.0f);        // (padding rows carry zeros)
        const float cv = live ? cn : c_prev;
        const float zv = live ? hn : 0.0f;                            // (times its dropout multiplier: see `stores`)
        __hip_atomic_store(FLOW_G(float, a.hph) + ((size_t)l * (T + 1) + t + 1) * bph + po, hval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int sl = threadIdx.x;
        float (&ob)[8][256] = outbox[t & 1];
        ob[0][sl] = gi; ob[1][sl] = gj; ob[2][sl] = gf; ob[3][sl] = go;
        ob[4][sl] = cv; ob[5][sl] = hval; ob[6][sl] = zv; ob[7][sl] = c_prev;
        c_prev = cv; h_prev = hval;
    };
    // the x half of step t+1 into fresh accumulators (gather_h: the loads of h_t go out part-way through it)
    auto x_half = [&](auto xpol, int t, u32x4_f (&xnext)[KX], bool gather_h) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (t + 1 < T) {
            if (decltype(xpol)::value != 2 && !(ASM && FWD2_EARLY_XCHECK)) settle(Remote{}, xnext, rx, (unsigned)((size_t)(t + 1) * bph * 4));
            half_product(xnext, wx, wxh, wxl, [&](int kb) {
                if (gather_h && kb == (FWD2_GATHER_AT < KX ? FWD2_GATHER_AT : KX - 1)) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue(Local{}, hv, rh, (unsigned)((size_t)(t + 1) * bph * 4));
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        }
    };
#ifndef FWD2_ISSUE_AT_TOP
#define FWD2_ISSUE_AT_TOP 0       // asm loop: the x panel / worker tiles of the NEXT step are requested behind the settle of h (0: at the end of the step, three / two steps ahead)
#endif
    constexpr bool TOP = ASM && FWD2_ISSUE_AT_TOP != 0;
    // xnext / wcur: the x panel of step t+1 (its products end this step) and the workers' tiles of step t; xfree / wfree: the register sets
    // of step t-1's, free now (TOP: they take the requests for step t+2 / t+1)
    auto step = [&](auto xpol, int t, u32x4_f (&xnext)[KX], u32x4_f (&xfree)[KX], u32x4_f (&wcur)[MVA], u32x4_f (&wfree)[MVA]) __attribute__((always_inline)) {
        // ---- h half of step t on top of the x half already in the accumulators
        F2STAMP(0);
        if constexpr (TOP) {
            // THE wait of the step, and it is exact: the only requests in flight are h_{t-1}'s (the x panel and the tiles were requested
            // a step ago, in front of it).  A retry of the settle below waits for its own KB loads and nothing else.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) pin(hv[kb]);
#pragma unroll
            for (int kb = 0; kb < KX; ++kb) pin(xnext[kb]);
#pragma unroll
            for (int p = 0; p < MVA; ++p) pin(wcur[p]);
        } else if constexpr (ASM) {
            // THE wait of the step: h_{t-1} has landed (and with it everything requested before it: this step's x panel and tiles);
            // the KX + MV loads requested behind it stay in flight
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KX + MV) : "memory");
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) pin(hv[kb]);
#pragma unroll
            for (int kb = 0; kb < KX; ++kb) pin(xnext[kb]);
#pragma unroll
            for (int p = 0; p < MVA; ++p) pin(wcur[p]);
        }
        settle(Local{}, hv, rh, (unsigned)((size_t)t * bph * 4));
        if constexpr (TOP) {
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < T) xissue(xpol, xfree, t + 2);       // (from memory: 1.6 steps until the x half of step t+1 reads it)
            if (t + 1 < T) wissue(wfree, t + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        F2STAMP(1);
        half_product(hv, wh, whh, whl, [](int) {});
        float (&rd)[NW][256][NT] = red_[0];
#pragma unroll
        for (int i = 0; i < 4; ++i)          // element lane*4 + i of the 16x16 tile: its four gates (N tiles) as one 16-byte word
            *reinterpret_cast<f32x4*>(&rd[wave][lane * 4 + i][0]) = (f32x4){acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
        const f32x4 pre = wsettle(wcur, t);                                  // bias + the x-product workers' tiles of frame t
        // (the panel of the x half behind B2 is checked HERE: it landed a step ago, and behind B2 its dozen compares sat in front
        //  of the x MFMAs of every step -- the layers above the bottom one ran 0.25 us per step behind it)
        if (ASM && FWD2_EARLY_XCHECK && decltype(xpol)::value != 2 && t + 1 < T) settle(Remote{}, xnext, rx, (unsigned)((size_t)(t + 1) * bph * 4));
        F2STAMP(2);
        lds_barrier();                                                       // B1: the partial sums of step t
        F2STAMP(3);
        if (epi) {
            epilogue(t, rd, pre);
        } else {
            if (t > 0) stores(t - 1);
        }
        F2STAMP(4);
        lds_barrier();                                                       // B2: every wave enters the MFMA phase together
        F2STAMP(5);
        // ---- x half of step t+1 into fresh accumulators; h_t is fetched under it
        F2STAMP(6);
        x_half(xpol, t, xnext, true);
        F2STAMP(7);
        if constexpr (TOP) {
        } else if constexpr (ASM) {
            __builtin_amdgcn_sched_barrier(0);
            xissue(xpol, xnext, t + 3 < T ? t + 3 : T - 1);      // (past the end: the last frame again -- one order of operations, one wait count)
            wissue(wcur, t + 2 < T ? t + 2 : T - 1);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            if (t + 1 < T) {
                if (t + 3 < T) xissue(xpol, xnext, t + 3);
            }
            if (t + 2 < T) wissue(wcur, t + 2);
        }
    };
#undef F2STAMP
    // ---- prologue: x half of step 0, the operands of steps 1 and 2, the initial state
    auto run = [&](auto xpol) __attribute__((always_inline)) {
        xissue(xpol, xa, 0);
        wait_all(xa);
        if (decltype(xpol)::value != 2) settle(Remote{}, xa, rx, 0u);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        half_product(xa, wx, wxh, wxl, [](int) {});
        xissue(xpol, xb, T > 1 ? 1 : 0);               // (clamped: a short sequence re-reads its last frame)
        if constexpr (!TOP) xissue(xpol, xa, T > 2 ? 2 : T - 1);
        issue(Local{}, hv, rh, 0u);                                              // slot 0: the packed initial state
        wissue(wa, 0);
        if constexpr (!TOP) wissue(wb, T > 1 ? 1 : 0);
        wait_all(xa); wait_all(xb); wait_all(hv); wait_all(wa); wait_all(wb);   // (once: the loop's own wait assumes its own order of requests)
        __syncthreads();
        for (int t = 0; t < T; t += 2) {
            step(xpol, t, xb, xa, wa, wb);                         // consumes x[t+1] (odd) at its end
            if (t + 1 < T) step(xpol, t + 1, xa, xb, wb, wa);      // consumes x[t+2] (even)
        }
        wait_all(xa); wait_all(xb); wait_all(wa); wait_all(wb);      // (the last steps' requests: nothing may land in a register after its last use)
    };
    if (l == 0) run(Local{}); else run(Remote{});
    __syncthreads();
    if (!epi) stores(T - 1);
#if defined(AMDSPEECH_DEVTRACE) && AMDSPEECH_DEVTRACE == 9      // (the knobs-only development build: tools/kernel_clocks.py)
    if (a.trace != nullptr && grp == 0 && ub == 0 && threadIdx.x == 0) {
        a.trace[0] = __builtin_readcyclecounter() - c_begin;
        a.trace[1] = wall_clock64() - t_begin;
    }
#endif
}


