// Part of lstm.hip -- what the whole-sequence ("dataflow") kernels share: sentinel, LDS-only barrier, FlowArgs, the in-register bf16 split, the parity tags.
// Not a standalone header: lstm.hip includes its kernel families in a fixed order, inside namespace amdspeech, after the helpers
// (layout, dropout multipliers, packs) they use.  Tuning macros (#ifndef ...) keep their defaults here; rnn-speech_amd/build.py
// passes overrides for development builds (AMDSPEECH_CXXFLAGS).

// ------------------------------------------------- dataflow forward (whole sequence, one launch)
// lstm_fwd_step pays, on every diagonal, a kernel boundary (~3.8 us), a cold first byte (~1.5 us) and the
// re-fetch of all 24 MB of weights (the per-XCD L2 is invalidated between kernels).  This kernel runs the
// whole sequence in ONE launch:
//  * a recurrence group = (layer l, 16-row batch tile mb) = H/16 workgroups of 16 units x 4 gates, ALL ON ONE
//    XCD (workgroups are dealt to the XCDs round-robin; each reads its XCC_ID and takes a ticket there).  The
//    loop-carried operand h_{t-1} is produced and consumed inside the group, so it only has to reach that XCD's
//    L2 -- plain stores, non-temporal loads (no L1 allocation, served by L2): 0.95 us per hand-off against
//    2.1-2.8 us through memory with sc1 (tools/xcd_bench.hip).  The input x_t of a layer comes from the group
//    of the layer below on ANOTHER XCD: write-through (sc1) stores, sc1 loads, fetched a step ahead;
//  * the weights stay on chip for all T steps, in registers: the 8 waves are SPECIALISED -- waves 0-3 ("h waves")
//    keep the h half of the workgroup's 64 gate columns (a K quarter each, 16*KQ VGPRs), waves 4-7 ("x waves")
//    the x half.  The h waves own the loop-carried path: wait for h_{t-1}, h product, K-split reduction through
//    LDS, the fused epilogue (hardware exp/rcp gates; c_{t-1}, h_{t-1} stay in registers), the hand-off store.
//    The x waves run one step ahead (their operand never depends on this group's progress), fetch their panels two
//    steps ahead, and take everything that is not loop-carried off the h waves: the write-through store of x to
//    the layer above and the BPTT stash (the epilogue passes the values through LDS).  gfx9 counts loads and
//    stores on one in-order vmcnt, so a write-through store issued by an h wave would sit in front of its next
//    poll for a memory round trip (~2 us);
//  * synchronisation between workgroups is pure dataflow, with no counters, flags or atomics: every slot of the
//    packed panel histories xph[l][t] / hph[l][t] is written exactly once per sequence and is pre-filled with a
//    NaN sentinel; a consumer (re)loads the float4s it needs until none carries the sentinel.  Inside a
//    workgroup: one s_barrier per step (B: epilogue done) for all 8 waves, and an LDS counter among the four h
//    waves where their partial sums meet (the x waves must not be held there).
// Measured and kept out (tools/xcd_bench.hip, tools/issue_bench.hip): s_setprio for the h waves, x waves that
// pause or leave gaps while the h waves run their MFMAs, a one-dword-per-producer probe before each full poll,
// warming the XCD's L2 with the next slots, re-loading only the pending fragments, starting the x waves' MFMA burst
// 0.3-1.3 us after the barrier (in xcd_bench mode 34 that lets the h waves' poll through: 5.06 -> 4.52 us; here it costs 4-8 %).
// Also measured: a RING of 8 h slots that stays in the XCD's L2 (each workgroup resets its part of a slot two steps after
// writing it) instead of one memory-cold slot per step: 7 % slower -- polls that come back sooner only add retry rounds.
// Every wait is bounded by a wall-clock limit; a time-out raises `err` (checked by amdspeech_lstm_status).
constexpr unsigned FLOW_SENTINEL = 0x7FC0DEADu;
// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() also drains vmcnt: every global load and store a wave has
// in flight (prefetches issued steps ahead, write-through stores that memory acknowledges ~2 us later) would have to
// complete at every step's barrier -- measured +0.7 us per step in the backward epilogue.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#ifndef FLOW_FWD_LDS_BARRIER
#define FLOW_FWD_LDS_BARRIER 1   // forward: the step barrier orders LDS only (loads / write-through stores stay in flight)
#endif
#if FLOW_FWD_LDS_BARRIER
#define FLOW_FWD_BARRIER() lds_barrier()
#else
#define FLOW_FWD_BARRIER() __syncthreads()
#endif
#ifndef FLOW_WORKER_WG_GATE
#define FLOW_WORKER_WG_GATE 1    // 1: one thread of a worker workgroup polls the chunk gate, then __syncthreads()
#endif
#ifndef FLOW_POLL_DELAY
#define FLOW_POLL_DELAY 6        // forward: s_sleep(1) periods (64 clocks each) between the step's barrier and the h waves' poll
#endif
#ifndef FLOW_REFILL_GROUPS
#define FLOW_REFILL_GROUPS 2     // backward: the next operand is re-loaded in place in this many batches under the down MFMAs
#endif

struct FlowArgs {
    const float* wp; const float* bias; long bias_stride;
    float* z; float* hs; float* cs; float* gates; const int* lengths;
    const float* xp0; float* xph; float* hph;
    unsigned* err;
    unsigned* tickets;             // [8] per-XCD arrival tickets (zeroed before the launch)
    int T, B, H, L;
    DropCfg drop;
    unsigned long long limit;      // wall_clock64 ticks (100 MHz) a workgroup may spend in this kernel
    unsigned long long* trace;     // dev builds (-DAMDSPEECH_DEVTRACE): wall-clock stamps of layer 1, unit block 3
    // x-product workers (lstm_fwd_flow2<., ., MV > 0>): the workgroups of the XCDs without a recurrence group form MV of every
    // recurrence wave's KB K blocks of x_t . W_ih and hand the groups pre-multiplied gate tiles through `xwp`
    float* xwp;                    // [T][L][nmt][H/16][MV][256][4 gates], every word tagged with xw_par (write-once per launch)
    unsigned xw_par;               // this launch's tag: the least significant mantissa bit of every word of xwp written by it
    int w_wpx;                     // worker workgroups per spare XCD (the others exit at once: room for amdspeech_lstm_beside_forward work)
    int w_wpw;                     // waves of a worker workgroup that take a role: 4 (waves 0-3, one per SIMD) or 8
    int trace_layer;               // dev builds only
    int cf_on, cf_nfw;             // the fused CTC head (ctc_flow.h): 0 = none; follower workgroups per spare XCD
    CtcFlow cf;                    // LAST, 64-byte aligned, and everything its role reads is INSIDE it (see CtcFlow)
};

typedef unsigned u32x4_f __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_f __attribute__((ext_vector_type(2)));

// ---- split precision ("bf16x3") inside the dataflow kernels: NO layout changes -- fragments arrive as f32 (memory, LDS,
// registers) and are split in registers.  Two consecutive f32 fragments (k-steps) make one 16x16x32 bf16 operand: a lane's
// element e = 0..7 is (fragment e/4, k-step e%4); A and B use the same order, and the contraction does not care which k sits
// where.  A product is hi.hi + hi.lo + lo.hi with f32 accumulation (the dropped lo.lo term is <= 2^-16 relative).
typedef __bf16 flow_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned flow_bf16_rne(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
typedef float flow_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 flow_bf16x2 __attribute__((ext_vector_type(2)));
// (vector conversions: hipcc emits v_cvt_pk_bf16_f32 -- round to nearest even, two values per instruction -- and v_pk_add_f32:
//  2.5 VALU instructions per value where the integer restatement of the rounding took 16; this sits on the loop-carried path)
__device__ __forceinline__ void flow_bf3_split(const float (&x)[8], u32x4_f& hi, u32x4_f& lo) {
#pragma unroll
    for (int p2 = 0; p2 < 4; ++p2) {
        const flow_f32x2 v = {x[2 * p2], x[2 * p2 + 1]};
        const flow_bf16x2 h = __builtin_convertvector(v, flow_bf16x2);
        const flow_f32x2 rest = v - __builtin_convertvector(h, flow_f32x2);
        const flow_bf16x2 l = __builtin_convertvector(rest, flow_bf16x2);
        hi[p2] = __builtin_bit_cast(unsigned, h);
        lo[p2] = __builtin_bit_cast(unsigned, l);
    }
}
// PR = 1 (bf16x3): hi.hi + hi.lo + lo.hi.  PR = 2 (bf16, round 4): the hi parts only -- ONE bf16 per value, one MFMA per product,
// what BASELINE configs[4] calls "bf16 MFMA"; the lo parts are dead code there and the compiler drops their computation.
template <int PR>
__device__ __forceinline__ f32x4 flow_bf_mma(f32x4 acc, const u32x4_f ah, const u32x4_f al, const u32x4_f bh, const u32x4_f bl) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(flow_bf16x8, ah), __builtin_bit_cast(flow_bf16x8, bh), acc, 0, 0, 0);
    if (PR == 1) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(flow_bf16x8, ah), __builtin_bit_cast(flow_bf16x8, bl), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(flow_bf16x8, al), __builtin_bit_cast(flow_bf16x8, bh), acc, 0, 0, 0);
    }
    return acc;
}

// "The flag is in the data" for REUSED slots (rings): the least significant mantissa bit of every word carries the parity of
// the slot's use count -- 1 ulp of the value, nothing to reset, and a torn 16-byte granule is harmless.
__device__ __forceinline__ u32x4_f flow_tag(const f32x4 v, const unsigned p) {
    u32x4_f r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (__float_as_uint(v[i]) & ~1u) | p;
    return r;
}
__device__ __forceinline__ bool flow_untagged(const u32x4_f v, const unsigned p) {      // some word still carries the old parity
    return (((v[0] ^ p) | (v[1] ^ p) | (v[2] ^ p) | (v[3] ^ p)) & 1u) != 0u;
}
__device__ __forceinline__ bool flow_pending(const u32x4_f v) {
    return v[0] == FLOW_SENTINEL || v[1] == FLOW_SENTINEL || v[2] == FLOW_SENTINEL || v[3] == FLOW_SENTINEL;
}

