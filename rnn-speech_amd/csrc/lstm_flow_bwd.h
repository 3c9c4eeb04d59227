// Part of lstm.hip -- lstm_bwd_flow2 (H <= 512: BPTT of the whole sequence in one launch) and its weight-gradient workers, DESIGN.md 4.2.
// Not a standalone header: lstm.hip includes its kernel families in a fixed order, inside namespace amdspeech, after the helpers
// (layout, dropout multipliers, packs) they use.  Tuning macros (#ifndef ...) keep their defaults here; rnn-speech_amd/build.py
// passes overrides for development builds (AMDSPEECH_CXXFLAGS).

// ------------------------------------------------- dataflow backward (whole sequence, one launch): arguments, GEMM workers
// A recurrence group = (layer l, 16-row batch tile mb) = H/16 workgroups of 16 units, ALL ON ONE XCD (workgroups are dealt to the
// XCDs round-robin; each reads its XCC_ID and takes a ticket there).  What is loop-carried is produced and consumed inside the
// group, so it only has to reach that XCD's L2: plain stores, non-temporal loads (no L1 allocation, served by L2) -- 0.95 us per
// hand-off against 2.1-2.8 us through memory with sc1 (tools/xcd_bench.hip).  Layer l-1 receives 1 KiB per workgroup and step from
// the layer above (its 16x16 slice of dX, through memory, sentinel-polled).  (Round 1's output-stationary lstm_bwd_flow -- every
// workgroup re-read the whole 128 KiB dG panel each step -- was removed in round 4; the kernel is lstm_bwd_flow2 below.)
struct FlowBwdArgs {
    const float* wq; const float* cs; const float* gates; float* dg; const float* dztop;
    float* prec; float* pdown;     // rings of lstm_bwd_flow2, zeroed before the launch: recurrent partial tiles [groups][2][H/16][H/16][256]
                                   // and down partials summed per K slice [groups][4][H/16][H/128][256]
    float* dxh;                    // dX history [L][T][bp][H] (slot [l] = gradient w.r.t. layer l's OUTPUT coming from
                                   // layer l+1), sentinel pre-filled, written through to memory
    unsigned* tickets;             // [8] per-XCD arrival tickets (zeroed before the launch)
    const int* lengths;
    unsigned* err;
    int T, B, H, L;
    DropCfg drop;
    unsigned long long limit;
    unsigned long long* trace;     // dev builds only
    int trace_layer;               // dev builds only
    int* progress;                 // [nmt] per layer-0 group: every frame >= progress[mb] is complete in memory (counts down from T)
    int nprog;                     // number of progress words the GEMM workers have to watch
    int prog_slack;                // a chunk [ta, tb) is released when every word is <= ta - prog_slack
    // in-kernel GEMM workers (the workgroups of the XCDs no recurrence group lives on): weight gradients of the
    // frames [w_t0, T), cut into w_pieces chunks, latest frames first
    const float* z; const float* hs; const float* kernels; float* dk; float* dbias; float* dz0;
    long kstride, bstride;
    int w_t0, w_pieces;
    int w_dz0;                     // 1: the workers also form dZ_0 of their frames
    int w_mode;                    // dev: see bwd_gemm_worker
    unsigned* w_counters;          // [w_pieces] (zeroed before the launch) or nullptr: the workers' quarter tiles of a chunk are dealt from a counter
    int dz0_inkernel;              // 1 (lstm_bwd_flow2): the layer-0 groups form dZ_0 = dG_0 . W_ih0^T themselves, masked, into dz0
    int cf_on;                     // the fused CTC head (ctc_flow.h): 0 = none -- dZ_top is then complete when the launch starts
    CtcFlow cf;                    // LAST, 64-byte aligned (see FlowArgs)
};

// ---- GEMM workers inside lstm_bwd_flow2 --------------------------------------------------------------------
// cfg2 uses 6 of the 8 XCDs for recurrence groups; the 64 workgroups dealt to the other two would exit.  Instead
// they run the time-independent weight-gradient GEMMs (dK_l += [Z_l;Hprev_l]^T.dG_l with the fused bias column
// sums, dZ_0 = dG_0.K_0x^T) of the frames the recurrence has already finished, while it is still running: each
// 512-thread workgroup is two 256-thread teams executing gemm_tile on their own LDS areas; a chunk of frames
// [ta, tb) is released when the progress word of the layer-0 group has passed ta - 2.  The teams synchronise among
// their own four waves through an LDS counter (TeamBarrier), so they drift apart and one team's operand staging
// overlaps the other's MFMAs; only the chunk gate is a workgroup-wide barrier.
template <int H>
__device__ void bwd_gemm_worker(const FlowBwdArgs& a, float* smem, int worker, int nworkers, unsigned long long t_begin) {
    const int T = a.T, B = a.B, L = a.L;
    // w_mode (dev, AMDSPEECH_FLOW_WORKER_MODE): 1 = only the first team of a workgroup computes (one wave per SIMD), 2 = nobody
    // does (the gates are still watched; gradients are then WRONG -- for power / clock experiments only)
    const bool active = a.w_mode == 0 || (a.w_mode == 1 && (threadIdx.x >> 8) == 0);
    const int team = a.w_mode == 1 ? worker : worker * 2 + (threadIdx.x >> 8), nteams = a.w_mode == 1 ? nworkers : nworkers * 2;
    const int tid = threadIdx.x & 255;
    float* lds = smem + (size_t)(threadIdx.x >> 8) * (2 * 2 * BK * LDS_LD);
    const size_t TB = (size_t)T * B;
    __shared__ unsigned team_count[2];
    if (threadIdx.x < 2) team_count[threadIdx.x] = 0;
    __syncthreads();
    TeamBarrier bar;
    bar.count = &team_count[threadIdx.x >> 8]; bar.waves = 4;
    for (int c = 0; c < a.w_pieces; ++c) {
        const int tb = T - (int)((long)(T - a.w_t0) * c / a.w_pieces), ta = T - (int)((long)(T - a.w_t0) * (c + 1) / a.w_pieces);
        if (tb <= ta) continue;
#if FLOW_WORKER_WG_GATE
        if (threadIdx.x == 0) {
            for (int pw = 0; pw < a.nprog; ++pw)
                while (__hip_atomic_load(a.progress + pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > ta - a.prog_slack) {
                    if (wall_clock64() - t_begin > a.limit) { atomicOr(a.err, 4u); break; }
                    __builtin_amdgcn_s_sleep(64);
                }
        }
        __syncthreads();
#else
        // every wave watches the gate itself: the two teams of a workgroup (and, in the LDS-free dK tasks, the four waves of a
        // team) never wait for each other at a chunk boundary
        for (int pw = 0; pw < a.nprog; ++pw)
            while (__hip_atomic_load(a.progress + pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > ta - a.prog_slack) {
                if (wall_clock64() - t_begin > a.limit) { if ((threadIdx.x & 63) == 0) atomicOr(a.err, 4u); break; }
                __builtin_amdgcn_s_sleep(64);
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        const int rows = (tb - ta) * B;
        const size_t r0 = (size_t)ta * B;
        // ---- dK_l: per layer two GEMMs (x rows, h rows), M = H, N = 4H, K = rows; split K so that a task is ~32 K tiles
        GemmArgs g;
        g.bias = nullptr; g.gate = nullptr; g.gate_err = nullptr; g.gate_need = 0; g.gate_limit = 0;
        g.M = H; g.N = 4 * H; g.K = rows; g.lda = H; g.ldb = 4 * H; g.ldc = 4 * H;
        g.tiles_n = 4 * H / BN; g.atomic = 1; g.a_vec = 1; g.b_vec = 1;
        const int tiles = (H / BM) * g.tiles_n;
        // as few K splits as keep every team busy: each task ends with a 128x128 tile of f32 atomics into dK, shared by
        // the two worker XCDs (5 splits of ~36 K tiles ran the workers at half the rate of the stand-alone GEMM)
        int splits = (nteams + L * 2 * tiles - 1) / (L * 2 * tiles);
        if (splits > rows / (BK * 8)) splits = rows / (BK * 8);
        if (splits < 1) splits = 1;
        g.k_chunk = ((rows + splits - 1) / splits + BK - 1) / BK * BK;
        splits = (rows + g.k_chunk - 1) / g.k_chunk;
        const int ndk = active ? L * 2 * tiles * splits : 0;
        auto dk_task = [&](int task, const int tid_) __attribute__((always_inline)) {
            const int split = task % splits; task /= splits;
            const int tile = task % tiles; task /= tiles;
            const int part = task & 1, l = task >> 1;
            const float* dg = a.dg + ((size_t)l * TB + r0) * 4 * H;
            g.A = part == 0 ? a.z + ((size_t)l * TB + r0) * H : a.hs + ((size_t)l * (T + 1) * B + r0) * H;
            g.B = dg;
            g.C = a.dk + l * a.kstride + (part ? (size_t)H * 4 * H : 0);
            g.colsum = part == 0 ? a.dbias + l * a.bstride : nullptr;
            gemm_tile_tn_direct(g, tile, split, tid_, true);     // (no LDS, no barrier: the two teams of a workgroup run free)
        };
        if (a.w_counters != nullptr && a.w_mode == 0) {
            // Fused CTC head: the teams that ran ctc_leader arrive here late.  The quarter tiles (one wave each: the tile code has no
            // barrier) of a chunk are DEALT from a counter instead of being assigned -- consecutive items are the four quarters of one
            // tile, so the waves of a team, which ask at about the same time, still share its operand strips through the L1.  The
            // next item is requested before the current one is computed.
            unsigned* ctr = a.w_counters + c;
            const int nitems = ndk * 4, ln = threadIdx.x & 63;
            auto fetch = [&]() -> int {
                unsigned v = 0u;
                if (ln == 0) v = atomicAdd(ctr, 1u);
                return __builtin_amdgcn_readfirstlane((int)v);
            };
            int item = fetch();
            while (item < nitems) {
                const int nxt = fetch();
                dk_task(item >> 2, (item & 3) * 64 + ln);
                item = nxt;
            }
        } else
        for (int t0 = team; t0 < ndk; t0 += nteams) dk_task(t0, tid);
        if (a.w_dz0 == 0) continue;      // (dZ_0 of these frames is left to the launch after the kernel)
        // ---- dZ_0 rows [r0, r0 + rows) = dG_0 . K_0[0:H, :]^T : M = rows, N = H, K = 4H, plain stores
        g.A = a.dg + r0 * 4 * H; g.B = a.kernels; g.C = a.dz0 + r0 * H; g.colsum = nullptr;
        g.M = rows; g.N = H; g.K = 4 * H; g.lda = 4 * H; g.ldb = 4 * H; g.ldc = H;
        g.tiles_n = H / BN; g.atomic = 0; g.k_chunk = 4 * H;
        const int ndz = ((rows + BM - 1) / BM) * g.tiles_n;
        for (int task = team; task < ndz; task += nteams)
            gemm_tile<true, true>(g, task, 0, lds, tid, 0, true, bar);
    }
}

// ------------------------------------------- dataflow backward, INPUT-STATIONARY recurrent product (whole sequence, one launch)
// BPTT contracts over the LONG axis (4H) to produce the SHORT one (H), so the recurrent product is input-stationary:
//   * a workgroup multiplies the dG tile it has JUST computed (16 rows x 64 gate columns of its own 16 units; it never leaves the
//     CU: registers -> 4 KiB of LDS -> MFMA A operand) with W_hh^T[its 64 rows, ALL H columns] and hands every workgroup j of its
//     group a 16x16 PARTIAL tile (1 KiB) of dh; workgroup j adds the H/16 partials it receives.  A consumer gathers 32 KiB per step
//     (an output-stationary product would re-read the whole 128 KiB panel in every workgroup), and nothing has to arrive before
//     the MFMAs can start;
//   * the partial tiles travel through a 2-slot RING per group in the XCD's L2.  The flag is IN the data: the least significant
//     mantissa bit of every float carries the parity of the slot's use count (1 ulp of a partial sum, 6e-8 relative), so there is
//     no sentinel to restore, no reset traffic, no counter, and a torn 16-byte granule is harmless (every word is tagged).  Slot
//     reuse is ordered by the data flow itself: a producer can only write step t-2 after it has gathered step t-1 from everybody,
//     which everybody stored after they had gathered step t (the slot's previous content);
//   * the "down" product dX_{l-1} = dG_l.W_ih^T (what the layer below needs, steps later) is NOT exchanged that way since round 4:
//     see "The down product" at the step -- a 2-D decomposition on the row-major dG rows, nothing polled;
//   * ALL EIGHT WAVES RUN THE SAME PHASE AT THE SAME TIME.  Measured (tools/trace_flow2.py) on a wave-specialised variant (waves
//     0-3: gather/epilogue/rec product; waves 4-7: down product and the memory work, half a step out of phase): beside a wave that
//     streams f32 MFMAs back to back, its partner on the SIMD issues NOTHING -- one store and eight loads took the whole 1.8 us of
//     a 128-MFMA stream, at any s_setprio and with or without a pause in front -- so two roles on one SIMD simply serialise (6.9 us
//     per step).  Work only overlaps INSIDE a wave (its own loads and stores between its own MFMAs).  The step (round 4):
//       [waves 0-3: the epilogue's dh-independent factors from the stash loaded a step ahead; sum the eight waves' down tiles of
//        frame t+6 | settle P[t+1] -> LDS; that sum -> Q ring] B1
//       [waves 0-3: epilogue(t) | waves 4-7: dX[t+9] and the row-major dG[t+1] out] B2
//       [Q gather, stash loads; rec MFMAs -> P[t] out] [down MFMAs of frame t+4, gather of P[t] issued half-way -> tiles to LDS]
//       [load the rows of dG[t+3]].   Two s_barriers per step; the hand-off latency of P[t] is covered by the down MFMAs.
// The in-kernel GEMM workers (bwd_gemm_worker) are gated by one progress word per layer-0 group.
#define FLOW2_BARRIER() __syncthreads()
#ifndef FLOW2_LOAD_AUX
#define FLOW2_LOAD_AUX 2          // cache policy of the ring gathers: 2 = nt (served by this XCD's L2), 16 = sc1
#endif
#ifndef FLOW2_GATHER_AT
#define FLOW2_GATHER_AT 2         // the gather of P[t] is issued after FLOW2_GATHER_AT quarters of the down MFMAs (4 = after them)
#endif
#ifndef FLOW2_DOWN_LAG
#define FLOW2_DOWN_LAG 4         // 3: the down product's operand is loaded behind B2 of the step that uses it; 4: at the END of the step before
#endif
#ifndef FLOW2_WINDOW
#define FLOW2_WINDOW 2
#endif
#ifndef FLOW2_CHECK_ORDER
#define FLOW2_CHECK_ORDER 0      // dev builds (tools/run_variants.sh): the down product's un-polled loads are CHECKED -- Q words carry a use-count
#endif                           // tag, dG is pre-filled with the sentinel by the host; a violation sets bits 8 / 16 of the error word
#ifndef FLOW2_FOLD_OFFSETS
#define FLOW2_FOLD_OFFSETS 1
#endif
#ifndef FLOW2_PRE_EPI
#define FLOW2_PRE_EPI 2           // the dh-independent factors of the epilogue formed ahead of B1: 2 = at the top of the step, 1 = at the end of the previous one (0: the raw stash handed over through LDS)
#endif
#ifndef FLOW2_STORE_AUX
#define FLOW2_STORE_AUX 0         // cache policy of the ring stores: 0 = plain (stay in this XCD's L2)
#endif
#ifndef FLOW2_Q
#define FLOW2_Q 4                 // the recurrent product cut BOTH ways (exact f32): the Q workgroups ub, ub ^ 1, ... share the K slice of THEIR Q x 64 gate
#endif                            // columns and contract it against 1/Q of the output units each (4 at H = 512, 2 at H = 256) -- see "Round 6" at the kernel
#ifndef FLOW2_DOWN_FIRST
#define FLOW2_DOWN_FIRST 2        // (Q = 4, steady state) this many output tiles of the DOWN product are formed between the own-tile and the
#endif                            // partner-tile MFMAs of the recurrent product: their 0.5 us cover the partner tiles' way through the L2
#ifndef FLOW2_XLOAD_AT
#define FLOW2_XLOAD_AT 4          // (Q = 2) the partner's dG tile is requested after this many quarters of the own-tile MFMAs (4: behind them)
#endif


template <int NTW, int PR, bool CF = false>     // 16-column N tiles (and gathered producer tiles) per wave: H/16/8 = H/128; PR: 0 f32, 1 bf16x3, 2 bf16;
                                                // CF: the instantiation with the fused CTC head's leader (see lstm_fwd_flow2)
__global__ __launch_bounds__(512) void lstm_bwd_flow2(FlowBwdArgs a_in) {
    constexpr bool BF3 = PR != 0;
    constexpr int NW = 8, H = 128 * NTW, NU = H / 16, NKB = 4 * H / 16, NRB = 2 * H / 16;
    // Round 6: the recurrent product cut in BOTH directions (exact f32, an even number of tiles per wave).  Q = 1 (rounds 2-5): a
    // workgroup contracts its OWN 64 gate columns against all H output units and hands every workgroup of the group a partial tile --
    // NU KiB out and NU KiB in per workgroup and step, the P ring's 6.3 MB per time step at 3x512 / B = 32, every byte of which reaches
    // the fabric once (the L2 writes dirty lines back as soon as misses pass through it).  A timing experiment with half of every
    // tile left out (same instructions, same ordering, wrong results) left the kernel's own time alone and took 0.3 ms off the STEP:
    // the launches around it run faster (the chip's power budget).  Q = 4 (H = 512) takes the same step again: quads share
    // 256 gate columns, ONE partial tile per wave (8 KiB out and in); the three partner tiles' way through the L2 (~1 us) is longer than
    // the own-tile MFMAs of both wave sets (0.5 us), so FLOW2_DOWN_FIRST tiles of the down product -- which depends on nothing of this
    // step -- are formed in between (measured, step on one box: Q = 2 12.10-12.22, Q = 4 with 0 / 2 / 3 such tiles 12.31 / 12.02-12.13 /
    // 12.25-12.31 ms).  Described for Q = 2: the workgroups 2p and 2p+1 share the K slice
    // of THEIR 128 gate columns; workgroup (p, nq) contracts it against half nq of the output units (NU/2 tiles) -- the same 64 weight
    // VGPRs, the same 64 MFMAs per wave.  What the two exchange is the INPUT: each stores its 4 KiB dG tile (tagged, in the order of
    // the LDS image) from the epilogue and copies the partner's into LDS under the own-tile half of the MFMAs (one more LDS barrier,
    // B3, between the halves -- the two wave sets of a SIMD take the matrix pipe one after the other anyway); the partial tiles are
    // NU/2 KiB out and in.  Ordering of the un-polled loads of the down product: a workgroup stores P[t] only after it has seen its
    // partner's tile X[t], which the partner stored behind ITS B1(t) -- so P[t] from one workgroup of every pair still implies that
    // the rows and Q tiles of ALL workgroups of step t+2 / t+1 have reached the L2 (see "The down product").
    constexpr int Q = PR != 0 ? 1 : (NTW % FLOW2_Q == 0 ? FLOW2_Q : (NTW % 2 == 0 && FLOW2_Q > 1 ? 2 : 1));      // 4 at H = 512, 2 at H = 256 (FLOW2_Q = 4)
    constexpr int NP = NTW / Q;                                                               // partial tiles a wave stores and gathers per step
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* a_lds = smem;                                                                      // [2 (step parity)][4 m][4 kq][16 i][4 g]: dG tiles as MFMA A fragments
    float (*red_r)[256] = reinterpret_cast<float (*)[256]>(smem + 2048);                     // [NW][256] partial sums of dh
    float (*stash_lds)[256] = reinterpret_cast<float (*)[256]>(smem + 2048 + NW * 256);      // [8][256] the next epilogue's forward stash
    float* qred = smem + 2048 + 2 * NW * 256;                                                 // [NW][NTW][64][4] per-wave partial tiles of the down product
    float* x_lds = qred + (FLOW2_WINDOW ? 2 : 1) * NW * NTW * 256;                            // (Q > 1) [Q - 1][4 m][4 kq][16 i][4 g]: the partners' dG tiles
    __shared__ unsigned s_ticket;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xF;
    if (threadIdx.x == 0) s_ticket = atomicAdd(a_in.tickets + xcc, 1u);
    __syncthreads();
    const int grp = (int)xcc, ub = __builtin_amdgcn_readfirstlane((int)s_ticket);      // (both wave-uniform, said so: the role dispatch below is then made of real branches)
    if (grp >= a_in.L * ((a_in.B + 15) / 16)) {           // an XCD without a recurrence group: GEMM workers
        if (ub < 32) {
            const int first = a_in.L * ((a_in.B + 15) / 16);
            // the CTC head's backward half (ctc_flow.h): beta, posterior, dlogits and dZ_top of one utterance per team, ahead of the
            // top layer's groups -- in the ~0.5 ms these workgroups would wait for their first chunk of frames
            if constexpr (CF) { if (a_in.cf_on) ctc_leader<H>(a_in.cf, smem, (grp - first) * 32 + ub, (8 - first) * 32); }
            if (a_in.w_pieces > 0) bwd_gemm_worker<H>(a_in, smem, (grp - first) * 32 + ub, (8 - first) * 32, wall_clock64());
        }
        return;
    }
    // (CF: the recurrence takes its OWN copy of the arguments, loaded here -- behind the role dispatch -- through a pointer the
    //  compiler cannot see through.  hipcc loads every kernel argument in the entry block and, with more arguments than scalar
    //  registers, spills them there; what the recurrence loops then re-read lane move by lane move depends on the allocation of the
    //  whole function, and with the leader's code in it that was 100 - 180 moves per step instead of 20)
    // (CF: the recurrence takes its OWN copy of the arguments, loaded here, behind the role dispatch, through a pointer the compiler
    //  cannot see through: see flow_args_again.  Pointers read that way are GENERIC to the compiler -- kernel arguments are known to
    //  be global -- and every access through them would be a flat_* instruction; with a flat access pending the wait-count pass
    //  gives up counting: vmcnt(0) at the top of every step instead of "the 12 youngest may stay in flight", +0.5 ms per launch.
    //  Hence FLOW_G at every plain access below: a no-op for kernel arguments, the address space said out loud for the copy)
    const FlowBwdArgs a = CF ? flow_args_again<FlowBwdArgs>() : a_in;
    const int T = a.T, B = a.B, L = a.L;
    const int nmt = (B + 15) / 16;
    if (ub >= NU) return;                                 // spare workgroups of a narrow layer
    const int l = grp / nmt, mb = grp % nmt;
    const size_t bph = (size_t)nmt * 16 * H;
    // Every layer but the bottom one owes the layer below dX = dG . W_ih^T (the "down" product).  The bottom layer's groups
    // would run half the MFMAs of the others and wait for them -- so they form dZ_0 = dG_0 . W_ih0^T (what the input Linear's
    // backward needs) with the same machinery, in the pipe time they have anyway: no [T*B, 4H] x [4H, H] GEMM after the kernel.
    const bool top = l + 1 == L, has_down = l > 0 || a.dz0_inkernel != 0;
    // (fused CTC head: dZ_top is PRODUCED during this launch, by ctc_leader on the worker XCDs -- the top layer then polls it like
    //  the other layers poll the gradient from the layer above)
    const bool top_ready = CF ? (top && a.cf_on == 0) : top;
    const unsigned long long t_begin = wall_clock64();
    const unsigned long long c_begin = __builtin_readcyclecounter();

    // ---- weights: B fragments of W_hh^T (rec) and W_ih^T (down) for this workgroup's 64 gate columns (K) and this
    // wave's NTW output tiles (N), straight from the K^T pack (pack_bwd_kernel): one float4 = the four k-steps of a gate
    // The two products are cut differently (see "down product" at the step): rec -- this workgroup's OWN 64 gate columns x all H
    // outputs (wave: NTW of the NU output tiles); down -- the gate columns of the 8 workgroups of K slice ks (wave: ONE of
    // them, dks) x the NTW output tiles of N slice ns.  Same register count either way.
    const int ks = ub >> 3, ns = ub & 7, dks = ks * 8 + wave;
    const int nq = ub & (Q - 1), pair = ub / Q;                                   // (Q > 1) the part of the output units this workgroup forms; its K slice
    f32x4 wr[NTW][4], wd[NTW][4];                                                 // (Q > 1) wr[j * NP + n]: tile j of the slice (0 own, j: workgroup ub ^ j's) x output tile n
    {
        const float* base = a.wq + (size_t)l * NRB * NKB * 256 + lane * 4;
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nt = nq * (NU / Q) + wave * NP + (n % NP);
                const int kb = g * (H / 16) + (ub ^ (n / NP));
                wr[n][g] = *(const f32x4 __attribute__((address_space(1)))*)(FLOW_G(const float, base) + ((size_t)(H / 16 + nt) * NKB + kb) * 256);
                wd[n][g] = has_down ? *(const f32x4 __attribute__((address_space(1)))*)(FLOW_G(const float, base) + ((size_t)(ns * NTW + n) * NKB + g * (H / 16) + dks) * 256)
                                    : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
    }

    // split-precision mode: the weight fragments as bf16 hi / lo pairs (same register count), built once; a 32-wide K block
    // is a pair of gates (g = 2s, 2s+1) x the four k-steps
    u32x4_f wrh[BF3 ? NTW : 1][2], wrl[BF3 ? NTW : 1][2], wdh[BF3 ? NTW : 1][2], wdl[BF3 ? NTW : 1][2];
    if (BF3) {
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const float xr[8] = {wr[n][2 * sp][0], wr[n][2 * sp][1], wr[n][2 * sp][2], wr[n][2 * sp][3],
                                     wr[n][2 * sp + 1][0], wr[n][2 * sp + 1][1], wr[n][2 * sp + 1][2], wr[n][2 * sp + 1][3]};
                flow_bf3_split(xr, wrh[n][sp], wrl[n][sp]);
                const float xd[8] = {wd[n][2 * sp][0], wd[n][2 * sp][1], wd[n][2 * sp][2], wd[n][2 * sp][3],
                                     wd[n][2 * sp + 1][0], wd[n][2 * sp + 1][1], wd[n][2 * sp + 1][2], wd[n][2 * sp + 1][3]};
                flow_bf3_split(xd, wdh[n][sp], wdl[n][sp]);
            }
    }

    // ---- element identity: thread (bl, u) of waves 0-3 owns (batch row b, unit) of the epilogue; the same thread
    // index in waves 4-7 owns that element of the dX tile this workgroup finishes for the layer below
    const int bl = (threadIdx.x & 255) >> 4, u = threadIdx.x & 15;
    const int b = mb * 16 + bl, unit = ub * 16 + u;
    const bool epi = threadIdx.x < 256;
    const bool pok = b < B;
    const int bc = min(b, B - 1);
    const size_t bec = (size_t)bc * H + unit;
    const int len = FLOW_G(const int, a.lengths)[bc];
    float dcin = 0.0f;
    const int e = ((bl >> 2) * 16 + u) * 4 + (bl & 3);           // this element inside a 16x16 accumulator tile
    const int a_slot = (((u & 3) * 4 + (u >> 2)) * 16 + bl) * 4;   // its four gates inside a dG tile: [m = u%4][kq = u/4][i = bl][g]

    // ---- the rings of this group.  P (recurrent partials, THE loop-carried hand-off): [2 slots][NU consumers][NU producers][256],
    // every word tagged.  Q (down partials, summed over a K slice): [4 slots][NU consumers][KS K slices][256], plain words.
    constexpr int KS = NU / 8;                                     // K slices of the down product (8 producers each, one per wave)
    constexpr unsigned SLOT_BYTES = (unsigned)NU * (NU / Q) * 1024u, QSLOT_BYTES = (unsigned)NU * KS * 1024u;
    // (the host sizes and zeroes the P ring for Q = 1: [2 slots][NU][NU][256] per group -- Q = 2 uses half of every group's share)
    const auto rp = __builtin_amdgcn_make_buffer_rsrc(a.prec + (size_t)grp * 2 * NU * NU * 256, 0, 2u * SLOT_BYTES, 0x00020000);
    const auto rq = __builtin_amdgcn_make_buffer_rsrc(a.pdown + (size_t)grp * 4 * NU * KS * 256, 0, 4u * QSLOT_BYTES, 0x00020000);
    const auto rdg = __builtin_amdgcn_make_buffer_rsrc(a.dg + (size_t)l * T * B * 4 * H, 0, (unsigned)((size_t)T * B * 4 * H * 4), 0x00020000);
    // X (Q = 2): the dG tiles the two workgroups of a pair show each other, [2 slots][NU][1024] per group, tagged like P; behind the Q rings
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(a.pdown + (size_t)L * nmt * 4 * NU * KS * 256 + (size_t)grp * 2 * NU * 1024, 0,
                                                      2u * NU * 4096u, 0x00020000);
    unsigned gather_off = (unsigned)(((ub * (NU / Q) + wave * NP) * 256 + lane * 4) * 4);      // + q KiB: producer (pair) wave*NP + q
    const unsigned store_off = (unsigned)((((nq * (NU / Q) + wave * NP) * (NU / Q) + pair) * 256 + lane * 4) * 4);     // + n*(NU/Q) KiB: consumer nq*NU/Q + wave*NP + n
    bool dead = false;
    u32x4_f gp[NP];
    auto issue = [&](decltype(rp) rs, u32x4_f (&buf)[NP], int slot) {
#pragma unroll
        for (int q = 0; q < NP; ++q)
            buf[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, gather_off + (unsigned)(q * 1024), (unsigned)slot * SLOT_BYTES, FLOW2_LOAD_AUX);
    };
    auto total = [&](const u32x4_f (&buf)[NP]) {
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NP; ++q)
            s += (f32x4){__uint_as_float(buf[q][0]), __uint_as_float(buf[q][1]), __uint_as_float(buf[q][2]), __uint_as_float(buf[q][3])};
        return s;
    };
    // Check the gathered tiles and add them up.  The sum exists TWICE, once per path: hipcc guards every later use of a register
    // that a retry loop MAY have re-loaded with the wait count of the re-load (vmcnt(0): nothing younger in flight there), so a
    // sum behind the merge of the two paths waited, on every step, for whatever the wave had issued since the gather -- the
    // write-back stores of the Q tiles in round 2's loop.  On the straight path the tag checks have already waited for exactly
    // the gathered tiles and nothing else.
    auto settle_total = [&](decltype(rp) rs, u32x4_f (&buf)[NP], int slot, unsigned par) __attribute__((always_inline)) -> f32x4 {
        bool again = false;
#pragma unroll
        for (int q = 0; q < NP; ++q) again = again || flow_untagged(buf[q], par);
        if (!__any(again) || dead) return total(buf);
        while (true) {
            if (wall_clock64() - t_begin > a.limit) { dead = true; if (lane == 0) __hip_atomic_fetch_or(FLOW_G(unsigned, a.err), 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            issue(rs, buf, slot);
            again = false;
#pragma unroll
            for (int q = 0; q < NP; ++q) again = again || flow_untagged(buf[q], par);
            if (!__any(again)) break;
        }
        f32x4 r = total(buf);
        asm volatile("; settled after a retry" : "+v"(r));      // (keeps the two sums apart)
        return r;
    };
    auto store_tiles = [&](decltype(rp) rs, const f32x4 (&acc)[NTW], int slot, unsigned par) {
        // HARDWARE HAZARD (gfx950, measured; not modelled by hipcc 7.2): a buffer_store_dwordx4 whose soffset is an SGPR
        // still reads its data VGPRs for a few cycles after issue -- a VALU write to them in the next slots corrupts the
        // stored tile (seen as wrong dwords 0 and 3 of the tiles of the arbitration-favoured waves).  The compiler
        // only inserts the wait state when soffset is NOT a register, so the slot offset goes into voffset.
#pragma unroll
        for (int n = 0; n < NP; ++n)
            __builtin_amdgcn_raw_buffer_store_b128(flow_tag(acc[n], par), rs,
                                                   store_off + (unsigned)(n * (NU / Q) * 1024) + (unsigned)slot * SLOT_BYTES, 0, FLOW2_STORE_AUX);
    };
    // (Q > 1) this thread's 16 bytes of the K slice's exchange: its own tile's element group out (epilogue threads, a_slot), the partners' in
    // (threads 0-255 copy the 4 KiB tile linearly into LDS).  Same 2-slot discipline as P: the partner overwrites slot (t & 1) at its
    // epilogue(t), which follows its gather of my P[t+1], which I stored after the MFMAs of step t+1 -- after those of step t+2 that read it.
    const unsigned x_store_off = (unsigned)((ub * 1024 + a_slot) * 4);
    unsigned x_load_off = (unsigned)((threadIdx.x & 255) * 16);      // + workgroup (ub ^ j) * 4 KiB
    auto x_issue = [&](int slot, u32x4_f (&xv)[Q > 1 ? Q - 1 : 1]) {
#pragma unroll
        for (int j = 1; j < Q; ++j)
            xv[j - 1] = __builtin_amdgcn_raw_buffer_load_b128(rx, x_load_off + (unsigned)((ub ^ j) * 4096) + (unsigned)slot * (unsigned)(NU * 4096), 0, FLOW2_LOAD_AUX);
    };
    // parity expected in slot (t & 1) for the P tiles of step t: the slot's use count, starting at 1 (the rings are zeroed)
    auto parity = [&](int t) -> unsigned { return ((((unsigned)(T - 1 - t)) >> 1) & 1u) ^ 1u; };

    auto ftanh = [](float x) {
        const float x2 = x * x;
        const float small = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f - 0.053968254f * x2)));
        const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
        return fabsf(x) < 0.25f ? small : big;
    };
    // forward stash of this thread's element: buffer resources over this layer's slices, ONE loop-invariant 32-bit offset per
    // thread and tensor, the frame in the scalar offset (five 64-bit pointers walked backwards in time cost ten VGPRs of a kernel
    // that sits at the 256-register limit of two waves per SIMD)
    struct Stash { float gi, gj, gf, go, c, cp, dtop; };
    const auto r_gate = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gates) + (size_t)l * T * B * 4 * H, 0,
                                                          (unsigned)((size_t)T * B * 4 * H * 4), 0x00020000);
    const auto r_cs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.cs) + (size_t)l * (T + 1) * B * H, 0,
                                                        (unsigned)((size_t)(T + 1) * B * H * 4), 0x00020000);
    const auto r_top = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dztop), 0, (unsigned)((size_t)T * B * H * 4), 0x00020000);
    const auto r_dx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dxh) + (size_t)l * T * bph, 0, (unsigned)((size_t)T * bph * 4),
                                                        0x00020000);      // gradient from the layer above (another XCD)
    unsigned vo_gate = (unsigned)(((size_t)bc * 4 * H + unit) * 4);
    const unsigned vo_bec = (unsigned)(bec * 4);
    const unsigned vo_dx = (unsigned)(((size_t)b * H + unit) * 4);
    const auto r_up = top ? r_top : r_dx;                   // (CF only)
    const float* up_base = top ? a.dztop : a.dxh + (size_t)l * T * bph;
    const size_t up_step = top ? (size_t)B * H : bph;
    const unsigned vo_up = top ? (unsigned)(bec * 4) : vo_dx, up_step_b = top ? (unsigned)((size_t)B * H * 4) : (unsigned)(bph * 4);
    const unsigned gate_step_b = (unsigned)((size_t)B * 4 * H * 4), cs_step_b = (unsigned)((size_t)B * H * 4), dx_step_b = (unsigned)(bph * 4);
#define FLOW2_LDF(rs, vo, so, aux) __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, aux))
    auto poll_dx = [&](const float* p) -> float {
        while (true) {
            const float v = __hip_atomic_load(FLOW_G(const float, p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__float_as_uint(v) != FLOW_SENTINEL || dead) return v;
            if (wall_clock64() - t_begin > a.limit) { dead = true; __hip_atomic_fetch_or(FLOW_G(unsigned, a.err), 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return 0.0f; }
        }
    };
    // The forward stash of the NEXT epilogue is fetched at the start of the MFMA phase, a whole step ahead (measured: the epilogue
    // took 0.8-1.3 us with the loads in it, 0.44 us without).  FLOW2_PRE_EPI = 0 (rounds 2-3): waves 4-7 hand their copy to the
    // epilogue waves through LDS at the end of the step; 2 (default): waves 0-3 turn THEIR copy into the epilogue's dh-independent
    // factors while they wait for the P tiles at the top of the next step (precompute, below).
    Stash sv;                     // in flight from B2 to the top of the next step
    float sv_dx = 0.0f;
    auto fetch_stash = [&](const int tf) {                // frame tf (wave-uniform)
        const unsigned sg = (unsigned)tf * gate_step_b, sc = (unsigned)tf * cs_step_b;
        sv.gi = FLOW2_LDF(r_gate, vo_gate, sg, 0);             sv.gj = FLOW2_LDF(r_gate, vo_gate + H * 4, sg, 0);
        sv.gf = FLOW2_LDF(r_gate, vo_gate, sg + 2 * H * 4, 0); sv.go = FLOW2_LDF(r_gate, vo_gate + H * 4, sg + 2 * H * 4, 0);
        sv.cp = FLOW2_LDF(r_cs, vo_bec, sc, 0);                sv.c = FLOW2_LDF(r_cs, vo_bec, sc + cs_step_b, 0);      // c_{t-1}; c_t is one frame further
        // (both unconditional -- the buffers exist for every layer and padded row, the epilogue picks the one that applies: a load
        //  under a condition costs a branch and an s_waitcnt vmcnt(0) at the join)
        if constexpr (CF) {
            // the gradient from above through ONE descriptor (the top layer's dZ_top, produced by ctc_leader on another XCD during
            // this launch, or dX from the layer above: both sentinel-polled, both sc1) -- one load per step and five scalar registers
            // less than the two unconditional loads below; the instantiation with the head needs them (see lstm_fwd_flow2's CF)
            sv_dx = FLOW2_LDF(r_up, vo_up, (unsigned)tf * up_step_b, 16);
            sv.dtop = 0.0f;      // (NOT a copy of sv_dx: a register copy of a value just requested is a wait for it, here, at the bottom of the step)
        } else {
            sv.dtop = FLOW2_LDF(r_top, vo_bec, sc, 0);
            sv_dx = FLOW2_LDF(r_dx, vo_dx, (unsigned)tf * dx_step_b, 16);      // sc1: written by another XCD
        }
    };
    auto publish_stash = [&]() {
        const int i = threadIdx.x & 255;
        stash_lds[0][i] = sv.gi; stash_lds[1][i] = sv.gj; stash_lds[2][i] = sv.gf; stash_lds[3][i] = sv.go;
        stash_lds[4][i] = sv.c; stash_lds[5][i] = sv.cp; stash_lds[6][i] = sv.dtop; stash_lds[7][i] = sv_dx;
    };
#if FLOW2_PRE_EPI
    // Everything of the epilogue that does not depend on dh is formed by the epilogue waves THEMSELVES, from their own copy of
    // the stash loads, in the idle time at the top of the step (they reach the settle ~1 us before the P tiles do): what is left
    // behind B1, on the loop-carried path, is the eight-word sum, six multiply-adds and one LDS store.
    struct Pre { float a, bx, by, bz, bw, gf, dz; } pf = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto precompute = [&](const int tf) {     // sv: frame tf's stash
        // (pins the first use of the loaded values HERE: without it a register copy of one of them lands in front of the rec
        //  MFMAs, with a wait for the stash loads issued a few instructions earlier)
        asm volatile("" : "+v"(sv.gi), "+v"(sv.gj), "+v"(sv.gf), "+v"(sv.go), "+v"(sv.c), "+v"(sv.cp), "+v"(sv.dtop), "+v"(sv_dx));
        const bool live = pok && tf < len;
        const float tc = ftanh(sv.c);
        pf.a = live ? sv.go * (1.0f - tc * tc) : 0.0f;
        pf.bx = live ? sv.gj * sv.gi * (1.0f - sv.gi) : 0.0f;
        pf.by = live ? sv.gi * (1.0f - sv.gj * sv.gj) : 0.0f;
        pf.bz = live ? sv.cp * sv.gf * (1.0f - sv.gf) : 0.0f;
        pf.bw = live ? tc * sv.go * (1.0f - sv.go) : 0.0f;
        pf.gf = live ? sv.gf : 0.0f;
        // gradient from above x its dropout multiplier -- or the sentinel itself, if the layer above has not delivered yet
        const float dup = CF ? sv_dx : (top ? sv.dtop : sv_dx);
        const float dz = dup * zmult(a.drop, l + 1, (uint32_t)((size_t)tf * B * H + bec));
        pf.dz = (top_ready || __float_as_uint(dup) != FLOW_SENTINEL) ? dz : dup;
    };
    fetch_stash(T - 1);
    if (FLOW2_PRE_EPI == 1 && epi) precompute(T - 1);
#else
    if (!epi) { fetch_stash(T - 1); publish_stash(); }    // frame T-1 (made visible by the first B1)
#endif
#if defined(AMDSPEECH_DEVTRACE) && AMDSPEECH_DEVTRACE == 1
    // (AMDSPEECH_TRACE_LAYER: which layer's unit block 3 is stamped; default the top one, which sets the pace)
#ifndef FLOW2_TRACE_WAVE
#define FLOW2_TRACE_WAVE 5        // the second traced wave (4: the partner of wave 0 on its SIMD)
#endif
    const bool tracing = a.trace != nullptr && l == a.trace_layer && ub == 3 && mb == 0 && (wave == 0 || wave == FLOW2_TRACE_WAVE) && lane == 0;
#define BSTAMP(i) do { if (tracing && t >= 500 && t < 508) a.trace[128 + ((t - 500) * 2 + (wave ? 1 : 0)) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define BSTAMP(i) do { } while (0)
#endif
    constexpr int DL = FLOW2_DOWN_LAG;                   // the down product of step t is frame t + DL's
    // FLOW2_WINDOW = 0: everything else about the down product happens in waves 4-7's B1-B2 window (wave sum + Q store of frame
    // t+DL+1, dX of frame t+DL+4, the row-major dG copy).  2: the wave sum moves to waves 0-3's idle time at the top of the step
    // (they reach the settle ~1 us before the P tiles do), two steps later from a double-buffered qred; dX and the row-major copy
    // stay in the window.  1: those two move behind B2 as well, where waves 4-7 wait for the matrix pipe anyway (measured: beside
    // their partners' MFMA stream the thirty instructions crawl and hold their own rec MFMAs back by more than the window saved).
    constexpr bool WO = FLOW2_WINDOW != 0;
    constexpr int RL = WO ? DL + 2 : DL + 1;             // wave sum + Q store: frame t + RL (WO: at the top of step t)
    constexpr int GL = RL + 2;                           // gather of Q issued behind B2 of step t: frame t + GL
    constexpr int XL = GL + 1;                           // dX leaves in step t: frame t + XL
    const int t_last = has_down ? -XL : -1;
    // The weight fragments (and the first stash) are loaded ONCE, above.  Without an explicit wait here hipcc's waitcnt pass
    // merges "weight loads still pending" from the loop entry into the loop header and guards every first use of a weight
    // register INSIDE the loop with s_waitcnt vmcnt(16) ... vmcnt(1): ladders in the middle of the MFMA streams that at run
    // time wait for whatever is in flight then.
    FLOW_WEIGHTS_RESIDENT();
    // ---- "The down product".  dX_{l-1} = dG_l . W_ih^T is NOT on this layer's loop-carried path (the layer below consumes it
    // steps later), so it does not use the rec product's 32-way exchange of partial tiles (rounds 2-3: 1 MiB written and 1 MiB
    // gathered per group and step through a 3 MiB ring that did not fit the 4 MiB L2 next to the P ring -- 8x the algorithmic
    // fabric traffic, 0.85 us of the step).  Round 4: a 2-D decomposition that re-uses what the kernel writes anyway.
    //   * operand: the ROW-MAJOR dG rows this group stores for the weight-gradient products.  Wave w of workgroup (ks, ns) reads
    //     the 64 gate columns of producer dks = 8 ks + w straight into MFMA A fragments -- lane (i, kq) takes 16 bytes of row i
    //     per gate: the four k steps of a float4 are units 4 kq + m, exactly the order of the packed weights -- 4 KiB per wave,
    //     128 KiB per workgroup-step summed over the group ... no LDS staging, nothing new is written;
    //   * product: [16 x 64] . W_ih^T[64, NTW tiles of N slice ns]: the same 16 NTW MFMAs per wave as before;
    //   * the eight waves' partial tiles meet in LDS (qred, double-buffered), waves 0-3 add them two steps later while they wait
    //     for the P tiles at the top of a step, and the workgroup stores NTW tiles (not 32) into the Q ring; the consumer adds its
    //     KS = H/128 tiles, one dword per K slice.
    // Nothing of this is polled.  Order comes from the P hand-off alone.  gfx9 retires a wave's loads and stores IN ORDER on one
    // counter, so a wave that has settled its gather of P[t+1] (top of step t; the youngest loads it has in flight) has also seen
    // the acknowledgement of every store it issued BEFORE that gather (the gather goes out half-way through the down MFMAs of step
    // t+1); behind B1(t) that holds for all waves of the workgroup, and only then (behind B2(t)) does any of them store P[t].
    // Hence: once P[t] of EVERY producer has settled here (top of step t-1), their row-major dG[t+2] (stored in the window of step
    // t+1) and the Q tiles they stored at the top of step t+1 are in this XCD's L2, and loads issued from now on (nt: no L1
    // allocation) see them.  The same chain orders slot reuse: a consumer stores P[s] only after the Q gather it issued behind
    // B2(s+1) has returned, and a producer writes a Q slot only behind the settle of everybody's P of the step before -- by then
    // the slot's previous frame (four frames later in time, read two steps earlier) has been consumed: four slots.  (The step
    // barriers themselves compile to "s_waitcnt lgkmcnt(0); s_barrier" here -- no vmcnt drain -- which is why the argument goes
    // through the settle.)  -DFLOW2_CHECK_ORDER=1 checks all of it at run time (tags on the Q words, a sentinel under the dG rows).
    // With the defaults (FLOW2_DOWN_LAG 4, FLOW2_WINDOW 2) frame f's down product is: rows loaded at the end of step f-3, MFMAs in
    // step f-4, wave sum at the top of step f-6 (Q store behind that step's settle), gather behind B2 of step f-8, dX out in the
    // window of step f-9.
    // Both waves of a SIMD run the SAME phase at the same time: beside a wave that streams f32 MFMAs back to back its partner
    // issues nothing at all (see the header comment), so work is only ever overlapped INSIDE a wave.
    // The body exists four times: with / without a "down" product (compile-time, so that the two kinds of group do not share
    // register assignments and wait states through a control-flow merge), and as a steady-state body (1 <= t <= T-8: every
    // "does frame t+k exist" test is true at compile time -- no conditionally issued memory operation, so the wait counts are
    // exact) next to the general one for the first and the last frames.
    unsigned dg_vo = (unsigned)((((size_t)min(mb * 16 + (lane & 15), B - 1) * 4 * H) + dks * 16 + 4 * (lane >> 4)) * 4);
    const unsigned dg_step_b = (unsigned)((size_t)B * 4 * H * 4);
    unsigned q_load_off = (unsigned)(((ub * KS) * 256 + e) * 4);
#if FLOW2_CHECK_ORDER
    auto qpar = [&](int f) -> unsigned { return ((((unsigned)(T - 1 - f)) >> 2) & 1u) ^ 1u; };      // tag of frame f's use of Q slot f & 3
#endif
    u32x4_f av2[4];                    // dG[t+3], producer dks: [gate] x the four units 4 kq + m
    float gq[KS];                      // (waves 4-7) this element of the KS down tiles of frame t+6
    auto uni = [](auto v) { return (decltype(v))__builtin_amdgcn_readfirstlane((int)v); };      // wave-uniform, said explicitly
    auto load_av2 = [&](const int f) __attribute__((always_inline)) {      // rows of frame f (wave-uniform), legal once P[f-2] has settled here
        const unsigned so = uni((unsigned)f * dg_step_b);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            av2[g] = __builtin_amdgcn_raw_buffer_load_b128(rdg, dg_vo + (unsigned)(g * H * 4), so, FLOW2_LOAD_AUX);
    };
    auto store_q = [&](const f32x4 sq, const int f, const int n) __attribute__((always_inline)) {      // tile n of N slice ns, frame f
#if FLOW2_CHECK_ORDER
        const u32x4_f sv4 = flow_tag(sq, qpar(f));
#else
        const u32x4_f sv4 = {__float_as_uint(sq[0]), __float_as_uint(sq[1]), __float_as_uint(sq[2]), __float_as_uint(sq[3])};
#endif
        __builtin_amdgcn_raw_buffer_store_b128(sv4, rq, (unsigned)(((((ns * NTW + n) * KS + ks) * 256) + lane * 4) * 4) + (unsigned)(f & 3) * QSLOT_BYTES,
                                               0, 0);      // (no SGPR soffset: see store_tiles)
    };
    // what waves 4-7 owe per step besides MFMAs (see FLOW2_WINDOW for where it runs)
    auto rest_of_window = [&](const int t, auto hd_tag, auto steady_tag) __attribute__((always_inline)) {
        constexpr bool HD = decltype(hd_tag)::value, S = decltype(steady_tag)::value;
        if (HD && (S || (t + XL >= 0 && t + XL < T)) && pok) {
            // dX_{l-1}[t+XL]: one dword per K slice, gathered behind B2 of step t+1
            float dx = gq[0];
#pragma unroll
            for (int k = 1; k < KS; ++k) dx += gq[k];
#if FLOW2_CHECK_ORDER
            {   // dev: every word must carry the tag of THIS use of its slot
                unsigned bad = 0u;
#pragma unroll
                for (int k = 0; k < KS; ++k) bad |= (__float_as_uint(gq[k]) ^ qpar(t + XL)) & 1u;
                if (bad) __hip_atomic_fetch_or(FLOW_G(unsigned, a.err), 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#endif
            if (l > 0)
                __hip_atomic_store(FLOW_G(float, a.dxh) + ((size_t)(l - 1) * T + t + XL) * bph + (size_t)b * H + unit, dx, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            else          // dZ_0, row-major, with the layer-0 input dropout mask (read by the launches after this kernel)
                FLOW_G(float, a.dz0)[((size_t)(t + XL) * B + b) * H + unit] = dx * zmult(a.drop, 0, (uint32_t)((size_t)(t + XL) * B * H + bec));
        }
        if (HD && !WO && (S || (t + RL >= 0 && t + RL < T)) && wave < 4 + NTW) {
            // the eight waves' partial tiles of frame t+RL (left in LDS at the end of step t+1): wave 4+n adds tile n and
            // stores it for consumer ns*NTW + n
            const float* src = qred + ((wave & 3) * 64 + lane) * 4;
            f32x4 sq = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
            for (int w = 1; w < NW; ++w) sq += *reinterpret_cast<const f32x4*>(src + w * NTW * 256);
            store_q(sq, t + RL, wave & 3);
        }
        if ((S || (t + 1 >= 0 && t + 1 < T)) && pok) {
            // row-major copy of dG[t+1] (the OTHER LDS tile) for the weight-gradient GEMMs and this group's own down product
            // (write-through: the in-kernel workers may read it before this kernel ends): thread (bl, u) stores gate u/4,
            // units 4*(u%4)..+3 of row bl
            const float* tile = a_lds + ((t + 1) & 1) * 1024;
            const int g = u >> 2, q4 = u & 3;
            u32x4_f row;
#pragma unroll
            for (int m = 0; m < 4; ++m) row[m] = __float_as_uint(tile[((m * 4 + q4) * 16 + bl) * 4 + g]);
            __builtin_amdgcn_raw_buffer_store_b128(row, rdg, (unsigned)((((size_t)(t + 1) * B + b) * 4 * H + g * H + ub * 16 + q4 * 4) * 4),
                                                   0, 16);      // sc1; (no SGPR soffset: see store_tiles)
        }
    };
    auto step = [&](const int t_in, auto hd_tag, auto steady_tag) __attribute__((always_inline)) {
        constexpr bool HD = decltype(hd_tag)::value, S = decltype(steady_tag)::value;
        // (the frame index is wave-uniform; said explicitly, or hipcc keeps it in a VGPR and wraps every buffer access whose
        //  scalar offset depends on it in a waterfall loop)
        const int t = __builtin_amdgcn_readfirstlane(t_in);
        constexpr int DF = (Q == 4 && HD && S && !BF3) ? FLOW2_DOWN_FIRST : 0;      // tiles of the down product formed inside the rec phase
        const bool rec_on = S || (HD ? t > t_last : t > 0);       // (HD, t <= 0: the product of a stale tile, for the hand-off's sake)
        BSTAMP(0);
#if FLOW2_FOLD_OFFSETS
        // (loop-invariant "base + k KiB" offsets are hoisted out of the loop one VGPR each -- fourteen of them -- before
        //  instruction selection could fold the constant into the load's immediate field; a base the compiler cannot see through
        //  keeps the additions in the loop body, where they fold)
        asm volatile("" : "+v"(gather_off), "+v"(q_load_off), "+v"(dg_vo), "+v"(vo_gate));
#endif
#if FLOW2_PRE_EPI == 2
        if (epi && (S || t >= 0)) precompute(t);
        __builtin_amdgcn_sched_barrier(0);
#endif
        // ---- (WO) waves 0-3 have ~1 us to spare here: wave n adds the eight waves' partial tiles n of frame t+RL (qred of two
        // steps ago) -- stored BEHIND the settle, so that the slot's previous readers are known to be done (see above)
        f32x4 sq = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bool sum_due = HD && WO && (S || (t + RL >= 0 && t + RL < T)) && wave < NTW;
        if (sum_due) {
            const float* src = qred + (t & 1) * (NW * NTW * 256) + (wave * 64 + lane) * 4;
            sq = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
            for (int w = 1; w < NW; ++w) sq += *reinterpret_cast<const f32x4*>(src + w * NTW * 256);
        }
        // ---- (A) the partial tiles of step t+1 addressed to this workgroup (gather issued during step t+1)
        {
            f32x4 sr = (f32x4){0.f, 0.f, 0.f, 0.f};
            // (a group with a down product keeps the P exchange going through its drain, t < 0: nothing reads those tiles, but
            //  their hand-off is what orders the down product's loads behind the other workgroups' stores -- see above)
            if (S || (t + 1 < T && (HD ? t >= t_last : t >= 0))) sr = settle_total(rp, gp, (t + 1) & 1, parity(t + 1));
            *reinterpret_cast<f32x4*>(&red_r[wave][lane * 4]) = sr;
        }
        if (sum_due) store_q(sq, t + RL, wave);
        BSTAMP(1);
        FLOW2_BARRIER();                                                         // B1: red_r (and qred of the previous step) complete
        BSTAMP(2);
        if (epi) {
            if (S || t >= 0) {
                float dh = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) dh += red_r[w][e];
#if defined(AMDSPEECH_DEVTRACE) && AMDSPEECH_DEVTRACE == 3      // dev: the gathered recurrent part of dh, [L][T][B][H]
                if (a.trace != nullptr && pok)
                    reinterpret_cast<float*>(a.trace)[(((size_t)l * T + t) * B + b) * H + unit] = dh;
#endif
#if FLOW2_PRE_EPI
                float dz = pf.dz;
                if constexpr (CF) {
                    if (!top_ready) dz = !pok ? 0.0f : (__float_as_uint(dz) != FLOW_SENTINEL ? dz
                                                   : poll_dx(up_base + (size_t)t * up_step + (size_t)b * H + unit)
                                                         * zmult(a.drop, l + 1, (uint32_t)((size_t)t * B * H + bec)));
                } else
                if (!top_ready) dz = !pok ? 0.0f : (__float_as_uint(dz) != FLOW_SENTINEL ? dz
                                               : poll_dx(a.dxh + ((size_t)l * T + t) * bph + (size_t)b * H + unit)
                                                     * zmult(a.drop, l + 1, (uint32_t)((size_t)t * B * H + bec)));
                dh += dz;
                const float dct = dcin + dh * pf.a;      // (a finished or padded row: all factors 0, dcin stays 0)
                float4 dgv;
                dgv.x = dct * pf.bx; dgv.y = dct * pf.by; dgv.z = dct * pf.bz; dgv.w = dh * pf.bw;
                *reinterpret_cast<float4*>(a_lds + (t & 1) * 1024 + a_slot) = dgv;
                if (Q > 1 && rec_on)       // ... and to the partners of the K slice (no SGPR soffset: see store_tiles)
                    __builtin_amdgcn_raw_buffer_store_b128(flow_tag((f32x4){dgv.x, dgv.y, dgv.z, dgv.w}, parity(t)), rx,
                                                           x_store_off + (unsigned)(t & 1) * (unsigned)(NU * 4096), 0, FLOW2_STORE_AUX);
                dcin = dct * pf.gf;
#else
                Stash st;
                {
                    const int i = threadIdx.x;
                    st.gi = stash_lds[0][i]; st.gj = stash_lds[1][i]; st.gf = stash_lds[2][i]; st.go = stash_lds[3][i];
                    st.c = stash_lds[4][i]; st.cp = stash_lds[5][i]; st.dtop = stash_lds[6][i];
                }
                const float dx_pre = stash_lds[7][threadIdx.x];
                float dup = st.dtop;
                if (!top) dup = !pok ? 0.0f : (__float_as_uint(dx_pre) != FLOW_SENTINEL ? dx_pre
                                                : poll_dx(a.dxh + ((size_t)l * T + t) * bph + (size_t)b * H + unit));
                else if (CF && !top_ready && pok && __float_as_uint(dup) == FLOW_SENTINEL) dup = poll_dx(a.dztop + ((size_t)t * B + b) * H + unit);
                dh += dup * zmult(a.drop, l + 1, (uint32_t)((size_t)t * B * H + bec));
                const bool live = pok && t < len;
                const float tc = ftanh(st.c);
                const float dct = dcin + dh * st.go * (1.0f - tc * tc);
                float4 dgv;
                dgv.x = dct * st.gj * st.gi * (1.0f - st.gi);
                dgv.y = dct * st.gi * (1.0f - st.gj * st.gj);
                dgv.z = dct * st.cp * st.gf * (1.0f - st.gf);
                dgv.w = dh * tc * st.go * (1.0f - st.go);
                float dcout = dct * st.gf;
                if (!live) { dgv = make_float4(0.f, 0.f, 0.f, 0.f); dcout = 0.0f; }
                *reinterpret_cast<float4*>(a_lds + (t & 1) * 1024 + a_slot) = dgv;   // the whole hand-off of this step: 16 bytes to LDS
                if (Q > 1 && rec_on)
                    __builtin_amdgcn_raw_buffer_store_b128(flow_tag((f32x4){dgv.x, dgv.y, dgv.z, dgv.w}, parity(t)), rx,
                                                           x_store_off + (unsigned)(t & 1) * (unsigned)(NU * 4096), 0, FLOW2_STORE_AUX);
                dcin = dcout;
#endif
            } else if (Q > 1 && rec_on) {
                // the drain of a group with a down product (t < 0): no epilogue, but the partner still waits for this tile -- the pair's
                // exchange is a link of the chain that orders the down product's loads, like the P hand-off it feeds
                __builtin_amdgcn_raw_buffer_store_b128(flow_tag((f32x4){0.f, 0.f, 0.f, 0.f}, parity(t)), rx,
                                                       x_store_off + (unsigned)(t & 1) * (unsigned)(NU * 4096), 0, FLOW2_STORE_AUX);
            }
        } else {
            if (FLOW2_WINDOW != 1) rest_of_window(t, hd_tag, steady_tag);
            // Every workgroup of this group has passed B1(t+1) when we have gathered its P[t+1]; its row-major dG[t+3] store
            // (issued between B1(t+2) and B2(t+2), in front of loads it has since waited for) is in memory by then.
            if (l == 0 && ub == 0 && threadIdx.x == 256 && t >= 0 && t + 3 < T)
                __hip_atomic_store(FLOW_G(int, a.progress) + mb, t + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        BSTAMP(3);
        FLOW2_BARRIER();                                                         // B2: the dG tile of step t is in LDS
        BSTAMP(4);
        if (FLOW2_WINDOW == 1 && !epi) rest_of_window(t, hd_tag, steady_tag);      // (beside waves 0-3's rec MFMAs: these waves could not issue one yet)
        // ---- issued first, consumed last: the down product's operand (dG[t+3], this wave's producer) and, for the window of the
        // NEXT step, this element of the KS down tiles of frame t+6.  By ALL waves although only waves 4-7 use the second: with the
        // same memory operations in every wave hipcc's wait counts are exact, otherwise it takes the minimum over the two paths.
        if (HD) {
            if (DL == 3 && (S || (t + 3 >= 0 && t + 3 < T))) load_av2(t + 3);
            if (S || (t + GL >= 0 && t + GL < T)) {
                // (said to be wave-uniform explicitly: strength reduction turns the slot offset into a VGPR recurrence, and a
                //  VGPR in the scalar offset makes every load a waterfall loop)
                const unsigned so = uni((unsigned)((t + GL) & 3) * QSLOT_BYTES);
#pragma unroll
                for (int k = 0; k < KS; ++k) gq[k] = FLOW2_LDF(rq, q_load_off + (unsigned)(k * 1024), so, FLOW2_LOAD_AUX);
            }
        }
        // the next epilogue's stash: in flight under the MFMAs.  Issued by ALL waves although only waves 4-7 hand it on (8 KiB of
        // loads per step wasted): see above
        if (S || t > 0) fetch_stash(t - 1);
        f32x4 acc[NTW];
        f32x4 av[4];
        if (S || HD || t >= 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<const f32x4*>(a_lds + (t & 1) * 1024 + (m * 64 + lane) * 4);
        }
        // (the machine scheduler otherwise sinks the stash loads into the MFMA stream and -- worse -- hoists a third of the down
        //  MFMAs above the P stores: THE hand-off of the step left 0.5 us late; measured 6.2 instead of 5.6 us per step)
        __builtin_amdgcn_sched_barrier(0);
        // ---- rec product: dh partials of step t for every workgroup of the group
        if (rec_on) {
#pragma unroll
            for (int n = 0; n < NTW; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (BF3) {
                u32x4_f ah[2], al[2];              // the dG tile's two 32-wide K blocks as bf16 hi / lo
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const float x[8] = {av[0][2 * sp], av[1][2 * sp], av[2][2 * sp], av[3][2 * sp],
                                        av[0][2 * sp + 1], av[1][2 * sp + 1], av[2][2 * sp + 1], av[3][2 * sp + 1]};
                    flow_bf3_split(x, ah[sp], al[sp]);
                }
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                    for (int n = 0; n < NTW; ++n) acc[n] = flow_bf_mma<PR>(acc[n], ah[sp], al[sp], wrh[n][sp], wrl[n][sp]);
            } else if constexpr (Q > 1) {
                // own tile x this workgroup's part of the outputs; the partners' tiles are requested part-way (their ~1 us through the L2
                // lies under the own-tile MFMAs of BOTH wave sets -- they take the matrix pipe one after the other -- and, at Q = 4, under
                // the first tile of the down product), copied into LDS by waves 0-3 -- tags checked, re-loaded until they match -- and
                // read by everybody behind B3
                u32x4_f xv[Q - 1];
#pragma unroll
                for (int j = 0; j < Q - 1; ++j) xv[j] = (u32x4_f){0u, 0u, 0u, 0u};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g == FLOW2_XLOAD_AT) {
                        __builtin_amdgcn_sched_barrier(0);
                        x_issue(t & 1, xv);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int n = 0; n < NP; ++n) {
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][g], wr[n][g][0], acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][g], wr[n][g][1], acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2][g], wr[n][g][2], acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3][g], wr[n][g][3], acc[n], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (FLOW2_XLOAD_AT >= 4) x_issue(t & 1, xv);
                if constexpr (DF > 0) {
                    // the first DF tiles of the down product of frame t + DL (steady state: the frame exists), complete -- they go straight
                    // to their place in qred; the rest follows behind the P stores
#pragma unroll
                    for (int n = 0; n < DF; ++n) {
                        f32x4 ad = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            ad = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av2[g][0]), wd[n][g][0], ad, 0, 0, 0);
                            ad = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av2[g][1]), wd[n][g][1], ad, 0, 0, 0);
                            ad = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av2[g][2]), wd[n][g][2], ad, 0, 0, 0);
                            ad = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av2[g][3]), wd[n][g][3], ad, 0, 0, 0);
                        }
                        *reinterpret_cast<f32x4*>(qred + (WO ? (t & 1) * (NW * NTW * 256) : 0) + ((wave * NTW + n) * 64 + lane) * 4) = ad;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (epi) {
                    const unsigned xpar = parity(t);
                    bool stale = false;
#pragma unroll
                    for (int j = 0; j < Q - 1; ++j) stale = stale || flow_untagged(xv[j], xpar);
                    if (__any(stale) && !dead) {
                        while (true) {
                            if (wall_clock64() - t_begin > a.limit) { dead = true; if (lane == 0) __hip_atomic_fetch_or(FLOW_G(unsigned, a.err), 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                            x_issue(t & 1, xv);
                            stale = false;
#pragma unroll
                            for (int j = 0; j < Q - 1; ++j) stale = stale || flow_untagged(xv[j], xpar);
                            if (!__any(stale)) break;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < Q - 1; ++j) *reinterpret_cast<u32x4_f*>(x_lds + j * 1024 + (threadIdx.x & 255) * 4) = xv[j];
                }
                lds_barrier();                                                        // B3: the partners' tiles are in LDS
#pragma unroll
                for (int j = 1; j < Q; ++j) {
                    f32x4 ax[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) ax[m] = *reinterpret_cast<const f32x4*>(x_lds + (j - 1) * 1024 + (m * 64 + lane) * 4);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int n = 0; n < NP; ++n) {
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[0][g], wr[j * NP + n][g][0], acc[n], 0, 0, 0);
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[1][g], wr[j * NP + n][g][1], acc[n], 0, 0, 0);
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[2][g], wr[j * NP + n][g][2], acc[n], 0, 0, 0);
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[3][g], wr[j * NP + n][g][3], acc[n], 0, 0, 0);
                        }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int n = 0; n < NTW; ++n) {
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][g], wr[n][g][0], acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][g], wr[n][g][1], acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2][g], wr[n][g][2], acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3][g], wr[n][g][3], acc[n], 0, 0, 0);
                    }
            }
            BSTAMP(5);
            __builtin_amdgcn_sched_barrier(0);
            store_tiles(rp, acc, t & 1, parity(t));
            __builtin_amdgcn_sched_barrier(0);
        }
        BSTAMP(6);
        // ---- down product of frame t+DL; the gather of P[t] (the next step's operand) goes out part-way through it: the
        // hand-off (~1 us through this XCD's L2) lands under the remaining MFMAs
        if (HD && (S || (t + DL >= 0 && t + DL < T))) {
#if FLOW2_CHECK_ORDER
            {   // dev: the host pre-filled this layer's dG with the sentinel
                bool pending = false;
#pragma unroll
                for (int g = 0; g < 4; ++g) pending = pending || flow_pending(av2[g]);
                if (pending) __hip_atomic_fetch_or(FLOW_G(unsigned, a.err), 16u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#endif
#pragma unroll
            for (int n = 0; n < NTW; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (BF3) {
                u32x4_f ah[2], al[2];
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const float x[8] = {__uint_as_float(av2[2 * sp][0]), __uint_as_float(av2[2 * sp][1]), __uint_as_float(av2[2 * sp][2]),
                                        __uint_as_float(av2[2 * sp][3]), __uint_as_float(av2[2 * sp + 1][0]), __uint_as_float(av2[2 * sp + 1][1]),
                                        __uint_as_float(av2[2 * sp + 1][2]), __uint_as_float(av2[2 * sp + 1][3])};
                    flow_bf3_split(x, ah[sp], al[sp]);
                }
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
#pragma unroll
                    for (int n = 0; n < NTW; ++n) acc[n] = flow_bf_mma<PR>(acc[n], ah[sp], al[sp], wdh[n][sp], wdl[n][sp]);
                    if (sp == 0 && rec_on) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue(rp, gp, t & 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g == FLOW2_GATHER_AT && rec_on) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue(rp, gp, t & 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int n = DF; n < NTW; ++n) {      // (the first DF tiles were formed inside the rec phase)
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av2[g][0]), wd[n][g][0], acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av2[g][1]), wd[n][g][1], acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av2[g][2]), wd[n][g][2], acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av2[g][3]), wd[n][g][3], acc[n], 0, 0, 0);
                    }
                }
                if (FLOW2_GATHER_AT >= 4 && rec_on) issue(rp, gp, t & 1);
            }
#pragma unroll
            for (int n = DF; n < NTW; ++n)
                *reinterpret_cast<f32x4*>(qred + (WO ? (t & 1) * (NW * NTW * 256) : 0) + ((wave * NTW + n) * 64 + lane) * 4) = acc[n];
        } else if (rec_on) {
            issue(rp, gp, t & 1);                                                // no down product (this step): nothing to hide it under
        }
        // the NEXT step's down operand, into the registers this step's product has just released: a whole step of flight time
        // (the rows were stored write-through: they may have to come back from memory)
        if (HD && DL == 4 && (S || (t + 3 >= 0 && t + 3 < T))) load_av2(t + 3);
        BSTAMP(7);
#if FLOW2_PRE_EPI == 1
        if (epi && (S || t > 0)) precompute(t - 1);
#elif FLOW2_PRE_EPI == 2
#else
        if (!epi && (S || t > 0)) publish_stash();                               // read by the epilogue after the next B1
#endif
    };
    auto run = [&](auto hd_tag) __attribute__((always_inline)) {
        int t = T - 1;
        for (; t >= t_last && t > T - (XL + 1); --t) step(t, hd_tag, std::false_type{});      // the first frames: not every neighbour exists
        // (once, so that nothing the general body left in flight -- in whatever registers ITS allocation chose -- is "pending"
        //  at the steady loop's header: hipcc would guard the first use of each such register with s_waitcnt vmcnt(0) on every trip)
        FLOW_WEIGHTS_RESIDENT();
        for (; t >= 1; --t) step(t, hd_tag, std::true_type{});                         // steady state
        for (; t >= t_last; --t) step(t, hd_tag, std::false_type{});                   // frame 0 and the drain
    };
    if (has_down) run(std::true_type{});
    else run(std::false_type{});
#undef BSTAMP
#if defined(AMDSPEECH_DEVTRACE) && AMDSPEECH_DEVTRACE == 9      // (the knobs-only development build: tools/kernel_clocks.py)
    if (a.trace != nullptr && grp == 0 && ub == 0 && threadIdx.x == 0) {      // (see lstm_fwd_flow2)
        a.trace[2] = __builtin_readcyclecounter() - c_begin;
        a.trace[3] = wall_clock64() - t_begin;
    }
#endif
}


