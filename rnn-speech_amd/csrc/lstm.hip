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
#include "gemm_core.h"
#include "ctc_core.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <mutex>
#include <unordered_map>

namespace amdspeech {

// s_waitcnt vmcnt(0) (expcnt / lgkmcnt untouched) in a form the compiler's own wait-count bookkeeping sees.  The whole-sequence
// kernels load their weight fragments once, in front of the time loop; without this in front of the loop hipcc merges "weight
// loads still pending" into the loop header and guards the first use of every weight register INSIDE the loop with a ladder of
// s_waitcnt vmcnt(n) ... vmcnt(0) in the middle of the MFMA stream, which at run time waits for whatever the wave has in flight
// then (in lstm_bwd_big: the write-through store of the row-major dG tile it has just issued).
#define FLOW_WEIGHTS_RESIDENT() __builtin_amdgcn_s_waitcnt(0x0F70)
// In-kernel wall-clock stamps / debug taps (tools/trace_*.py) write through a device pointer the TOOL hands over in
// AMDSPEECH_TRACE_PTR: development builds (-DAMDSPEECH_DEVTRACE) only -- a release library never takes an address from the
// environment.
static unsigned long long* dev_trace_ptr() {
#ifdef AMDSPEECH_DEVTRACE
    if (const char* e = dev_knob_str("AMDSPEECH_TRACE_PTR")) return reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0));
#endif
    return nullptr;
}
#ifndef BIG_WEIGHTS_RESIDENT
#define BIG_WEIGHTS_RESIDENT 1    // (dev: 0 = the H = 1024 kernels without it)
#endif

// ------------------------------------------------------------------ workspace
struct LstmLayout {
    size_t wp, wq, z, hs, cs, gates, dg, dztop, dz0, dc, xp0, xp, hp, dgp, sync, xph, hph, dxh, prec, pdown, xwp, bigring, wopack, bfs, total;  // float offsets
    size_t fwd_set = 0;     // distance (floats) between the two sets of forward panels {xph, hph}
};

// The dataflow ("flow") kernels keep a workgroup's weight slice on chip for the whole sequence and place one
// recurrence group (layer, 16-row batch tile) per XCD: H a multiple of 128 up to 512, at most 8 groups.
static bool flow_shape_ok(const amdspeech_lstm_desc* d) {
    // (the kernels address one layer's [T][B][4H] gradients through a 32-bit buffer resource)
    // (split precision pairs K blocks: H a multiple of 256 there)
    return (d->precision == 0 || ((d->precision == 1 || d->precision == 2) && d->H % 256 == 0)) && d->H % 128 == 0 && d->H <= 512 && (long)d->L * ((d->B + 15) / 16) <= 8 &&
           (size_t)d->T * ((d->B + 15) / 16 * 16) * 4 * d->H * 4 < (1ull << 32);
}
// x-product workers of the forward dataflow kernel (fwd_x_worker): the largest number of K blocks per recurrence wave they can
// take at this shape -- exact f32 at H = 512 (four K blocks per wave and half), at least one XCD without a recurrence group, and
// the tile history addressable through one 32-bit buffer resource.  lstm_fwd picks the number it uses (<= this) at the launch.
#ifndef FWD2_WORKER_PARTS
#define FWD2_WORKER_PARTS 1        // K blocks per recurrence wave the workers take (2 is built and measured slower: DESIGN.md 4.2)
#endif
static int fwd_workers_max(const amdspeech_lstm_desc* d) {
    const long groups = (long)d->L * ((d->B + 15) / 16);
    if (d->precision != 0 || d->H != 512 || groups >= 8) return 0;
    for (int mv = FWD2_WORKER_PARTS; mv > 0; --mv)
        if ((size_t)d->T * groups * (d->H / 16) * mv * 4096 < (1ull << 32)) return mv;
    return 0;
}
// ---- the batched products of the H = 1024 path through bf16 copies (precision = 2; gemm_bf16p.hip) ----------------------------------
// One region of the workspace: Z as bf16 [TB][H] (x . W_ih), W_ih^T [4H][H]; dG as bf16 [TB][4H] and W_ih [H][4H] (dX);
// [Z ; Hprev]^T [2H][TB] and dG^T [4H][TB] (dK, both halves of a layer's kernel gradient as ONE product); the partial tiles of dK.
// The region is RESERVED for every sequence length of the shape (ops.LstmWorkspace lays ONE allocation out for the longest sequence and
// re-lays it out per mini-batch with a shorter T: a layout's size has to be monotone in T) and USED by the calls whose row count the
// 64 x 64 transposing copies take; the others fall back to gemm_bf16 inside the same layout.
static bool bf16p_layout_reserved(const amdspeech_lstm_desc* d) {
    static const int env = runtime_switch("AMDSPEECH_BF16_PACKED", 1);      // 0: gemm_bf16 (f32 operands converted on the way into LDS: round 4)
    return env != 0 && d->precision == 2 && d->H == 1024;
}
static bool bf16p_layout_on(const amdspeech_lstm_desc* d) {
    return bf16p_layout_reserved(d) && ((long)d->T * d->B) % 64 == 0 && (long)d->T * d->B >= 256;
}
// (the split-K partial tiles of whichever of the three batched products of a layer needs most: short runs split the x / dX products too.
//  bf16p_splits never makes more than 256 partial tiles of 256 x 256 floats, whatever the row count -- a T-independent bound, so that
//  the region's size stays monotone in T)
static size_t bf16p_partial_need(size_t TB, size_t H) {
    const size_t a = bf16p_partial_bytes(2 * (int)H, 4 * (int)H, (int)TB), b = bf16p_partial_bytes((int)TB, 4 * (int)H, (int)H),
                 c = bf16p_partial_bytes((int)TB, (int)H, 4 * (int)H);
    const size_t need = a > b ? (a > c ? a : c) : (b > c ? b : c), bound = (size_t)256 * 256 * 256 * sizeof(float);
    return need > bound ? need : bound;
}
struct Bf16pBufs { unsigned short *zb, *wtb, *dgb, *wb, *zht, *dgt; char* partial; size_t partial_bytes; };
static size_t bf16p_scratch_floats(const amdspeech_lstm_desc* d) {
    const size_t TB = ((size_t)d->T * d->B + 63) / 64 * 64, H = d->H;      // (reserved for every T: see bf16p_layout_reserved)
    const size_t bytes = TB * H * 2 + 4 * H * H * 2 + TB * 4 * H * 2 + H * 4 * H * 2 + 2 * H * TB * 2 + 4 * H * TB * 2 +
                         bf16p_partial_need(TB, H) + 8 * 256;
    return (bytes + 3) / 4;
}
static Bf16pBufs bf16p_bufs(const amdspeech_lstm_desc* d, float* base) {
    const size_t TB = (size_t)d->T * d->B, H = d->H;
    char* p = reinterpret_cast<char*>(base);
    auto take = [&](size_t bytes) { char* r = p; p += align_up(bytes, 256); return r; };
    Bf16pBufs b;
    b.zb = reinterpret_cast<unsigned short*>(take(TB * H * 2));
    b.wtb = reinterpret_cast<unsigned short*>(take(4 * H * H * 2));
    b.dgb = reinterpret_cast<unsigned short*>(take(TB * 4 * H * 2));
    b.wb = reinterpret_cast<unsigned short*>(take(H * 4 * H * 2));
    b.zht = reinterpret_cast<unsigned short*>(take(2 * H * TB * 2));
    b.dgt = reinterpret_cast<unsigned short*>(take(4 * H * TB * 2));
    b.partial_bytes = bf16p_partial_need(TB, H);
    b.partial = take(b.partial_bytes);
    return b;
}
// G[rows][4H] = Z[rows][H] . K[0:H, :] + bias
static int bf16p_xw(hipStream_t s, const Bf16pBufs& b, int rows, int H, const float* Z, const float* K, float* G, const float* bias) {
    if (int rc = bf16p_copy(s, Z, H, rows, H, false, b.zb, H, nullptr)) return rc;
    if (int rc = bf16p_copy(s, K, 4 * H, H, 4 * H, true, b.wtb, H, nullptr)) return rc;              // [H][4H] -> [4H][H]
    return bf16p_gemm(s, rows, 4 * H, H, b.zb, H, b.wtb, H, G, 4 * H, bias, false, b.partial, b.partial_bytes);
}
// dX[rows][H] = dG[rows][4H] . K[0:H, :]^T
static int bf16p_dx(hipStream_t s, const Bf16pBufs& b, int rows, int H, const float* dG, const float* K, float* dX) {
    if (int rc = bf16p_copy(s, dG, 4 * H, rows, 4 * H, false, b.dgb, 4 * H, nullptr)) return rc;
    if (int rc = bf16p_copy(s, K, 4 * H, H, 4 * H, false, b.wb, 4 * H, nullptr)) return rc;
    return bf16p_gemm(s, rows, H, 4 * H, b.dgb, 4 * H, b.wb, 4 * H, dX, H, nullptr, false, b.partial, b.partial_bytes);
}
// A whole layer's batched backward products behind its recurrence launch (all T x B rows): ONE read of dG gives its row-major
// copy (dX), its transposed copy (dK) and the bias gradient; dX[rows][H] = dG . K[0:H, :]^T; dK[2H][4H] += [Z ; Hprev]^T . dG
static int bf16p_layer_bwd(hipStream_t s, const Bf16pBufs& b, int rows, int H, const float* Z, const float* Hp, const float* dG, const float* K,
                           float* dX, float* dK, float* dbias) {
    if (int rc = bf16p_copy(s, dG, 4 * H, rows, 4 * H, true, b.dgt, rows, dbias, b.dgb)) return rc;
    if (int rc = bf16p_copy(s, K, 4 * H, H, 4 * H, false, b.wb, 4 * H, nullptr)) return rc;
    if (int rc = bf16p_gemm(s, rows, H, 4 * H, b.dgb, 4 * H, b.wb, 4 * H, dX, H, nullptr, false, b.partial, b.partial_bytes)) return rc;
    if (int rc = bf16p_copy(s, Z, H, rows, H, true, b.zht, rows, nullptr)) return rc;
    if (int rc = bf16p_copy(s, Hp, H, rows, H, true, b.zht + (size_t)H * rows, rows, nullptr)) return rc;
    return bf16p_gemm(s, 2 * H, 4 * H, rows, b.zht, rows, b.dgt, rows, dK, 4 * H, nullptr, true, b.partial, b.partial_bytes);
}
// ... when the recurrence kernel has written dG's row-major bf16 copy and its column sums itself (lstm_bwd_big1)
static int bf16p_layer_bwd_copied(hipStream_t s, const Bf16pBufs& b, int rows, int H, const float* Z, const float* Hp, const float* K,
                                  float* dX, float* dK) {
    if (int rc = bf16p_copy(s, K, 4 * H, H, 4 * H, false, b.wb, 4 * H, nullptr)) return rc;
    if (int rc = bf16p_gemm(s, rows, H, 4 * H, b.dgb, 4 * H, b.wb, 4 * H, dX, H, nullptr, false, b.partial, b.partial_bytes)) return rc;
    if (int rc = bf16p_copy(s, Z, H, rows, H, true, b.zht, rows, nullptr)) return rc;
    if (int rc = bf16p_copy(s, Hp, H, rows, H, true, b.zht + (size_t)H * rows, rows, nullptr)) return rc;
    if (int rc = bf16p_transpose(s, b.dgb, rows, 4 * H, b.dgt, rows)) return rc;
    return bf16p_gemm(s, 2 * H, 4 * H, rows, b.zht, rows, b.dgt, rows, dK, 4 * H, nullptr, true, b.partial, b.partial_bytes);
}
// dK[2H][4H] += [Z ; Hprev]^T . dG over `rows` frames x batch rows (a multiple of 64); dbias[4H] += column sums of dG
static int bf16p_dk(hipStream_t s, const Bf16pBufs& b, int rows, int H, const float* Z, const float* Hp, const float* dG, float* dK, float* dbias) {
    if (int rc = bf16p_copy(s, Z, H, rows, H, true, b.zht, rows, nullptr)) return rc;
    if (int rc = bf16p_copy(s, Hp, H, rows, H, true, b.zht + (size_t)H * rows, rows, nullptr)) return rc;
    if (int rc = bf16p_copy(s, dG, 4 * H, rows, 4 * H, true, b.dgt, rows, dbias)) return rc;
    return bf16p_gemm(s, 2 * H, 4 * H, rows, b.zht, rows, b.dgt, rows, dK, 4 * H, nullptr, true, b.partial, b.partial_bytes);
}

static LstmLayout lstm_layout(const amdspeech_lstm_desc* d) {
    const size_t T = d->T, B = d->B, H = d->H, L = d->L;
    const size_t tbh = T * B * H;
    LstmLayout o;
    size_t off = 0;
    auto take = [&](size_t n) { size_t r = off; off += (n + 63) / 64 * 64; return r; };
    // lstm_fwd_flow2's x-product workers: pre-multiplied gate tiles, [T][L][batch tiles][H/16][parts][256][4], written once per
    // launch and tagged with the launch's parity.  FIRST and time-major: frame t lives at the same address whatever T the
    // descriptor names (ops.LstmWorkspace.prefix lays ONE allocation out for every sequence length of a training run), so the
    // tags survive from one launch to the next with another T (AMDSPEECH_LSTM_SAME_WS)
    o.xwp = 0;
    if (flow_shape_ok(d) && fwd_workers_max(d) > 0) o.xwp = take(T * L * ((B + 15) / 16) * (H / 16) * fwd_workers_max(d) * 1024);
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
    // fragment-major ("packed") copies of the panels the NEXT diagonal consumes as MFMA A operands
    const size_t bp = (B + 15) / 16 * 16;
    o.xp0 = take(T * bp * H);          // layer-0 input, whole sequence
    o.xp = take(L * 2 * bp * H);       // layer l>=1 input, 2-slot ring (slot = diagonal parity)
    o.hp = take(L * 2 * bp * H);       // h_{t-1}, 2-slot ring
    o.dgp = take(L * 2 * bp * 4 * H);  // dG, 2-slot ring
    o.sync = take(64);                 // error word of the dataflow kernels, backward progress word, XCD tickets
    // full-history fragment-major panels of the dataflow kernels (every slot written once per sequence)
    o.xph = o.hph = o.dxh = o.prec = o.pdown = o.wopack = off;
    if (flow_shape_ok(d)) {
        o.xph = take((L + 1) * T * bp * H);    // layer l >= 1 input x_t  (slot [l][t]; [0][*] unused; [L][*]: the top layer's output for the fused CTC head)
        o.hph = take(L * (T + 1) * bp * H);    // h_{t-1}                  (slot [l][t]; [l][0] = initial state)
        // a SECOND set of the two (AMDSPEECH_LSTM_ARM_NEXT): a training cycle's forward calls alternate between the sets, and the
        // set the next call will use gets its sentinels beside THIS call's kernel -- not behind it, where the 330 MB fill met the
        // output layer and the log-softmax
        o.fwd_set = off - o.xph;
        take((L + 1) * T * bp * H);
        take(L * (T + 1) * bp * H);
        o.wopack = take((H / 16) * CF_NTC * 256);      // W_o as MFMA B fragments (fused CTC head)
        o.dxh = take(L * T * bp * H);          // dX_l[t]: gradient of layer l's output coming from layer l+1 (through memory)
        // lstm_bwd_flow2: partial-tile rings, [group][slots][H/16 consumers][H/16 producers][256 floats]
        const size_t slot = (size_t)L * (bp / 16) * (H / 16) * (H / 16) * 256;
        o.prec = take(2 * slot);               // rec partials: 2 slots
        o.pdown = take(4 * L * (bp / 16) * (H / 16) * (H / 128) * 256);   // down partials, summed per K slice: 4 slots of [H/16 consumers][H/128 K slices][256]
        // ... and, directly behind them (the kernel finds it there), the dG tiles the two workgroups of a pair show each other when the
        // recurrent product is cut both ways (FLOW2_Q = 2): [group][2 slots][H/16][1024], tagged; zeroed with the rings
        take(L * (bp / 16) * 2 * (H / 16) * 1024);
    }
    // lstm_bwd_big (H = 1024), ONE layer at a time: the partial-tile rings of the two XCDs of every pair, [2 slots][batch tiles]
    // [2][32][32][256 floats], and the dG tiles that cross between them, [2 slots][batch tiles][64][1024]
    o.bigring = off;
    if (!flow_shape_ok(d) && d->precision >= 0 && d->precision <= 2 && d->H == 1024 && bp / 16 <= 4)
        o.bigring = take((size_t)2 * (bp / 16) * (2 * 32 * 32 * 256 + 64 * 1024));
    // precision = 2 at H = 1024 (gemm_bf16p.hip): bf16 copies of the batched products' operands + the split-K partial tiles
    o.bfs = off;
    if (bf16p_layout_reserved(d)) o.bfs = take(bf16p_scratch_floats(d));
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
    AS_CHECK_ARG(d->precision == 0 || ((d->precision == 1 || d->precision == 2) && d->H % 32 == 0),
                 "lstm: precision %d unsupported (0 = f32; 1 = bf16x3, 2 = bf16: both need H %% 32 == 0, H = %d)", d->precision, d->H);
    AS_CHECK_ARG((d->flags & ~(AMDSPEECH_LSTM_ARMED | AMDSPEECH_LSTM_ARM_NEXT | AMDSPEECH_LSTM_SAME_WS | AMDSPEECH_LSTM_PER_DIAGONAL |
                               AMDSPEECH_LSTM_INJECT_TIMEOUT)) == 0, "lstm: unknown flags 0x%x", d->flags);
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
// grouped != 0 (persistent kernel): every N tile holds all four gates of 4 units instead,
//   unit u = nt*4 + j%4, gate g = j/4.
__global__ void pack_fwd_kernel(const float* __restrict__ kernels, long kstride, float* __restrict__ wp,
                                int H, int L, int UW, int grouped) {
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
    if (grouped) { g = j >> 2; u = nt * 4 + (j & 3); }
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

// Fragment-major layout of a [rows, K] panel (rows padded to 16): tile (mt = row/16, kb = k/16)
// is one 1 KiB block ordered [lane][m] with lane = ((k/4)%4)*16 + row%16, m = k%4 -- exactly the
// v_mfma_f32_16x16x4_f32 A operand of four consecutive MFMAs, so a wave reads it with ONE fully
// coalesced float4 load instead of touching 16 rows.
__device__ __forceinline__ size_t packed_off(int row, int k, int K) {
    return ((((size_t)(row >> 4) * (K >> 4) + (k >> 4)) * 64) + (((k >> 2) & 3) * 16 + (row & 15))) * 4 + (k & 3);
}

// src: nmat row-major [B][K] panels (stride src_stride) -> dst: nmat packed panels (stride bp*K)
__global__ void pack_rows_kernel(const float* __restrict__ src, size_t src_stride, float* __restrict__ dst,
                                 int B, int K, int nmat) {
    const size_t per = (size_t)B * K;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per * nmat) return;
    const int mat = i / per;
    const size_t r = i % per;
    const int row = r / K, k = r % K;
    const size_t bpk = (size_t)((B + 15) / 16 * 16) * K;
    dst[(size_t)mat * bpk + packed_off(row, k, K)] = src[(size_t)mat * src_stride + r];
}

// Layer-0 input of the dataflow forward kernel in one pass: Z_0 *= input-dropout multiplier (in place: the backward pass
// reads the masked Z_0) and the packed panels of all T frames.  One thread = four consecutive features of one row.
__global__ __launch_bounds__(256) void mask_pack_rows_kernel(float* __restrict__ z, float* __restrict__ dst, int B, int K, int T,
                                                             DropCfg c, int masked) {
    const size_t per4 = (size_t)B * K / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per4 * T) return;
    const int t = i / per4;
    const size_t r = (i % per4) * 4;
    const int row = r / K, k = r % K;
    const size_t e = (size_t)t * B * K + r;
    float4 v = *reinterpret_cast<const float4*>(z + e);
    if (masked) {
        v.x *= zmult(c, 0, (uint32_t)e); v.y *= zmult(c, 0, (uint32_t)(e + 1));
        v.z *= zmult(c, 0, (uint32_t)(e + 2)); v.w *= zmult(c, 0, (uint32_t)(e + 3));
        *reinterpret_cast<float4*>(z + e) = v;
    }
    const size_t bpk = (size_t)((B + 15) / 16 * 16) * K;
    *reinterpret_cast<float4*>(dst + (size_t)t * bpk + packed_off(row, k, K)) = v;
}

// Everything small the dataflow forward kernel needs before it starts, in one launch: the initial state rows hs[l][0] /
// cs[l][0] (given, or zeros), the packed h_{-1} panels (slot 0 of hph, padding rows zero), the error word and the tickets.
__global__ __launch_bounds__(256) void flow_fwd_prepare_kernel(const float* __restrict__ h0, const float* __restrict__ c0,
                                                               float* __restrict__ hs, float* __restrict__ cs,
                                                               float* __restrict__ hph, unsigned* __restrict__ sync_words,
                                                               int T, int B, int H, int L) {
    const int bp = (B + 15) / 16 * 16;
    const size_t per = (size_t)bp * H;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 40) sync_words[i] = 0u;              // error word (+ progress words), the per-XCD tickets
    if (i >= per * L) return;
    const int l = i / per;
    const size_t r = i % per;
    const int row = r / H, k = r % H;
    float hv = 0.f;
    if (row < B) {
        const size_t e = (size_t)row * H + k, bh = (size_t)B * H;
        hv = h0 ? h0[l * bh + e] : 0.f;
        hs[(size_t)l * (T + 1) * bh + e] = hv;
        cs[(size_t)l * (T + 1) * bh + e] = c0 ? c0[l * bh + e] : 0.f;
    }
    hph[(size_t)l * (T + 1) * per + packed_off(row, k, H)] = hv;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

#include "lstm_step_fwd.h"
#include "lstm_flow.h"
}  // namespace amdspeech
#include "ctc_flow.h"      // the CTC head inside the dataflow kernels (needs FLOW_SENTINEL / flow_pending above)
namespace amdspeech {

#include "lstm_flow_fwd.h"
#include "lstm_big_fwd.h"
#include "lstm_step_bwd.h"
#include "lstm_flow_bwd.h"
#include "lstm_big_bwd.h"
#include "lstm_step_bf3.h"
// ---------------------------------------------------------------- profiling
// HIP-event time of the recurrence kernels of the last call, per direction.  The per-layer paths (H = 1024) launch one kernel
// per layer with GEMMs in between: every kernel gets its own event pair (a "segment") and the reported time is their sum.
constexpr int PROF_SEGS = 16;
static bool g_prof_on = false;
static hipEvent_t g_prof_ev[2][PROF_SEGS][2];
static int g_prof_launches[2] = {0, 0};
static int g_prof_nseg[2] = {0, 0};
static bool g_prof_valid[2] = {false, false};
static double g_prof_flops[2][2] = {{0, 0}, {0, 0}};      // [which][0: recurrence products, 1: other products inside the same launches]
static void prof_flops(int which, double recurrence, double other) { g_prof_flops[which][0] = recurrence; g_prof_flops[which][1] = other; }

static void prof_begin(int which, hipStream_t s, int seg = 0) {
    if (g_prof_on && seg < PROF_SEGS) (void)hipEventRecord(g_prof_ev[which][seg][0], s);
}
static void prof_end(int which, hipStream_t s, int launches, int seg = 0) {
    if (!g_prof_on || seg >= PROF_SEGS) return;
    (void)hipEventRecord(g_prof_ev[which][seg][1], s);
    g_prof_launches[which] = launches;
    g_prof_nseg[which] = seg + 1;
    g_prof_valid[which] = true;
}

// Two independent launch chains (disjoint batch rows) on two streams: a single chain is bound by
// per-step latencies (kernel boundary, first-byte latency from MALL, weight re-fetch), so a second
// chain in flight fills the machine while the first one waits.
static hipStream_t g_side = nullptr;
static hipEvent_t g_fork = nullptr, g_join = nullptr;
static int side_stream_init() {
    if (g_side) return AMDSPEECH_OK;
    AS_CHECK_HIP(hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking));
    AS_CHECK_HIP(hipEventCreateWithFlags(&g_fork, hipEventDisableTiming));
    AS_CHECK_HIP(hipEventCreateWithFlags(&g_join, hipEventDisableTiming));
    return AMDSPEECH_OK;
}
// Weight-gradient GEMMs of finished time chunks run on the side stream UNDER the rest of the BPTT chain
// (the chain leaves 64 CUs idle and the MFMA pipes mostly free).  0 = off (everything after the chain).
// CU partition (hipExtStreamCreateWithCUMask; mask bit i = CU i/8 of XCD i%8 on this part, measured with
// tools/cumask_probe.hip): the chain gets 24 CUs of every XCD (its grids are 192 workgroups anyway), the GEMMs
// the other 8 -- un-partitioned, the MFMA-saturating GEMM waves share SIMDs with the chain's and make every
// diagonal 1.7x slower, which cancels the overlap.
static hipStream_t g_chain = nullptr, g_gemm = nullptr;
static hipEvent_t g_ev_a = nullptr, g_ev_b = nullptr, g_ev_c = nullptr;
static int g_overlap_state = 0;      // 0 = not tried, 1 = ready, -1 = unavailable on this device
static int overlap_init() {
    if (g_overlap_state != 0) return g_overlap_state;
    g_overlap_state = -1;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus != 256) return -1;
    uint32_t chain_mask[8] = {0, 0, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}, gemm_mask[8] = {~0u, ~0u, 0, 0, 0, 0, 0, 0};
    if (hipExtStreamCreateWithCUMask(&g_chain, 8, chain_mask) != hipSuccess) return -1;
    if (hipExtStreamCreateWithCUMask(&g_gemm, 8, gemm_mask) != hipSuccess) return -1;
    if (hipEventCreateWithFlags(&g_ev_a, hipEventDisableTiming) != hipSuccess) return -1;
    if (hipEventCreateWithFlags(&g_ev_b, hipEventDisableTiming) != hipSuccess) return -1;
    if (hipEventCreateWithFlags(&g_ev_c, hipEventDisableTiming) != hipSuccess) return -1;
    g_overlap_state = 1;
    return 1;
}
// AMDSPEECH_OVERLAP_DK = "chunks:side": the T axis is cut into `chunks` pieces; the first `side` of them (in the
// order the chain finishes them) run on the GEMM partition under the chain, the rest after it on the whole chip.
static void dk_overlap_plan(int* chunks, int* side) {
    static int c = -1, sd = 0;
    if (c < 0) {
        c = 8; sd = 5;
        if (const char* e = dev_knob_str("AMDSPEECH_OVERLAP_DK")) {
            c = atoi(e); sd = c - 1;
            if (const char* q = strchr(e, ':')) sd = atoi(q + 1);
        }
        if (c < 0) c = 0;
        if (c > 64) c = 64;
        if (sd > c - 1) sd = c - 1;
        if (sd < 0) sd = 0;
    }
    *chunks = c; *side = sd;
}
static int num_chains(int B) {
    static const int env = dev_knob("AMDSPEECH_CHAINS", 1);   // 2 measured no faster
    return (env >= 2 && B > 16) ? 2 : 1;
}

// --------------------------------------------------------------- host side
// precision = bf16x3 also covers the BATCHED products around the recurrence (round 3; gemm_bf3.hip): the hoisted x . W_ih and
// dX = dG . W_ih^T of the H = 1024 path, the weight gradients and dZ_0 of every path.  AMDSPEECH_BF3_GEMM=0: exact f32 there.
static int gemm_f32_plain(hipStream_t s, bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                          int ldc, const float* bias, bool accumulate) {
    return gemm_f32(s, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate);
}
static bool bf3_gemm(const amdspeech_lstm_desc* d) {      // the batched products in the descriptor's reduced precision
    static const int env = dev_knob("AMDSPEECH_BF3_GEMM", 1);
    return d->precision != 0 && env != 0;
}
// ... through the GEMM of that precision (1: three bf16 MFMAs per product, 2: one)
static int gemm_reduced(const amdspeech_lstm_desc* d, hipStream_t s, bool ta, bool tb, int M, int N, int K, const float* A, int lda,
                        const float* B, int ldb, float* C, int ldc, const float* bias, bool accumulate) {
    return (d->precision == 2 ? gemm_bf16 : gemm_bf3)(s, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate);
}
static int pick_uw(const amdspeech_lstm_desc* d) {
    if (const int uw = dev_knob("AMDSPEECH_UW", 0)) return uw;
    // 8 units (two 16-column N tiles) per workgroup halves the redundant re-reads of the
    // [B, 2H] activation panel; fall back to 4 when that would leave most CUs without work.
    const long wgs8 = (long)d->L * (d->H / 8) * ceil_div(d->B, 32);
    return (d->H % 8 == 0 && wgs8 >= 96) ? 8 : 4;
}

static int device_cus() {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    }
    return cus;
}
// Shapes whose weights do not fit on chip (H = 1024: 16 MB per layer and direction) run layer by layer with the
// time-independent half of every product HOISTED out of the recurrence into one big GEMM per layer: forward
// x.W_ih + b for all T frames (the launch-per-frame kernel then contracts only h_{t-1}.W_hh and adds the stored row),
// backward dX_{l-1} = dG_l.W_ih^T for all frames once layer l is done (the per-frame kernel keeps only the
// recurrent product).  Half the per-launch weight traffic and MFMA work, and the hoisted half runs at GEMM rate.
// AMDSPEECH_HOIST = bit mask (1 = forward, 2 = backward) overrides the default below.
// Measured (5x1024, B = 64, T = 998): the backward pass gains (147 -> 129 ms); the forward pass does not (82 ms either way: a
// launch per frame and layer costs what a launch per diagonal of five layers saved), and with ONE batch tile (3x1024, B = 10)
// tripling the launch count loses (179 -> 237 ms).  Returns bit 0 = forward, bit 1 = backward.
static int use_hoist(const amdspeech_lstm_desc* d, bool flow) {
    static const int env = dev_knob("AMDSPEECH_HOIST", -1);
    if (flow || d->precision != 0) return 0;
    if (env >= 0) return env & 3;
    return (d->H >= 768 && (d->B + 15) / 16 >= 2) ? 2 : 0;
}

// H = 1024 forward: one weight-stationary launch per layer (lstm_fwd_big); AMDSPEECH_BIG=0 turns it off
static bool use_big_fwd(const amdspeech_lstm_desc* d) {
    static const int env = runtime_switch("AMDSPEECH_BIG", 1);
    // (AMDSPEECH_LSTM_PER_DIAGONAL: the re-run of a mini-batch whose launch gave up waiting takes NO kernel with bounded waits)
    return env != 0 && !(d->flags & AMDSPEECH_LSTM_PER_DIAGONAL) && d->precision >= 0 && d->precision <= 2 && d->H == 1024 &&
           (d->B + 15) / 16 <= 4 && device_cus() == 256 &&
           (size_t)2 * ((d->B + 15) / 16 * 16) * d->H * 4 < (1ull << 32);
}

// ... in plain bf16 (precision 2) a batch tile's group fits ONE XCD (lstm_fwd_big1 / lstm_bwd_big1), and two stacks of one shape run
// side by side on the two halves of the chip (amdspeech_lstm_fwd_pair / _bwd_pair); AMDSPEECH_BIG1=0: one after the other on the XCD pairs
static bool use_big1_fwd(const amdspeech_lstm_desc* d) {
    static const int env = runtime_switch("AMDSPEECH_BIG1", 1);
    return env != 0 && d->precision == 2 && use_big_fwd(d);
}

// AMDSPEECH_FLOW=0 falls back to one launch per diagonal
static bool use_flow(const amdspeech_lstm_desc* d) {
    static const int env = runtime_switch("AMDSPEECH_FLOW", 1);
    // (8 XCDs x 32 CUs: the backward kernel places one recurrence group per XCD)
    // (AMDSPEECH_LSTM_PER_DIAGONAL: this call asks for the launch-per-diagonal kernels -- the re-run of a mini-batch whose dataflow
    //  launch timed out; the workspace layout does not depend on it)
    return env != 0 && !(d->flags & AMDSPEECH_LSTM_PER_DIAGONAL) && flow_shape_ok(d) && device_cus() == 256 && d->L * ((d->B + 15) / 16) <= 8;
}

static void (*flow_fwd_kernel(int H, int pr, int mv, bool cf))(FlowArgs) {      // (flow_shape_ok: reduced precision only at H = 256, 512)
    if (cf)       // with the fused CTC head's follower (any precision: the role does not depend on it)
        switch (H / 128) {
            case 1: return lstm_fwd_flow2<1, 0, 0, true>;
            case 2: return pr == 2 ? lstm_fwd_flow2<2, 2, 0, true> : (pr == 1 ? lstm_fwd_flow2<2, 1, 0, true> : lstm_fwd_flow2<2, 0, 0, true>);
            case 3: return lstm_fwd_flow2<3, 0, 0, true>;
            default:
                if (pr == 0 && mv == 2) return lstm_fwd_flow2<4, 0, 2, true>;
                if (pr == 0 && mv == 1) return lstm_fwd_flow2<4, 0, 1, true>;
                return pr == 2 ? lstm_fwd_flow2<4, 2, 0, true> : (pr == 1 ? lstm_fwd_flow2<4, 1, 0, true> : lstm_fwd_flow2<4, 0, 0, true>);
        }
    switch (H / 128) {
        case 1: return lstm_fwd_flow2<1, 0, 0>;
        case 2: return pr == 2 ? lstm_fwd_flow2<2, 2, 0> : (pr == 1 ? lstm_fwd_flow2<2, 1, 0> : lstm_fwd_flow2<2, 0, 0>);
        case 3: return lstm_fwd_flow2<3, 0, 0>;
        default:
            if (pr == 0 && mv == 2) return lstm_fwd_flow2<4, 0, 2>;
            if (pr == 0 && mv == 1) return lstm_fwd_flow2<4, 0, 1>;
            return pr == 2 ? lstm_fwd_flow2<4, 2, 0> : (pr == 1 ? lstm_fwd_flow2<4, 1, 0> : lstm_fwd_flow2<4, 0, 0>);
    }
}
// How many K blocks per recurrence wave the x-product workers take at this launch (0: none), and how many workgroups of every
// spare XCD run them (one role per wave).  FWD2_WORKER_RESERVE workgroups of every spare XCD exit at once: their CUs are what work
// ordered behind amdspeech_lstm_beside_forward (the next mini-batch's front end, the side-stream fills) runs on.
// AMDSPEECH_FLOW_FWD_WORKERS=0: the kernel of rounds 2 - 4 (every recurrence wave multiplies its whole x half).
#ifndef FWD2_WORKER_RESERVE
#define FWD2_WORKER_RESERVE 8
#endif
static int fwd_worker_plan(const amdspeech_lstm_desc* d, int* wpx, int* wpw) {
    static const int env = runtime_switch("AMDSPEECH_FLOW_FWD_WORKERS", 1);
    *wpx = 0; *wpw = 8;
    if (env == 0) return 0;
    const int groups = d->L * ((d->B + 15) / 16), spare = 8 - groups;
    const int mv_cap = dev_knob("AMDSPEECH_FWD_MV", FWD2_WORKER_PARTS), w0 = dev_knob("AMDSPEECH_FWD_WPW", 4);      // (development builds only)
    for (int mv = fwd_workers_max(d) < mv_cap ? fwd_workers_max(d) : mv_cap; mv > 0; --mv)
        for (int waves = w0; waves <= 8; waves += 4) {      // one role per SIMD where that fits, else two
            const int wgs = (groups * (d->H / 16) * mv + waves - 1) / waves, per = (wgs + spare - 1) / spare;
            if (per <= 32 - FWD2_WORKER_RESERVE) { *wpx = per; *wpw = waves; return mv; }
        }
    return 0;
}

// ---- the panels the dataflow kernels poll (amdspeech.h: AMDSPEECH_LSTM_ARMED / ARM_NEXT)
// forward: sentinel in every slot the kernel will write (each exactly once; layer 0 reads xp0, not xph[0])
static int flow_fill_fwd_panels(hipStream_t s, const amdspeech_lstm_desc* d, float* ws, const LstmLayout& lo, int set) {
    const size_t bph = (size_t)(d->B + 15) / 16 * 16 * d->H;
    float* base = ws + (size_t)set * lo.fwd_set;
    // (slot [L]: the top layer's output panels, polled by the fused CTC head -- filled whether or not this call has one: the set
    //  is armed for the NEXT call, whose head is not known yet)
    AS_CHECK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(base + lo.xph + (size_t)d->T * bph), (int)FLOW_SENTINEL,
                                   (size_t)d->L * d->T * bph, s));
    AS_CHECK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(base + lo.hph), (int)FLOW_SENTINEL, (size_t)d->L * (d->T + 1) * bph, s));
    return AMDSPEECH_OK;
}
// backward: the dG panels (round-1 kernel) or the two partial-tile rings (parity 0), and the dX panels between the layers
static int flow_fill_bwd_panels(hipStream_t s, const amdspeech_lstm_desc* d, float* ws, const LstmLayout& lo, bool ctc_head = false) {
    const size_t bpg = (size_t)((d->B + 15) / 16) * 16 * 4 * d->H;
    AS_CHECK_HIP(hipMemsetAsync(ws + lo.prec, 0, (lo.total - lo.prec) * sizeof(float), s));
    if (ctc_head)      // dZ_top is produced DURING the backward launch (ctc_leader) and polled by the top layer's groups
        AS_CHECK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ws + lo.dztop), (int)FLOW_SENTINEL, (size_t)d->T * d->B * d->H, s));
    if (d->L > 1)
        AS_CHECK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ws + lo.dxh), (int)FLOW_SENTINEL,
                                       (size_t)(d->L - 1) * d->T * (bpg / 4), s));
    return AMDSPEECH_OK;
}
// Side-stream fills: flow_arm_fork orders the side stream behind everything enqueued on `s` so far; the fills enqueued on it
// since are "pending" until some later lstm call makes its stream wait for them (flow_arm_settle: every dataflow call does)
// The pending state belongs to the WORKSPACE the fills write into (keyed by its base address; amdspeech_lstm_workspace_release
// forgets it): two engines -- or the two stacks of a bidirectional model -- never wait for each other's fills.
struct ArmState {
    hipEvent_t join = nullptr; bool pending = false;
    // amdspeech_lstm_beside_forward: recorded on the caller's stream just in front of the last forward dataflow launch on this
    // workspace; idle_xcds = how many XCDs that launch leaves without a recurrence group
    hipEvent_t pre = nullptr; int idle_xcds = 0;
    // amdspeech_lstm_beside_tail: recorded just behind the last backward dataflow launch on this workspace (in front of the
    // weight-gradient launches that follow it); post_flags: 1 = recorded, 2 = dZ_0 is complete at that point
    hipEvent_t post = nullptr; int post_flags = 0;
    int clean_set = 0;      // the set of forward panels an ARMED forward call finds prepared
    int xw_par = -1;        // the tag (0 / 1) the last forward launch left in EVERY word of the x-product workers' tile history it
                            // wrote; -1: unknown (the next launch zeroes the history and uses 1)
    int xw_cover = 0;       // ... and the number of leading frames that carry it (that launch's T)
    long xw_key = 0;        // ... at this shape (B, H, L, parts)
};
static std::mutex g_arm_mutex;
static std::unordered_map<const void*, ArmState> g_arm;
// The tag of this launch's tiles.  A call that may trust the history (ARMED / SAME_WS, amdspeech.h: the previous lstm_fwd on this
// workspace ran at the same B / H / L and nothing else has written to it) flips the tag the previous launch left in frames
// [0, cover) and, when it runs more frames than that launch, gives the frames [cover, T) the OLD tag first (they may hold either:
// a shorter launch in between left them alone); any other call zeroes the frames it will use (0.8 GB per part at the benchmark
// shape: once per training run).
static int flow_xw_parity(hipStream_t s, const void* ws, bool trust, long key, float* xwp, size_t frame_floats, int T, unsigned* par) {
    std::lock_guard<std::mutex> lock(g_arm_mutex);
    ArmState& st = g_arm[ws];
    if (trust && st.xw_par >= 0 && st.xw_key == key) {
        const int old = st.xw_par;
        if (T > st.xw_cover)
            AS_CHECK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(xwp + (size_t)st.xw_cover * frame_floats), old,
                                           (size_t)(T - st.xw_cover) * frame_floats, s));
        st.xw_par = old ^ 1;
    } else {
        AS_CHECK_HIP(hipMemsetAsync(xwp, 0, (size_t)T * frame_floats * sizeof(float), s));
        st.xw_par = 1;
    }
    st.xw_cover = T; st.xw_key = key;
    *par = (unsigned)st.xw_par;
    return AMDSPEECH_OK;
}
static void flow_xw_forget(const void* ws) {      // (a launch that did not complete: its tiles carry either tag)
    std::lock_guard<std::mutex> lock(g_arm_mutex);
    auto it = g_arm.find(ws);
    if (it != g_arm.end()) it->second.xw_par = -1;
}
static int flow_arm_fork(hipStream_t s) {
    if (int rc = side_stream_init()) return rc;
    AS_CHECK_HIP(hipEventRecord(g_fork, s));
    AS_CHECK_HIP(hipStreamWaitEvent(g_side, g_fork, 0));
    return AMDSPEECH_OK;
}
static int flow_clean_set(const void* ws) {
    std::lock_guard<std::mutex> lock(g_arm_mutex);
    auto it = g_arm.find(ws);
    return it == g_arm.end() ? 0 : it->second.clean_set;
}
static int flow_arm_publish(const void* ws, int clean_set) {
    std::lock_guard<std::mutex> lock(g_arm_mutex);
    ArmState& st = g_arm[ws];
    st.clean_set = clean_set;
    if (!st.join) AS_CHECK_HIP(hipEventCreateWithFlags(&st.join, hipEventDisableTiming));
    AS_CHECK_HIP(hipEventRecord(st.join, g_side));
    st.pending = true;
    return AMDSPEECH_OK;
}
static int flow_arm_settle(hipStream_t s, const void* ws) {
    std::lock_guard<std::mutex> lock(g_arm_mutex);
    auto it = g_arm.find(ws);
    if (it == g_arm.end() || !it->second.pending) return AMDSPEECH_OK;
    AS_CHECK_HIP(hipStreamWaitEvent(s, it->second.join, 0));
    it->second.pending = false;
    return AMDSPEECH_OK;
}
static int flow_arm_release(hipStream_t s, const void* ws) {
    std::lock_guard<std::mutex> lock(g_arm_mutex);
    auto it = g_arm.find(ws);
    if (it == g_arm.end()) return AMDSPEECH_OK;
    if (it->second.pending) AS_CHECK_HIP(hipStreamWaitEvent(s, it->second.join, 0));
    if (it->second.join) (void)hipEventDestroy(it->second.join);
    if (it->second.pre) (void)hipEventDestroy(it->second.pre);
    if (it->second.post) (void)hipEventDestroy(it->second.post);
    g_arm.erase(it);
    return AMDSPEECH_OK;
}
// the point in stream `s` just in front of a forward launch on `ws` (idle_xcds = 0: a launch that leaves nothing idle)
static int flow_mark_postlaunch(hipStream_t s, const void* ws, int flags) {
    std::lock_guard<std::mutex> lock(g_arm_mutex);
    ArmState& st = g_arm[ws];
    st.post_flags = flags;
    if (flags == 0) return AMDSPEECH_OK;
    if (!st.post) AS_CHECK_HIP(hipEventCreateWithFlags(&st.post, hipEventDisableTiming));
    AS_CHECK_HIP(hipEventRecord(st.post, s));
    return AMDSPEECH_OK;
}
static int flow_mark_prelaunch(hipStream_t s, const void* ws, int idle_xcds) {
    std::lock_guard<std::mutex> lock(g_arm_mutex);
    ArmState& st = g_arm[ws];
    st.idle_xcds = idle_xcds;
    if (idle_xcds <= 0) return AMDSPEECH_OK;
    if (!st.pre) AS_CHECK_HIP(hipEventCreateWithFlags(&st.pre, hipEventDisableTiming));
    AS_CHECK_HIP(hipEventRecord(st.pre, s));
    return AMDSPEECH_OK;
}

// The fused CTC head (amdspeech.h: amdspeech_lstm_ctc_fusable): which launches take it, and how many workgroups of every spare
// XCD follow the forward recurrence (behind the x-product workers; the rest stay free for side-stream work)
static int ctc_head_plan(const amdspeech_lstm_desc* d, int C, int U) {
    static const int env = runtime_switch("AMDSPEECH_FLOW_CTC", 1);      // 0: the CTC stage as launches between the two recurrence kernels
    if (env == 0 || d == nullptr || !use_flow(d) || (d->flags & AMDSPEECH_LSTM_PER_DIAGONAL)) return 0;
    const int groups = d->L * ((d->B + 15) / 16), spare = 8 - groups, smax = 2 * U + 1;
    if (spare < 1 || C < 16 || C > 16 * CF_NTC || C % 16 != 0 || U < 1 || smax > 384 || d->H % 64 != 0) return 0;
    if ((size_t)d->B * d->T * smax * 4 >= (1ull << 31) || (size_t)d->T * d->B * d->H * 4 >= (1ull << 31)) return 0;
    int wpx = 0, wpw = 8;
    fwd_worker_plan(d, &wpx, &wpw);
    int nfw = 32 - wpx < 4 ? 32 - wpx : 4;
    if (nfw < 1 || d->B > spare * nfw * 2 * 2) return 0;      // at most two utterances per team
    return nfw;
}
static CtcFlow ctc_head_args(const amdspeech_lstm_desc* d, const amdspeech_ctc_head* h, float* ws, const LstmLayout& lo, float* panels, int nfw) {
    const CtcLayout cl = ctc_layout(d->T, d->B, h->C, h->U);
    char* w = static_cast<char*>(h->ctc_ws);
    CtcFlow c;
    c.on = 1; c.C = h->C; c.smax = cl.smax; c.nfw = nfw; c.T = d->T; c.B = d->B;
    c.ztp = panels ? panels + lo.xph + (size_t)d->L * d->T * ((size_t)(d->B + 15) / 16 * 16 * d->H) : nullptr;
    c.wo = h->w_out; c.wo_pack = ws + lo.wopack; c.bo = h->b_out;
    c.logits = h->logits; c.logp = reinterpret_cast<float*>(w + cl.logp); c.alpha = reinterpret_cast<float*>(w + cl.alpha);
    c.ll = reinterpret_cast<float*>(w + cl.ll); c.loss = h->loss; c.dlogits = h->dlogits; c.dztop = ws + lo.dztop;
    c.ext = reinterpret_cast<const int*>(w + cl.ext); c.slen = reinterpret_cast<const int*>(w + cl.slen);
    c.valid = reinterpret_cast<const int*>(w + cl.valid);
    return c;
}

// ------------------------------------------------------------------- H = 1024 forward, layer by layer: one stack, or two side by side
struct BigStack { const amdspeech_lstm_desc* d; float* ws; const float* kernels; long kstride; const float* biases; long bstride; const int* lengths; };
static int big_fwd_layers(hipStream_t s, int n, const BigStack* st) {
    const amdspeech_lstm_desc* d = st[0].d;
    const int T = d->T, B = d->B, H = d->H, L = d->L, nmt = ceil_div(B, 16);
    const size_t TB = (size_t)T * B, bp = (size_t)nmt * 16, bh = (size_t)B * H;
    // (one stack alone: the XCD pairs are faster -- 13.3 against 14.7 ms of recurrence at configs[2]'s shape, half the MFMAs and half
    //  the LDS traffic per CU and step; AMDSPEECH_BIG1=2 runs it on the one-XCD groups all the same)
    static const int big1_env = runtime_switch("AMDSPEECH_BIG1", 1);
    const bool big1 = use_big1_fwd(d) && (n == 2 || big1_env == 2);
    AS_CHECK_ARG(n == 1 || (n == 2 && big1), "lstm_fwd: two stacks side by side need the one-XCD groups");
    if (big1) {
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_big1), hipFuncAttributeMaxDynamicSharedMemorySize, BIG1_LDS_BYTES);
        AS_CHECK_HIP(once);
    }
    BigFwd1Args b1;
    b1.n = n;
    LstmLayout lo[2];
    for (int k = 0; k < n; ++k) {
        const BigStack& q = st[k];
        lo[k] = lstm_layout(q.d);
        unsigned* err = reinterpret_cast<unsigned*>(q.ws + lo[k].sync);
        BigFwdArgs& ba = b1.b[k];
        ba.wp = q.ws + lo[k].wp; ba.z = q.ws + lo[k].z; ba.hs = q.ws + lo[k].hs; ba.cs = q.ws + lo[k].cs; ba.gates = q.ws + lo[k].gates;
        ba.lengths = q.lengths;
        ba.err = err; ba.tickets = err + 16;
        ba.T = T; ba.B = B; ba.H = H; ba.L = L; ba.drop = DropCfg{q.d->keep_in, q.d->keep_out, q.d->seed, L};
        ba.limit = (q.d->flags & AMDSPEECH_LSTM_INJECT_TIMEOUT) ? 0ull : 100000000ull + (unsigned long long)T * 10000ull;      // (INJECT_TIMEOUT: tests)
    }
    if (n == 1) b1.b[1] = b1.b[0];
    for (int l = 0; l < L; ++l) {
        for (int k = 0; k < n; ++k) {
            const BigStack& q = st[k];
            float* ws = q.ws;
            const LstmLayout& lk = lo[k];
            // pre-activations of ALL frames: [T*B, H] . K_l[0:H, :] + b_l -> gates[l] (replaced frame by frame by the kernel)
            if (int rc = bf16p_layout_on(q.d) ? bf16p_xw(s, bf16p_bufs(q.d, ws + lk.bfs), (int)TB, H, ws + lk.z + (size_t)l * TB * H, q.kernels + l * q.kstride,
                                                         ws + lk.gates + (size_t)l * TB * 4 * H, q.biases + l * q.bstride)
                       : bf3_gemm(q.d) ? gemm_reduced(q.d, s, false, false, (int)TB, 4 * H, H, ws + lk.z + (size_t)l * TB * H, H, q.kernels + l * q.kstride,
                                                      4 * H, ws + lk.gates + (size_t)l * TB * 4 * H, 4 * H, q.biases + l * q.bstride, false)
                                       : gemm_f32_plain(s, false, false, (int)TB, 4 * H, H, ws + lk.z + (size_t)l * TB * H, H, q.kernels + l * q.kstride,
                                                        4 * H, ws + lk.gates + (size_t)l * TB * 4 * H, 4 * H, q.biases + l * q.bstride, false)) return rc;
            // the h ring of this layer: slot 0 = the packed initial state with every word tagged 1, slot 1 = zeros (tag 0)
            float* ring = ws + lk.hp + (size_t)l * 2 * bp * H;
            AS_CHECK_HIP(hipMemsetAsync(ring, 0, 2 * bp * H * sizeof(float), s));
            hipLaunchKernelGGL(pack_rows_kernel, dim3(ceil_div(bh, 256)), dim3(256), 0, s,
                               ws + lk.hs + (size_t)l * (T + 1) * bh, bh, ring, B, H, 1);
            hipLaunchKernelGGL(tag_panel_kernel, dim3(ceil_div(bp * H, 256)), dim3(256), 0, s, ring, bp * H, 1u);
            AS_CHECK_HIP(hipMemsetAsync(b1.b[k].tickets, 0, 8 * sizeof(unsigned), s));
            b1.b[k].hring = ring; b1.b[k].layer = l;
        }
        if (n == 1) b1.b[1] = b1.b[0];
        const BigFwdArgs& ba = b1.b[0];
        prof_begin(0, s, l);
        if (big1) hipLaunchKernelGGL(lstm_fwd_big1, dim3(256), dim3(512), BIG1_LDS_BYTES, s, b1);
        else if (d->precision == 2) hipLaunchKernelGGL(lstm_fwd_big<2>, dim3(256), dim3(512), 0, s, ba);
        else if (d->precision == 1) hipLaunchKernelGGL(lstm_fwd_big<1>, dim3(256), dim3(512), 0, s, ba);
        else hipLaunchKernelGGL(lstm_fwd_big<0>, dim3(256), dim3(512), 0, s, ba);      // one workgroup per CU; each finds its place by XCC_ID
        prof_end(0, s, T * L, l);
    }
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}


int lstm_fwd(hipStream_t s, const amdspeech_lstm_desc* d, float* ws, const float* kernels, long kstride,
             const float* biases, long bstride, const int* lengths, const float* h0, const float* c0,
             const amdspeech_ctc_head* head = nullptr, bool defer_big = false) {
    if (int rc = check_desc(d)) return rc;
    AS_CHECK_ARG(ws && kernels && biases && lengths, "lstm_fwd: null pointer");
    AS_CHECK_ARG(((uintptr_t)ws % 256) == 0, "lstm_fwd: workspace must be 256-byte aligned");
    if (int rc = flow_arm_settle(s, ws)) return rc;      // (fills a previous call on THIS workspace left on the side stream: see AMDSPEECH_LSTM_ARM_NEXT)
    if (int rc = flow_mark_prelaunch(s, ws, 0)) return rc;       // (until a dataflow launch below says otherwise)
    const LstmLayout lo = lstm_layout(d);
    const int T = d->T, B = d->B, H = d->H, L = d->L;
    const bool flow = use_flow(d);
    AS_CHECK_ARG(head == nullptr || flow, "lstm_fwd_ctc: the fused CTC head needs the whole-sequence kernels (amdspeech_lstm_ctc_fusable)");
    const bool big = !flow && use_big_fwd(d);
    const bool bf3 = d->precision != 0 && !flow && !big;      // (precision 2 outside the dataflow / per-layer shapes: the bf16x3 step kernels, a superset in accuracy) (the dataflow and per-layer kernels split their f32 fragments in registers: f32 packs)
    const bool hoist = big || (use_hoist(d, flow) & 1);
    prof_flops(0, 0.0, 0.0);
    AS_CHECK_HIP(hipMemsetAsync(ws + lo.sync, 0, 64, s));      // error word read by amdspeech_lstm_status (every path)
    const int uw = (flow || big) ? 16 : pick_uw(d);      // the dataflow kernels own 16 units x 4 gates per workgroup
    const long wtotal = (long)L * 2 * H * 4 * H;
    if (bf3)
        hipLaunchKernelGGL(pack_fwd_bf3_kernel, dim3(ceil_div(wtotal, 256)), dim3(256), 0, s, kernels, kstride,
                           reinterpret_cast<unsigned short*>(ws + lo.wp), H, L);
    else
        hipLaunchKernelGGL(pack_fwd_kernel, dim3(ceil_div(wtotal, 256)), dim3(256), 0, s, kernels, kstride,
                           ws + lo.wp, H, L, uw, 0);
    AS_CHECK_LAUNCH();
    const size_t bh = (size_t)B * H;
    for (int l = 0; l < L && !flow; ++l) {      // (dataflow path: flow_fwd_prepare_kernel below)
        float* hs0 = ws + lo.hs + (size_t)l * (T + 1) * bh;
        float* cs0 = ws + lo.cs + (size_t)l * (T + 1) * bh;
        if (h0) AS_CHECK_HIP(hipMemcpyAsync(hs0, h0 + l * bh, bh * 4, hipMemcpyDeviceToDevice, s));
        else AS_CHECK_HIP(hipMemsetAsync(hs0, 0, bh * 4, s));
        if (c0) AS_CHECK_HIP(hipMemcpyAsync(cs0, c0 + l * bh, bh * 4, hipMemcpyDeviceToDevice, s));
        else AS_CHECK_HIP(hipMemsetAsync(cs0, 0, bh * 4, s));
    }
    DropCfg dc{d->keep_in, d->keep_out, d->seed, L};
    if (d->keep_in < 1.0f && !flow) {
        const long n = (long)T * bh;
        hipLaunchKernelGGL(apply_zmult_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, ws + lo.z, n, dc, 0);
        AS_CHECK_LAUNCH();
    }
    if (!flow) {   // packed A panels: layer-0 input for every frame, initial h of every layer (slot = l & 1)
        const size_t bp = (size_t)(B + 15) / 16 * 16;
        const size_t n0 = (size_t)T * bh;
        if (bf3) {
            hipLaunchKernelGGL(pack_rows_bf3_kernel, dim3(ceil_div(n0, 256)), dim3(256), 0, s, ws + lo.z, bh,
                               reinterpret_cast<unsigned short*>(ws + lo.xp0), B, H, T);
            for (int l = 0; l < L; ++l)
                hipLaunchKernelGGL(pack_rows_bf3_kernel, dim3(ceil_div(bh, 256)), dim3(256), 0, s,
                                   ws + lo.hs + (size_t)l * (T + 1) * bh, bh,
                                   reinterpret_cast<unsigned short*>(ws + lo.hp + ((size_t)l * 2 + (l & 1)) * bp * H), B, H, 1);
        } else {
            if (!hoist)
                hipLaunchKernelGGL(pack_rows_kernel, dim3(ceil_div(n0, 256)), dim3(256), 0, s, ws + lo.z, bh, ws + lo.xp0, B, H, T);
            for (int l = 0; l < L; ++l)      // (slot of the first launch that reads it: diagonal l, or frame 0 when hoisted)
                hipLaunchKernelGGL(pack_rows_kernel, dim3(ceil_div(bh, 256)), dim3(256), 0, s,
                                   ws + lo.hs + (size_t)l * (T + 1) * bh, bh, ws + lo.hp + ((size_t)l * 2 + (hoist ? 0 : (l & 1))) * bp * H,
                                   B, H, 1);
        }
        AS_CHECK_LAUNCH();
    }
    FwdArgs a;
    a.xp0 = ws + lo.xp0; a.xp = ws + lo.xp; a.hp = ws + lo.hp;
    a.wp = ws + lo.wp; a.bias = biases; a.bias_stride = bstride;
    a.z = ws + lo.z; a.hs = ws + lo.hs; a.cs = ws + lo.cs; a.gates = ws + lo.gates; a.lengths = lengths;
    a.T = T; a.B = B; a.H = H; a.L = L; a.drop = dc;
    a.hoist = 0; a.l0 = 0;
    a.dbg = dev_knob("AMDSPEECH_DBG", 0);
    a.trace = dev_trace_ptr(); a.trace_d = a.trace ? dev_knob("AMDSPEECH_TRACE_D", T / 2) : -1;      // (development builds only)
    if (flow) {
        const size_t bp = (size_t)(B + 15) / 16 * 16, bph = bp * H;
        unsigned* err = reinterpret_cast<unsigned*>(ws + lo.sync);
        // sentinel pre-fill of every slot the kernel will write (unless the previous forward call of the training cycle has
        // done it behind its own kernel: AMDSPEECH_LSTM_ARMED) ...
        const int set = (d->flags & AMDSPEECH_LSTM_ARMED) ? flow_clean_set(ws) : 0;
        if (!(d->flags & AMDSPEECH_LSTM_ARMED))
            if (int rc = flow_fill_fwd_panels(s, d, ws, lo, set)) return rc;
        float* const panels = ws + (size_t)set * lo.fwd_set;
        // ... then, in one launch each: the initial state (rows + packed slot 0 of every layer), error word and tickets; and
        // the layer-0 operand panels of all frames, the input dropout mask applied on the way
        hipLaunchKernelGGL(flow_fwd_prepare_kernel, dim3(ceil_div((long)L * bph, 256)), dim3(256), 0, s, h0, c0, ws + lo.hs, ws + lo.cs,
                           panels + lo.hph, err, T, B, H, L);
        hipLaunchKernelGGL(mask_pack_rows_kernel, dim3(ceil_div((long)T * bh / 4, 256)), dim3(256), 0, s, ws + lo.z, ws + lo.xp0, B, H, T,
                           dc, d->keep_in < 1.0f ? 1 : 0);
        AS_CHECK_LAUNCH();
        FlowArgs fa;
        fa.wp = a.wp; fa.bias = biases; fa.bias_stride = bstride;
        fa.z = a.z; fa.hs = a.hs; fa.cs = a.cs; fa.gates = a.gates; fa.lengths = lengths;
        fa.xp0 = a.xp0; fa.xph = panels + lo.xph; fa.hph = panels + lo.hph; fa.err = err;
        fa.T = T; fa.B = B; fa.H = H; fa.L = L; fa.drop = dc;
        // generous bound on the whole sequence: 100 us per step plus a second (100 MHz ticks)
        fa.limit = (d->flags & AMDSPEECH_LSTM_INJECT_TIMEOUT) ? 0ull : 100000000ull + (unsigned long long)T * 10000ull;      // (INJECT_TIMEOUT: tests)
        fa.trace = a.trace; fa.trace_layer = dev_knob("AMDSPEECH_TRACE_LAYER", L > 1 ? 1 : 0);
        fa.tickets = err + 16;
        int wpx = 0, wpw = 8;
        const int mv = fwd_worker_plan(d, &wpx, &wpw);
        fa.xwp = ws + lo.xwp; fa.xw_par = 0u; fa.w_wpx = wpx; fa.w_wpw = wpw;
        if (mv > 0)
            if (int rc = flow_xw_parity(s, ws, (d->flags & (AMDSPEECH_LSTM_ARMED | AMDSPEECH_LSTM_SAME_WS)) != 0,
                                        (((long)B * 4096 + H) * 64 + L) * 8 + mv, fa.xwp, (size_t)L * (bp / 16) * (H / 16) * mv * 1024, T,
                                        &fa.xw_par)) return rc;
        void (*fk)(FlowArgs) = flow_fwd_kernel(H, d->precision, mv, head != nullptr || dev_knob("AMDSPEECH_FORCE_CF", 0) != 0);      // (dev: the CF instantiation without a head)
        fa.cf = CtcFlow{}; fa.cf_on = 0; fa.cf_nfw = 0;
        size_t fwd_lds = 0;
        if (head != nullptr) {
            // the fused CTC head: extended targets and W_o's fragments first (both read by the follower workgroups of the launch)
            const int nfw = ctc_head_plan(d, head->C, head->U);
            AS_CHECK_ARG(nfw > 0, "lstm_fwd_ctc: this shape does not take the fused CTC head (amdspeech_lstm_ctc_fusable)");
            fa.cf = ctc_head_args(d, head, ws, lo, panels, nfw); fa.cf_on = 1; fa.cf_nfw = nfw;
            fa.cf.lengths = lengths; fa.cf.err = err; fa.cf.limit = fa.limit;
            if (int rc = ctc_prepare_targets(s, head->dense_labels, lengths, T, B, head->C, head->U, head->ctc_ws)) return rc;
            hipLaunchKernelGGL(ctc_pack_wo_kernel, dim3(ceil_div((H / 16) * CF_NTC * 64, 256)), dim3(256), 0, s, head->w_out, ws + lo.wopack, H, head->C);
            AS_CHECK_LAUNCH();
            fwd_lds = (size_t)2 * CF_FOLLOW_TEAM_FLOATS * sizeof(float);
            AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds));
        }
        prof_begin(0, s);
        // ... and, in a training cycle, the backward call's panels go out beside the kernel (it leaves two XCDs idle)
        const bool arm = (d->flags & AMDSPEECH_LSTM_ARM_NEXT) != 0;
        if (arm)
            if (int rc = flow_arm_fork(s)) return rc;
        // (amdspeech_lstm_beside_forward; with x-product workers on the spare XCDs nothing is "idle": the next mini-batch's front end
        //  beside them cost the recurrence 0.1 - 0.2 ms and the step 0.06 - 0.13 -- the caller then places it beside the CTC stage)
        // (with the fused CTC head there is no CTC stage to place it beside: the remaining reserved workgroups' CUs take it again)
        if (int rc = flow_mark_prelaunch(s, ws, (mv > 0 && head == nullptr) ? 0 : 8 - L * ((B + 15) / 16))) return rc;
        hipLaunchKernelGGL(fk, dim3(256), dim3(512), fwd_lds, s, fa);  // one workgroup per CU; each finds its group by XCC_ID
        prof_end(0, s, T + L - 1);
        prof_flops(0, (double)T * L * 2.0 * B * 2 * H * 4 * H, 0.0);
        AS_CHECK_LAUNCH();
        if (arm) {
            // beside the kernel (two XCDs and all of HBM idle): what lstm_bwd polls, its transposed weight pack, and the OTHER set
            // of forward panels for the next forward call of the same shape (rounds 2 - 3a re-filled this call's own set behind
            // the kernel: 330 MB beside the output layer and the log-softmax, +35 us on the critical path).  Nothing is joined
            // here: the next dataflow call on any stream waits for the side stream first (flow_arm_settle)
            // (The fills as ONE work-queue launch that really runs beside the forward kernel were built and measured in round 5 -- the
            //  forward kernel 4.46-4.51 -> 4.70-5.20 ms, the step 12.40-12.43 -> 12.59-12.73 ms: 550 MB of stores through the fabric the
            //  x-product workers and the CTC follower read through cost the recurrence more than the 0.1 ms these launches spend between
            //  the two recurrence kernels -- and removed in round 6.)
            if (int rc = flow_fill_bwd_panels(g_side, d, ws, lo, head != nullptr)) return rc;
            hipLaunchKernelGGL(pack_bwd_kernel, dim3(ceil_div(wtotal, 256)), dim3(256), 0, g_side, kernels, kstride, ws + lo.wq, H, L);
            AS_CHECK_LAUNCH();      // (the backward call's K^T pack: the weights do not change between the two halves of a cycle)
            if (int rc = flow_fill_fwd_panels(g_side, d, ws, lo, 1 - set)) return rc;
            if (int rc = flow_arm_publish(ws, 1 - set)) return rc;
        }
        return AMDSPEECH_OK;
    }
    if (bf3) {
        a.mt0 = 0;
        dim3 grid(H / 8, L, ceil_div(ceil_div(B, 16), 2)), block(8 * 64);
        prof_begin(0, s);
        for (int dd = 0; dd < T + L - 1; ++dd) {
            a.d = dd;
            hipLaunchKernelGGL(lstm_fwd_step_bf3<8>, grid, block, 0, s, a);
        }
        prof_end(0, s, T + L - 1);
        AS_CHECK_LAUNCH();
        return AMDSPEECH_OK;
    }
    static const int fwd_nw = dev_knob("AMDSPEECH_FWD_NW", 8);
    static const int fwd_un = dev_knob("AMDSPEECH_FWD_UN", 8);
    static const int fwd_db = dev_knob("AMDSPEECH_FWD_DB", 0);
    const int nmt = ceil_div(B, 16);
    if (big) {
        if (defer_big) return AMDSPEECH_OK;      // (amdspeech_lstm_fwd_pair: the layers of the two stacks run together, below)
        const BigStack one{d, ws, kernels, kstride, biases, bstride, lengths};
        return big_fwd_layers(s, 1, &one);
    }
    if (hoist) {
        void (*kern)(FwdArgs) = nullptr;
        const int mt = (nmt % 2 == 0) ? 2 : 1;
#define FWD_CASE(U, W, N, D) if (uw == U && fwd_nw == W && fwd_un == N && fwd_db == D) \
        kern = mt == 2 ? lstm_fwd_step<U, W, N, D != 0, 2> : lstm_fwd_step<U, W, N, D != 0, 1>;
        FWD_CASE(4, 8, 8, 0) FWD_CASE(8, 8, 8, 0) FWD_CASE(8, 8, 4, 1) FWD_CASE(8, 4, 8, 0)
#undef FWD_CASE
        AS_CHECK_ARG(kern != nullptr, "lstm_fwd (hoisted): no kernel variant for UW=%d NW=%d UN=%d", uw, fwd_nw, fwd_un);
        a.hoist = 1; a.mt0 = 0;
        dim3 grid(H / uw, 1, nmt / mt), block(fwd_nw * 64);
        const size_t TB = (size_t)T * B;
        prof_begin(0, s);
        for (int l = 0; l < L; ++l) {
            // pre-activations of ALL frames: [T*B, H] . K_l[0:H, :] + b_l -> gates[l] (replaced frame by frame below)
            if (int rc = gemm_f32(s, false, false, (int)TB, 4 * H, H, ws + lo.z + (size_t)l * TB * H, H, kernels + l * kstride, 4 * H,
                                  ws + lo.gates + (size_t)l * TB * 4 * H, 4 * H, biases + l * bstride, false)) return rc;
            a.l0 = l;
            for (int t = 0; t < T; ++t) {
                a.d = t;
                hipLaunchKernelGGL(kern, grid, block, 0, s, a);
            }
        }
        prof_end(0, s, T * L);
        AS_CHECK_LAUNCH();
        return AMDSPEECH_OK;
    }
    const int chains = num_chains(B);
    prof_begin(0, s);
    if (chains == 2) {
        if (int rc = side_stream_init()) return rc;
        AS_CHECK_HIP(hipEventRecord(g_fork, s));
        AS_CHECK_HIP(hipStreamWaitEvent(g_side, g_fork, 0));
    }
    for (int c = 0; c < chains; ++c) {
        const int t0 = c * nmt / chains, t1 = (c + 1) * nmt / chains;   // 16-row tiles of this chain
        const int mt = ((t1 - t0) % 2 == 0) ? 2 : 1;
        void (*kern)(FwdArgs) = nullptr;
#define FWD_CASE(U, W, N, D) if (uw == U && fwd_nw == W && fwd_un == N && fwd_db == D) \
        kern = mt == 2 ? lstm_fwd_step<U, W, N, D != 0, 2> : lstm_fwd_step<U, W, N, D != 0, 1>;
        FWD_CASE(4, 4, 8, 1) FWD_CASE(4, 8, 8, 0) FWD_CASE(4, 8, 4, 1) FWD_CASE(4, 16, 4, 0)
        FWD_CASE(8, 4, 4, 1) FWD_CASE(8, 8, 4, 1) FWD_CASE(8, 8, 8, 0) FWD_CASE(8, 16, 4, 0) FWD_CASE(8, 4, 8, 0)
        FWD_CASE(8, 4, 16, 0) FWD_CASE(8, 8, 2, 1) FWD_CASE(8, 8, 1, 1) FWD_CASE(8, 8, 2, 0) FWD_CASE(8, 4, 4, 1) FWD_CASE(8, 4, 2, 1)
#undef FWD_CASE
        AS_CHECK_ARG(kern != nullptr, "lstm_fwd: no kernel variant for UW=%d NW=%d UN=%d", uw, fwd_nw, fwd_un);
        dim3 grid(H / uw, L, (t1 - t0) / mt), block(fwd_nw * 64);
        hipStream_t cs = c == 0 ? s : g_side;
        a.mt0 = t0;
        // chains are enqueued one after the other (each queue drains independently on the GPU)
        for (int dd = 0; dd < T + L - 1; ++dd) {
            a.d = dd;
            hipLaunchKernelGGL(kern, grid, block, 0, cs, a);
        }
    }
    if (chains == 2) {
        AS_CHECK_HIP(hipEventRecord(g_join, g_side));
        AS_CHECK_HIP(hipStreamWaitEvent(s, g_join, 0));
    }
    prof_end(0, s, T + L - 1);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

// H = 1024 backward in plain bf16 on the one-XCD groups (lstm_bwd_big1), layer by layer, top first: two stacks side by side
struct BigBwdStack { const amdspeech_lstm_desc* d; float* ws; const float* kernels; long kstride; float* dkernels; float* dbiases; long bstride; const int* lengths; };
static bool use_big1_bwd(const amdspeech_lstm_desc* d) {
    const int nmt = (d->B + 15) / 16;
    return use_big1_fwd(d) && !use_flow(d) && bf16p_layout_on(d) && (size_t)2 * nmt * 64 * 64 * 1024 < (1ull << 32);
}
static int big1_bwd_layers(hipStream_t s, int n, const BigBwdStack* st) {
    const amdspeech_lstm_desc* d = st[0].d;
    const int T = d->T, B = d->B, H = d->H, L = d->L, nmt = ceil_div(B, 16);
    const size_t TB = (size_t)T * B;
    const size_t pring_floats = (size_t)2 * nmt * 2 * 32 * 32 * 256, xring_floats = (size_t)2 * nmt * 64 * 1024;
    BigBwd1Args b1;
    b1.n = n;
    LstmLayout lo[2];
    for (int k = 0; k < n; ++k) {
        const BigBwdStack& q = st[k];
        lo[k] = lstm_layout(q.d);
        unsigned* err = reinterpret_cast<unsigned*>(q.ws + lo[k].sync);
        BigBwdArgs& b2 = b1.b[k];
        b2.wq = q.ws + lo[k].wq; b2.cs = q.ws + lo[k].cs; b2.gates = q.ws + lo[k].gates; b2.dg = q.ws + lo[k].dg; b2.dup = q.ws + lo[k].dztop;
        b2.lengths = q.lengths; b2.pring = q.ws + lo[k].bigring; b2.xring = q.ws + lo[k].bigring + pring_floats; b2.err = err; b2.tickets = err + 16;
        b2.T = T; b2.B = B; b2.H = H; b2.L = L; b2.drop = DropCfg{q.d->keep_in, q.d->keep_out, q.d->seed, L};
        b2.limit = (q.d->flags & AMDSPEECH_LSTM_INJECT_TIMEOUT) ? 0ull : 100000000ull + (unsigned long long)T * 10000ull;      // (INJECT_TIMEOUT: tests)
    }
    for (int l = L - 1; l >= 0; --l) {
        for (int k = 0; k < n; ++k) {
            AS_CHECK_HIP(hipMemsetAsync(b1.b[k].pring, 0, (pring_floats + xring_floats) * sizeof(float), s));
            AS_CHECK_HIP(hipMemsetAsync(b1.b[k].tickets, 0, 8 * sizeof(unsigned), s));
            b1.b[k].layer = l;
            const Bf16pBufs bufs = bf16p_bufs(st[k].d, st[k].ws + lo[k].bfs);
            b1.b[k].dgb = bufs.dgb; b1.b[k].dbias = st[k].dbiases + l * st[k].bstride;
        }
        if (n == 1) b1.b[1] = b1.b[0];
        prof_begin(1, s, L - 1 - l);
        hipLaunchKernelGGL(lstm_bwd_big1<BIG1_Q>, dim3(256), dim3(512), 0, s, b1);
        prof_end(1, s, T * L, L - 1 - l);
        for (int k = 0; k < n; ++k) {      // everything this layer owes, now (dZ_0 for the bottom layer)
            const BigBwdStack& q = st[k];
            float* ws = q.ws;
            if (int rc = bf16p_layer_bwd_copied(s, bf16p_bufs(q.d, ws + lo[k].bfs), (int)TB, H, ws + lo[k].z + (size_t)l * TB * H,
                                                ws + lo[k].hs + (size_t)l * (T + 1) * B * H, q.kernels + l * q.kstride,
                                                l > 0 ? ws + lo[k].dztop : ws + lo[k].dz0, q.dkernels + l * q.kstride)) return rc;
        }
    }
    AS_CHECK_LAUNCH();
    for (int k = 0; k < n; ++k)
        if (st[k].d->keep_in < 1.0f) {     // the layer-0 input dropout mask on dZ_0
            const long cnt = (long)T * B * H;
            hipLaunchKernelGGL(apply_zmult_kernel, dim3(ceil_div(cnt, 256)), dim3(256), 0, s, st[k].ws + lo[k].dz0, cnt,
                               DropCfg{st[k].d->keep_in, st[k].d->keep_out, st[k].d->seed, L}, 0);
            AS_CHECK_LAUNCH();
        }
    return AMDSPEECH_OK;
}

int lstm_bwd(hipStream_t s, const amdspeech_lstm_desc* d, float* ws, const float* kernels, long kstride,
             float* dkernels, float* dbiases, long bstride, const int* lengths, const amdspeech_ctc_head* head = nullptr,
             bool defer_big = false) {
    if (int rc = check_desc(d)) return rc;
    AS_CHECK_ARG(ws && kernels && dkernels && dbiases && lengths, "lstm_bwd: null pointer");
    if (int rc = flow_arm_settle(s, ws)) return rc;      // (fills lstm_fwd left on the side stream: see AMDSPEECH_LSTM_ARM_NEXT)
    if (int rc = flow_mark_postlaunch(s, ws, 0)) return rc;      // (until a dataflow launch below says otherwise)
    const LstmLayout lo = lstm_layout(d);
    const int T = d->T, B = d->B, H = d->H, L = d->L;
    const long wtotal = (long)L * 2 * H * 4 * H;
    const bool bf3 = d->precision != 0 && !use_flow(d) && !use_big_fwd(d);
    if (bf3)
        hipLaunchKernelGGL(pack_bwd_bf3_kernel, dim3(ceil_div(wtotal, 256)), dim3(256), 0, s, kernels, kstride,
                           reinterpret_cast<unsigned short*>(ws + lo.wq), H, L);
    else if (!(use_flow(d) && (d->flags & AMDSPEECH_LSTM_ARMED)))      // (armed: lstm_fwd packed K^T beside its kernel)
        hipLaunchKernelGGL(pack_bwd_kernel, dim3(ceil_div(wtotal, 256)), dim3(256), 0, s, kernels, kstride,
                           ws + lo.wq, H, L);
    AS_CHECK_LAUNCH();
    DropCfg dc{d->keep_in, d->keep_out, d->seed, L};
    const bool flow = use_flow(d);
    AS_CHECK_ARG(head == nullptr || flow, "lstm_bwd_ctc: the fused CTC head needs the whole-sequence kernels (amdspeech_lstm_ctc_fusable)");
    const bool hoist = (use_hoist(d, flow) & 2) != 0;
    BwdArgs a;
    a.hoist = 0; a.l0 = 0;
    a.wq = ws + lo.wq; a.cs = ws + lo.cs; a.gates = ws + lo.gates; a.dg = ws + lo.dg;
    a.dztop = ws + lo.dztop; a.dc = ws + lo.dc; a.lengths = lengths; a.dgp = ws + lo.dgp;
    a.T = T; a.B = B; a.H = H; a.L = L; a.drop = dc;
    static const int bwd_nw = dev_knob("AMDSPEECH_BWD_NW", 8);
    static const int bwd_un = dev_knob("AMDSPEECH_BWD_UN", 8);
    static const int bwd_db = dev_knob("AMDSPEECH_BWD_DB", 1);
    void (*kern)(BwdArgs) = nullptr;
#define BWD_CASE(W, N, D) if (bwd_nw == W && bwd_un == N && bwd_db == D) kern = lstm_bwd_step<W, N, D != 0>;
    BWD_CASE(4, 8, 1) BWD_CASE(8, 8, 1) BWD_CASE(8, 16, 0) BWD_CASE(16, 8, 0) BWD_CASE(4, 16, 0)
#undef BWD_CASE
    if (bf3) kern = lstm_bwd_step_bf3<8>;
    AS_CHECK_ARG(kern != nullptr, "lstm_bwd: no kernel variant for NW=%d UN=%d", bwd_nw, bwd_un);
    const int nmt = ceil_div(B, 16);
    const int chains = num_chains(B);
    // Time-independent weight gradients of the frames [ta, tb): dK_l += [Z_l ; Hprev_l]^T . dG_l,
    // db_l += colsum(dG_l) (rides on the first GEMM), and dZ_0 = dG_0 . K_0[0:H,:]^T.
    unsigned* gate_err = reinterpret_cast<unsigned*>(ws + lo.sync);
    prof_flops(1, 0.0, 0.0);
    auto weight_grads = [&](hipStream_t gs, int ta, int tb, const int* gate, int need, int dz_tb = -1) -> int {
        const size_t TB = (size_t)T * B, r0 = (size_t)ta * B;
        const int rows = (tb - ta) * B;
        const int dz_rows = ((dz_tb < 0 ? tb : dz_tb) - ta) * B;      // dZ_0 may cover more frames than the weight gradients
        // the 2 L products dK_l = [Z_l ; Hprev_l]^T . dG_l in ONE launch (GEMM_GROUP_MAX problems at a time)
        const float* pa[GEMM_GROUP_MAX]; const float* pb[GEMM_GROUP_MAX]; float* pc[GEMM_GROUP_MAX]; float* ps[GEMM_GROUP_MAX];
        int np = 0;
        for (int l = 0; l < L; ++l) {
            const float* dg = ws + lo.dg + ((size_t)l * TB + r0) * 4 * H;
            float* dk = dkernels + l * kstride;
            pa[np] = ws + lo.z + ((size_t)l * TB + r0) * H; pb[np] = dg; pc[np] = dk; ps[np] = dbiases + l * bstride; ++np;
            pa[np] = ws + lo.hs + ((size_t)l * (T + 1) * B + r0) * H;   // slots 0..T-1 = h_{t-1}
            pb[np] = dg; pc[np] = dk + (size_t)H * 4 * H; ps[np] = nullptr; ++np;
            // (per layer: the two products share dG_l, and 2 x 64 tiles x 2 K splits = one workgroup per CU; all 2 L in one launch
            //  put three waves on every SIMD and ran 30 % slower)
            static const int group_max = dev_knob("AMDSPEECH_GEMM_GROUP", 2);
            if (bf16p_layout_on(d) && gate == nullptr && rows % 64 == 0 && rows >= 64) {
                // plain bf16 through operand copies: both halves of the layer's kernel gradient as ONE product, the bias gradient on
                // the transposing copy of dG
                if (int rc = bf16p_dk(gs, bf16p_bufs(d, ws + lo.bfs), rows, H, pa[0], pa[1], pb[0], pc[0], ps[0])) return rc;
                np = 0;
                continue;
            }
            if (bf3_gemm(d) && gate == nullptr) {      // split precision: one launch per product, the bias gradient on its own
                for (int i = 0; i < np; ++i) {
                    if (int rc = gemm_reduced(d, gs, true, false, H, 4 * H, rows, pa[i], H, pb[i], 4 * H, pc[i], 4 * H, nullptr, true)) return rc;
                    if (ps[i] != nullptr)
                        if (int rc = colsum_accumulate(gs, pb[i], rows, 4 * H, 4 * H, ps[i])) return rc;
                }
                np = 0;
                continue;
            }
            if (np + 2 > group_max || np + 2 > GEMM_GROUP_MAX || l + 1 == L) {
                bool direct = true;
                for (int i = 0; i < np; ++i) direct = direct && gemm_f32_tn_group_ok(H, 4 * H, rows, pa[i], H, pb[i], 4 * H);
                if (direct) {
                    if (int rc = gemm_f32_tn_group(gs, np, H, 4 * H, rows, pa, H, pb, 4 * H, pc, 4 * H, ps, true, gate, need, gate_err)) return rc;
                } else {      // (operands the LDS-free kernel cannot address: the general GEMM, one product per launch)
                    for (int i = 0; i < np; ++i)
                        if (int rc = gemm_f32(gs, true, false, H, 4 * H, rows, pa[i], H, pb[i], 4 * H, pc[i], 4 * H, nullptr, true, ps[i],
                                              gate, need, gate_err)) return rc;
                }
                np = 0;
            }
        }
        if (dz_rows <= 0) return AMDSPEECH_OK;
        if (bf16p_layout_on(d) && gate == nullptr && dz_rows >= 256)
            return bf16p_dx(gs, bf16p_bufs(d, ws + lo.bfs), dz_rows, H, ws + lo.dg + r0 * 4 * H, kernels, ws + lo.dz0 + r0 * H);
        if (bf3_gemm(d) && gate == nullptr)
            return gemm_reduced(d, gs, false, true, dz_rows, H, 4 * H, ws + lo.dg + r0 * 4 * H, 4 * H, kernels, 4 * H, ws + lo.dz0 + r0 * H, H,
                                nullptr, false);
        return gemm_f32(gs, false, true, dz_rows, H, 4 * H, ws + lo.dg + r0 * 4 * H, 4 * H, kernels, 4 * H,
                        ws + lo.dz0 + r0 * H, H, nullptr, false, nullptr, gate, need, gate_err);
    };
    // chunk c covers frames [T*(nch-1-c)/nch, T*(nch-c)/nch): the chain walks time downwards, and every layer
    // has finished frame t after diagonal (T-1-t) + (L-1)
    if (flow) {
        unsigned* err = reinterpret_cast<unsigned*>(ws + lo.sync);
        int* progress = reinterpret_cast<int*>(err) + 8;
        unsigned* tickets = err + 16;
        if (!(d->flags & AMDSPEECH_LSTM_ARMED))      // (else: lstm_fwd has prepared them beside its kernel)
            if (int rc = flow_fill_bwd_panels(s, d, ws, lo, head != nullptr)) return rc;
        AS_CHECK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(progress), T, 8, s));
        AS_CHECK_HIP(hipMemsetAsync(tickets, 0, 16 * sizeof(unsigned), s));      // (+ the workers' eight item counters behind them)
#if FLOW2_CHECK_ORDER
        AS_CHECK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(a.dg), (int)FLOW_SENTINEL, (size_t)L * T * B * 4 * H, s));
        AS_CHECK_HIP(hipMemsetAsync(ws + lo.pdown, 0, (lo.total - lo.pdown) * sizeof(float), s));
#endif
        FlowBwdArgs fb;
        fb.wq = a.wq; fb.cs = a.cs; fb.gates = a.gates; fb.dg = a.dg; fb.dztop = a.dztop;
        fb.prec = ws + lo.prec; fb.pdown = ws + lo.pdown;
        fb.nprog = nmt; fb.prog_slack = 0;
        fb.dxh = ws + lo.dxh; fb.tickets = tickets; fb.lengths = lengths; fb.err = err; fb.progress = progress;
        fb.T = T; fb.B = B; fb.H = H; fb.L = L; fb.drop = dc;
        fb.limit = (d->flags & AMDSPEECH_LSTM_INJECT_TIMEOUT) ? 0ull : 100000000ull + (unsigned long long)T * 10000ull;      // (INJECT_TIMEOUT: tests)
        fb.trace = dev_trace_ptr();                                       // (development builds only; nullptr otherwise)
        fb.trace_layer = dev_knob("AMDSPEECH_TRACE_LAYER", L - 1);
        fb.cf = CtcFlow{}; fb.cf_on = 0;
        if (head != nullptr) {
            const int nfw = ctc_head_plan(d, head->C, head->U);
            AS_CHECK_ARG(nfw > 0, "lstm_bwd_ctc: this shape does not take the fused CTC head (amdspeech_lstm_ctc_fusable)");
            fb.cf = ctc_head_args(d, head, ws, lo, nullptr, nfw); fb.cf_on = 1;
            fb.cf.lengths = lengths; fb.cf.err = err; fb.cf.limit = fb.limit;
        }
        void (*bk)(FlowBwdArgs);
        if (head != nullptr || dev_knob("AMDSPEECH_FORCE_CF", 0) != 0) {      // the instantiations with the CTC head's leader (dev knob: without a head)
            if (d->precision == 2) bk = H == 256 ? lstm_bwd_flow2<2, 2, true> : lstm_bwd_flow2<4, 2, true>;
            else if (d->precision == 1) bk = H == 256 ? lstm_bwd_flow2<2, 1, true> : lstm_bwd_flow2<4, 1, true>;
            else bk = H == 128 ? lstm_bwd_flow2<1, 0, true> : (H == 256 ? lstm_bwd_flow2<2, 0, true> : (H == 384 ? lstm_bwd_flow2<3, 0, true> : lstm_bwd_flow2<4, 0, true>));
        } else if (d->precision == 2)      // (flow_shape_ok: H = 256 or 512 in the reduced precisions)
            bk = H == 256 ? lstm_bwd_flow2<2, 2> : lstm_bwd_flow2<4, 2>;
        else if (d->precision == 1)
            bk = H == 256 ? lstm_bwd_flow2<2, 1> : lstm_bwd_flow2<4, 1>;
        else
            bk = H == 128 ? lstm_bwd_flow2<1, 0> : (H == 256 ? lstm_bwd_flow2<2, 0> : (H == 384 ? lstm_bwd_flow2<3, 0> : lstm_bwd_flow2<4, 0>));
        // two dG tiles, the dh reduction buffer, the stash, the down product's per-wave tiles (double-buffered)
        size_t lds = ((size_t)2 * 1024 + 2 * 8 * 256 + (FLOW2_WINDOW ? 2 : 1) * 8 * (H / 128) * 256 + 3 * 1024) * sizeof(float);      // (+ the partners' tiles, FLOW2_Q = 4)
        const size_t lds_workers = (size_t)2 * 2 * 2 * BK * LDS_LD * sizeof(float);         // two GEMM teams per workgroup
        if (lds < lds_workers) lds = lds_workers;
        if (lds < (size_t)2 * CF_LEAD_TEAM_FLOATS * sizeof(float)) lds = (size_t)2 * CF_LEAD_TEAM_FLOATS * sizeof(float);      // (ctc_leader's two teams)
        AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        // AMDSPEECH_FLOW_GEMM = "pieces:percent": the weight-gradient GEMMs of the LAST `percent` % of the frames
        // (the first the recurrence finishes) are computed INSIDE the kernel, in `pieces` chunks, by the workgroups
        // of the XCDs that carry no recurrence group (bwd_gemm_worker); 0:0 leaves all of them to the launches below.
        static int pieces = -1, percent = 0;
        if (pieces < 0) {
            // measured at cfg2 with lstm_bwd_flow2 and the LDS-free worker tiles (dK only, see w_dz0): ms per step at 28 / 34 / 40 /
            // 44 / 48 % = 16.04 / 15.69 / 15.44-15.73 / 15.93 / 16.32 -- past ~40 % the kernel waits for its workers, steeply
            // round 6 (Q = 4 kernel, fused head, pieces:percent -> ms per step, two alternations on one box): 4:35 12.02 / 11.98, 8:38 11.92 /
            // 11.93, 8:40 12.14 / 12.13, 8:42 12.29, 6:40 12.15 -- eight chunks release the first frames to the workers 0.24 ms earlier
            pieces = 8; percent = 38;
            if (const char* e = dev_knob_str("AMDSPEECH_FLOW_GEMM")) {
                pieces = atoi(e);
                if (const char* q = strchr(e, ':')) percent = atoi(q + 1);
            }
            if (pieces < 0) pieces = 0;
            if (percent < 0) percent = 0;
            if (percent > 90) percent = 90;
        }
        // (the XCD-local placement needs every CU of the XCDs it uses, so the 24+8-CUs-per-XCD partition cannot
        //  be used next to it; the GEMMs follow the kernel -- AMDSPEECH_FLOW_GEMM is only honoured with FLOW_XCD=0)
        const bool overlap = false;
        // in-kernel workers exist when some XCD carries no recurrence group; they take the LAST `percent` % of the
        // frames (the first the recurrence finishes), the host-launched GEMMs the rest after the kernel
        const bool workers = pieces > 0 && percent > 0 && T >= 64 && L * nmt < 8 && H % 128 == 0;
        // (split precision: the recurrence is ~1 us per step shorter, the f32 worker GEMMs are not)
        // (fused CTC head: the teams that run ctc_leader first join the weight-gradient work ~1 ms late -- 30 / 32 / 34 / 36 / 38 % ->
        //  12.43 / 12.47 / 12.38 / 12.30 / 12.56 ms per step on one box, the separate launches 12.67 - 12.88 there)
        // (round 6, reduced precisions WITH the head: the recurrence is a third shorter, the f32 worker products are not, and the leader
        //  teams still join ~1 ms late -- 16 / 20 / 24 / 28 % -> 8.20 / 8.32 / 8.54 / 9.12 ms per step in bf16x3 at 3x512 on one box (no
        //  workers: 8.53); round 5 ran it at 28 %: the "regression" of that mode against round 4's 8.60)
        const int share = dev_knob_str("AMDSPEECH_FLOW_GEMM") ? percent
                          : (d->precision != 0 ? (head != nullptr ? percent / 2 - 1 : percent * 3 / 4) : percent);
        fb.z = ws + lo.z; fb.hs = ws + lo.hs; fb.kernels = kernels; fb.dk = dkernels; fb.dbias = dbiases; fb.dz0 = ws + lo.dz0;
        fb.kstride = kstride; fb.bstride = bstride;
        static const int worker_dz0 = dev_knob("AMDSPEECH_FLOW_WORKER_DZ0", 0);
        // dZ_0 = dG_0 . W_ih0^T by the bottom layer's groups (default since round 4: with the 2-D down product the kernel pays 0.2 ms
        // for it and the 0.61 ms GEMM + the mask launch behind the kernel go: 13.45 -> 13.36 ms per step; rounds 2-3, with the 32-way
        // exchange of down partials: a draw, off).  AMDSPEECH_FLOW_DZ0=0: the GEMM after the kernel.
        static const int dz0_in = runtime_switch("AMDSPEECH_FLOW_DZ0", 1);
        fb.dz0_inkernel = dz0_in ? 1 : 0;
        fb.w_dz0 = fb.dz0_inkernel ? 0 : (workers ? worker_dz0 : 1);
        fb.w_mode = dev_knob("AMDSPEECH_FLOW_WORKER_MODE", 0);
        fb.w_pieces = workers ? pieces : 0;
        static const int dyn = runtime_switch("AMDSPEECH_FLOW_WORKER_DEAL", -1);      // -1: with the fused CTC head only; 0 / 1: never / always
        fb.w_counters = (workers && pieces <= 8 && (dyn > 0 || (dyn < 0 && head != nullptr))) ? tickets + 8 : nullptr;
        fb.w_t0 = workers ? T - (int)((long)T * share / 100) : T;
        if (fb.w_t0 < 2) fb.w_t0 = 2;
        int t_split = T;
        hipStream_t ks = s;
        if (overlap) {
            t_split = T - (int)((long)T * percent / 100);
            if (t_split < 2) t_split = 2;
            ks = g_chain;
            AS_CHECK_HIP(hipEventRecord(g_ev_a, s));
            AS_CHECK_HIP(hipStreamWaitEvent(g_chain, g_ev_a, 0));
            AS_CHECK_HIP(hipStreamWaitEvent(g_gemm, g_ev_a, 0));
        }
        prof_begin(1, ks);
        hipLaunchKernelGGL(bk, dim3(256), dim3(512), lds, ks, fb);      // one workgroup per CU; each finds its group by XCC_ID
        prof_end(1, ks, T + L - 1);
        if (int rc = flow_mark_postlaunch(ks, ws, 1 | (fb.dz0_inkernel ? 2 : 0))) return rc;      // (amdspeech_lstm_beside_tail)
        {   // algorithmic flops of this launch: L recurrent + (L - 1) down products (+ dZ_0 when the layer-0 groups form it) per
            // frame, and the weight-gradient products of the frames [w_t0, T) its worker workgroups take
            const double prod = 2.0 * B * 4 * H * H;
            const double wframes = fb.w_pieces > 0 ? (double)(T - fb.w_t0) : 0.0;
            prof_flops(1, (double)T * (2 * L - 1 + (fb.dz0_inkernel ? 1 : 0)) * prod,
                       wframes * (L * 2.0 * prod + (fb.w_dz0 ? prod : 0.0)));
        }
        AS_CHECK_LAUNCH();
        if (overlap) {
            for (int i = 0; i < pieces; ++i) {       // latest frames first: that is the order they are finished in
                const int tb = T - (int)((long)(T - t_split) * i / pieces), ta = T - (int)((long)(T - t_split) * (i + 1) / pieces);
                if (tb > ta)
                    if (int rc = weight_grads(g_gemm, ta, tb, progress, ta - 2)) return rc;
            }
            AS_CHECK_HIP(hipEventRecord(g_ev_b, g_chain));
            AS_CHECK_HIP(hipStreamWaitEvent(s, g_ev_b, 0));
            if (int rc = weight_grads(s, 0, t_split, nullptr, 0)) return rc;       // the rest, on the whole chip
            AS_CHECK_HIP(hipEventRecord(g_ev_c, g_gemm));
            AS_CHECK_HIP(hipStreamWaitEvent(s, g_ev_c, 0));
        } else {
            // what the workers did not take (dZ_0: nothing if the layer-0 groups formed it, else the frames the workers left)
            if (int rc = weight_grads(s, 0, workers ? fb.w_t0 : T, nullptr, 0, fb.dz0_inkernel ? 0 : (fb.w_dz0 ? -1 : T))) return rc;
        }
        if (d->keep_in < 1.0f && !fb.dz0_inkernel) {     // the layer-0 input dropout mask on dZ_0
            const long n = (long)T * B * H;
            hipLaunchKernelGGL(apply_zmult_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, ws + lo.dz0, n, dc, 0);
            AS_CHECK_LAUNCH();
        }
        return AMDSPEECH_OK;
    }
    if (use_big1_bwd(d)) {
        if (defer_big) return AMDSPEECH_OK;      // (amdspeech_lstm_bwd_pair: the layers of the two stacks run together)
        const BigBwdStack one{d, ws, kernels, kstride, dkernels, dbiases, bstride, lengths};
        return big1_bwd_layers(s, 1, &one);      // (one stack alone: the one-XCD groups are the faster backward kernel too, see lstm_big_bwd.h)
    }
    if (!flow && use_big_fwd(d) && (size_t)2 * nmt * 64 * 64 * 1024 < (1ull << 32)) {
        // H = 1024: one weight-stationary launch per layer (lstm_bwd_big), top first; after each, ONE GEMM hands the finished
        // layer's gradient down: dX_{l-1} [T*B, H] = dG_l [T*B, 4H] . K_l[0:H, :]^T, into the (by then dead) dztop buffer
        const size_t TB = (size_t)T * B;
        unsigned* err = reinterpret_cast<unsigned*>(ws + lo.sync);
        BigBwdArgs b2;
        b2.wq = a.wq; b2.cs = a.cs; b2.gates = a.gates; b2.dg = a.dg; b2.dup = ws + lo.dztop; b2.lengths = lengths;
        const size_t pring_floats = (size_t)2 * nmt * 2 * 32 * 32 * 256, xring_floats = (size_t)2 * nmt * 64 * 1024;
        b2.pring = ws + lo.bigring; b2.xring = ws + lo.bigring + pring_floats; b2.err = err; b2.tickets = err + 16;
        b2.T = T; b2.B = B; b2.H = H; b2.L = L; b2.drop = dc;
        b2.limit = (d->flags & AMDSPEECH_LSTM_INJECT_TIMEOUT) ? 0ull : 100000000ull + (unsigned long long)T * 10000ull;      // (INJECT_TIMEOUT: tests)
        for (int l = L - 1; l >= 0; --l) {
            AS_CHECK_HIP(hipMemsetAsync(ws + lo.bigring, 0, (pring_floats + xring_floats) * sizeof(float), s));
            AS_CHECK_HIP(hipMemsetAsync(b2.tickets, 0, 8 * sizeof(unsigned), s));
            b2.layer = l;
            prof_begin(1, s, L - 1 - l);
            if (d->precision == 2) hipLaunchKernelGGL(lstm_bwd_big<2>, dim3(256), dim3(512), 0, s, b2);
            else if (d->precision == 1) hipLaunchKernelGGL(lstm_bwd_big<1>, dim3(256), dim3(512), 0, s, b2);
            else hipLaunchKernelGGL(lstm_bwd_big<0>, dim3(256), dim3(512), 0, s, b2);
            prof_end(1, s, T * L, L - 1 - l);
            if (bf16p_layout_on(d)) {      // plain bf16 through operand copies: everything this layer owes, now (dZ_0 for the bottom layer)
                if (int rc = bf16p_layer_bwd(s, bf16p_bufs(d, ws + lo.bfs), (int)TB, H, ws + lo.z + (size_t)l * TB * H,
                                             ws + lo.hs + (size_t)l * (T + 1) * B * H, ws + lo.dg + (size_t)l * TB * 4 * H, kernels + l * kstride,
                                             l > 0 ? ws + lo.dztop : ws + lo.dz0, dkernels + l * kstride, dbiases + l * bstride)) return rc;
                continue;
            }
            if (l > 0)
                if (int rc = bf3_gemm(d) ? gemm_reduced(d, s, false, true, (int)TB, H, 4 * H, ws + lo.dg + (size_t)l * TB * 4 * H, 4 * H,
                                                        kernels + l * kstride, 4 * H, ws + lo.dztop, H, nullptr, false)
                                         : gemm_f32_plain(s, false, true, (int)TB, H, 4 * H, ws + lo.dg + (size_t)l * TB * 4 * H, 4 * H,
                                                          kernels + l * kstride, 4 * H, ws + lo.dztop, H, nullptr, false)) return rc;
        }
        AS_CHECK_LAUNCH();
        if (!bf16p_layout_on(d))
            if (int rc = weight_grads(s, 0, T, nullptr, 0)) return rc;
        if (d->keep_in < 1.0f) {     // the layer-0 input dropout mask on dZ_0
            const long n = (long)T * B * H;
            hipLaunchKernelGGL(apply_zmult_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, ws + lo.dz0, n, dc, 0);
            AS_CHECK_LAUNCH();
        }
        return AMDSPEECH_OK;
    }
    if (hoist) {
        // layer by layer, top first: T launches of the recurrent product, then ONE GEMM hands the finished layer's
        // gradient down: dX_{l-1} [T*B, H] = dG_l [T*B, 4H] . K_l[0:H, :]^T, into the (by then dead) dztop buffer
        a.hoist = 1; a.mt0 = 0;
        dim3 grid(H / 16, 1, nmt), block(bwd_nw * 64);
        const size_t TB = (size_t)T * B;
        prof_begin(1, s);
        for (int l = L - 1; l >= 0; --l) {
            a.l0 = l;
            for (int dd = 0; dd < T; ++dd) {
                a.d = dd;
                hipLaunchKernelGGL(kern, grid, block, 0, s, a);
            }
            if (l > 0)
                if (int rc = gemm_f32(s, false, true, (int)TB, H, 4 * H, ws + lo.dg + (size_t)l * TB * 4 * H, 4 * H,
                                      kernels + l * kstride, 4 * H, ws + lo.dztop, H, nullptr, false)) return rc;
        }
        prof_end(1, s, T * L);
        AS_CHECK_LAUNCH();
        if (int rc = weight_grads(s, 0, T, nullptr, 0)) return rc;
        if (d->keep_in < 1.0f) {     // the layer-0 input dropout mask on dZ_0
            const long n = (long)T * B * H;
            hipLaunchKernelGGL(apply_zmult_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, ws + lo.dz0, n, dc, 0);
            AS_CHECK_LAUNCH();
        }
        return AMDSPEECH_OK;
    }
    int nch = 0, nside = 0;
    if (chains == 1 && T >= 64) dk_overlap_plan(&nch, &nside);
    // (CU-masked streams are "blocking" streams: against the legacy NULL stream every launch on them pays an
    // implicit cross-stream synchronisation -- measured 20 us per launch -- so the caller must be on a real stream)
    if (nside > 0 && (s == nullptr || overlap_init() != 1)) nside = 0;
    hipStream_t chain_stream = s;
    if (nside > 0) {
        chain_stream = g_chain;
        AS_CHECK_HIP(hipEventRecord(g_ev_a, s));
        AS_CHECK_HIP(hipStreamWaitEvent(g_chain, g_ev_a, 0));
    } else if (chains == 2) {
        if (int rc = side_stream_init()) return rc;
        AS_CHECK_HIP(hipEventRecord(g_fork, s));
        AS_CHECK_HIP(hipStreamWaitEvent(g_side, g_fork, 0));
    }
    prof_begin(1, chain_stream);
    int next_chunk = 0;
    for (int c = 0; c < chains; ++c) {
        const int t0 = c * nmt / chains, t1 = (c + 1) * nmt / chains;
        dim3 grid(H / 16, L, t1 - t0), block((bf3 ? 8 : bwd_nw) * 64);
        hipStream_t cs = c == 0 ? chain_stream : g_side;
        a.mt0 = t0;
        for (int dd = 0; dd < T + L - 1; ++dd) {
            a.d = dd;
            hipLaunchKernelGGL(kern, grid, block, 0, cs, a);
            if (next_chunk < nside) {
                const int ta = (int)((long)T * (nch - 1 - next_chunk) / nch), tb = (int)((long)T * (nch - next_chunk) / nch);
                if (dd == (T - 1 - ta) + (L - 1)) {
                    AS_CHECK_HIP(hipEventRecord(g_ev_b, g_chain));
                    AS_CHECK_HIP(hipStreamWaitEvent(g_gemm, g_ev_b, 0));
                    if (int rc = weight_grads(g_gemm, ta, tb, nullptr, 0)) return rc;
                    ++next_chunk;
                }
            }
        }
    }
    prof_end(1, chain_stream, T + L - 1);
    AS_CHECK_LAUNCH();
    if (nside > 0) {
        AS_CHECK_HIP(hipEventRecord(g_ev_a, g_chain));
        AS_CHECK_HIP(hipStreamWaitEvent(s, g_ev_a, 0));
        if (int rc = weight_grads(s, 0, (int)((long)T * (nch - nside) / nch), nullptr, 0)) return rc;   // the rest, whole chip
        AS_CHECK_HIP(hipEventRecord(g_ev_c, g_gemm));
        AS_CHECK_HIP(hipStreamWaitEvent(s, g_ev_c, 0));
    } else {
        if (chains == 2) {
            AS_CHECK_HIP(hipEventRecord(g_join, g_side));
            AS_CHECK_HIP(hipStreamWaitEvent(s, g_join, 0));
        }
        if (int rc = weight_grads(s, 0, T, nullptr, 0)) return rc;
    }
    if (d->keep_in < 1.0f) {     // the layer-0 input dropout mask on dZ_0
        const long n = (long)T * B * H;
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
            for (int k = 0; k < PROF_SEGS; ++k)
                for (int j = 0; j < 2; ++j) AS_CHECK_HIP(hipEventCreate(&g_prof_ev[i][k][j]));
    }
    if (!on && g_prof_on) {
        for (int i = 0; i < 2; ++i)
            for (int k = 0; k < PROF_SEGS; ++k)
                for (int j = 0; j < 2; ++j) (void)hipEventDestroy(g_prof_ev[i][k][j]);
        g_prof_valid[0] = g_prof_valid[1] = false;
    }
    g_prof_on = on != 0;
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_profile_get(int which, float* elapsed_ms, int* time_steps) {
    AS_CHECK_ARG(which == 0 || which == 1, "profile_get: which must be 0 or 1");
    AS_CHECK_ARG(elapsed_ms && time_steps, "profile_get: null pointer");
    AS_CHECK_ARG(g_prof_on && g_prof_valid[which], "profile_get: nothing recorded (enable profiling first)");
    float total = 0.f;
    for (int k = 0; k < g_prof_nseg[which]; ++k) {
        float ms = 0.f;
        AS_CHECK_HIP(hipEventSynchronize(g_prof_ev[which][k][1]));
        AS_CHECK_HIP(hipEventElapsedTime(&ms, g_prof_ev[which][k][0], g_prof_ev[which][k][1]));
        total += ms;
    }
    *elapsed_ms = total;
    *time_steps = g_prof_launches[which];
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_profile_get_flops(int which, double* recurrence_flops, double* other_flops) {
    AS_CHECK_ARG(which == 0 || which == 1, "profile_get_flops: which must be 0 or 1");
    AS_CHECK_ARG(recurrence_flops && other_flops, "profile_get_flops: null pointer");
    *recurrence_flops = g_prof_flops[which][0];
    *other_flops = g_prof_flops[which][1];
    return AMDSPEECH_OK;
}

// The bytes a workspace for sequences of UP TO d->T frames needs.  ops.LstmWorkspace.prefix lays one allocation out again for every
// shorter run length, and two regions exist only below a sequence length (the 32-bit buffer resources of the whole-sequence kernels:
// flow_shape_ok, fwd_workers_max) -- a prefix just below such a threshold can need MORE than the full length above it.  With the
// set of regions fixed the size is monotone in T, so the maximum over T' <= T is taken at T or at the last T' of either set.
extern "C" size_t amdspeech_lstm_workspace_bytes(const amdspeech_lstm_desc* d) {
    if (check_desc(d)) return 0;
    size_t need = lstm_layout(d).total;
    amdspeech_lstm_desc q = *d;
    auto last_with = [&](auto pred) {      // the largest T' <= d->T with pred (true below a threshold, false above), or 0
        q.T = d->T;
        if (pred(&q)) return d->T;
        int lo = 0, hi = d->T;             // pred(lo) true (or lo == 0), pred(hi) false
        while (hi - lo > 1) { q.T = lo + (hi - lo) / 2; if (pred(&q)) lo = q.T; else hi = q.T; }
        return lo;
    };
    const int cand[2] = {last_with([](const amdspeech_lstm_desc* x) { return flow_shape_ok(x); }),
                         last_with([](const amdspeech_lstm_desc* x) { return flow_shape_ok(x) && fwd_workers_max(x) > 0; })};
    for (int c : cand)
        if (c > 0 && c < d->T) { q.T = c; const size_t n = lstm_layout(&q).total; if (n > need) need = n; }
    return need * sizeof(float);
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

// The multipliers the kernels above apply, as a tensor (tests feed them to the oracle's DropoutWrapper restatement): the SAME
// zmult() with the other mask switched off.
__global__ void export_zmult_kernel(float* out, long n, DropCfg c, int lp) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = zmult(c, lp, (uint32_t)i);
}
extern "C" int amdspeech_lstm_dropout_multipliers(void* stream, const amdspeech_lstm_desc* d, int which, int layer, float* out) {
    if (int rc = check_desc(d)) return rc;
    AS_CHECK_ARG(out != nullptr && (which == 0 || which == 1) && layer >= 0 && layer < d->L,
                 "lstm_dropout_multipliers: which must be 0 (input mask) or 1 (output mask), layer in [0, L)");
    DropCfg dc{which == 0 ? d->keep_in : 1.0f, which == 1 ? d->keep_out : 1.0f, d->seed, d->L};
    const long n = (long)d->T * d->B * d->H;
    hipLaunchKernelGGL(export_zmult_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), out, n, dc,
                       which == 0 ? layer : layer + 1);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_lstm_workspace_release(void* stream, void* ws) {
    AS_CHECK_ARG(ws != nullptr, "lstm_workspace_release: null workspace");
    return flow_arm_release(static_cast<hipStream_t>(stream), ws);
}

// (A CU-masked stream confined to the idle XCDs would be the obvious tool, and does not exist: hipExtStreamCreateWithCUMask
// applies ONE per-XCD CU pattern to all eight XCDs -- tools/cumask_probe.hip: a mask with only the bits of "XCDs 6 and 7" set
// enables all 256 CUs.  The caller's kernels are dealt to every XCD like any other; see amdspeech.h for what that means.)
extern "C" int amdspeech_lstm_beside_forward(void* stream, const void* ws) {
    AS_CHECK_ARG(ws != nullptr, "lstm_beside_forward: null workspace");
    std::lock_guard<std::mutex> lock(g_arm_mutex);
    auto it = g_arm.find(ws);
    if (it == g_arm.end() || it->second.idle_xcds <= 0 || it->second.pre == nullptr) return 0;
    AS_CHECK_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), it->second.pre, 0));
    return it->second.idle_xcds;
}

extern "C" int amdspeech_lstm_beside_tail(void* stream, const void* ws) {
    AS_CHECK_ARG(ws != nullptr, "lstm_beside_tail: null workspace");
    std::lock_guard<std::mutex> lock(g_arm_mutex);
    auto it = g_arm.find(ws);
    if (it == g_arm.end() || it->second.post_flags == 0 || it->second.post == nullptr) return 0;
    AS_CHECK_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), it->second.post, 0));
    return it->second.post_flags;
}

extern "C" int amdspeech_lstm_status(const amdspeech_lstm_desc* d, void* ws) {
    if (int rc = check_desc(d)) return rc;
    AS_CHECK_ARG(ws != nullptr, "lstm_status: null workspace");
    const LstmLayout lo = lstm_layout(d);
    unsigned err = 0;
    AS_CHECK_HIP(hipMemcpy(&err, static_cast<float*>(ws) + lo.sync, sizeof(err), hipMemcpyDeviceToHost));
    if (err != 0) {
        flow_xw_forget(ws);
        set_error("LSTM dataflow kernels: a bounded wait timed out (flags 0x%x: 1 = forward, 2 = backward -- the workgroups of "
                  "one launch were not all resident; 4 = a weight-gradient GEMM gave up waiting for the backward kernel, "
                  "8 = an x-product worker of the forward kernel gave up waiting for the layer below, "
                  "32 = the fused CTC head gave up waiting for the top layer, "
                  "e.g. under a tool that serialises kernels: set AMDSPEECH_FLOW_GEMM=0:0); results of this step are invalid", err);
        return AMDSPEECH_ETIMEOUT;
    }
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_lstm_fwd(void* stream, const amdspeech_lstm_desc* d, void* ws, const float* kernels,
                                  long kernel_stride, const float* biases, long bias_stride,
                                  const int* lengths, const float* h0, const float* c0) {
    return lstm_fwd(static_cast<hipStream_t>(stream), d, static_cast<float*>(ws), kernels, kernel_stride,
                    biases, bias_stride, lengths, h0, c0);
}

extern "C" int amdspeech_lstm_pair_fusable(const amdspeech_lstm_desc* d) {
    return d != nullptr && check_desc(d) == AMDSPEECH_OK && !(d->flags & AMDSPEECH_LSTM_PER_DIAGONAL) && !use_flow(d) && use_big1_fwd(d);
}

extern "C" int amdspeech_lstm_fwd_pair(void* stream, const amdspeech_lstm_desc* d_a, void* ws_a, const float* kernels_a, const float* biases_a,
                                       const amdspeech_lstm_desc* d_b, void* ws_b, const float* kernels_b, const float* biases_b,
                                       long kernel_stride, long bias_stride, const int* lengths, const float* h0_a, const float* c0_a) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    AS_CHECK_ARG(d_a && d_b, "lstm_fwd_pair: null descriptor");
    const bool together = d_a->T == d_b->T && d_a->B == d_b->B && d_a->H == d_b->H && d_a->L == d_b->L && d_a->precision == d_b->precision &&
                          ws_a != ws_b && amdspeech_lstm_pair_fusable(d_a) && amdspeech_lstm_pair_fusable(d_b);
    if (int rc = lstm_fwd(s, d_a, static_cast<float*>(ws_a), kernels_a, kernel_stride, biases_a, bias_stride, lengths, h0_a, c0_a, nullptr, together)) return rc;
    if (int rc = lstm_fwd(s, d_b, static_cast<float*>(ws_b), kernels_b, kernel_stride, biases_b, bias_stride, lengths, nullptr, nullptr, nullptr, together)) return rc;
    if (!together) return AMDSPEECH_OK;
    const BigStack two[2] = {{d_a, static_cast<float*>(ws_a), kernels_a, kernel_stride, biases_a, bias_stride, lengths},
                             {d_b, static_cast<float*>(ws_b), kernels_b, kernel_stride, biases_b, bias_stride, lengths}};
    return big_fwd_layers(s, 2, two);
}

extern "C" int amdspeech_lstm_bwd_pair(void* stream, const amdspeech_lstm_desc* d_a, void* ws_a, const float* kernels_a, float* dkernels_a,
                                       float* dbiases_a, const amdspeech_lstm_desc* d_b, void* ws_b, const float* kernels_b, float* dkernels_b,
                                       float* dbiases_b, long kernel_stride, long bias_stride, const int* lengths) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    AS_CHECK_ARG(d_a && d_b, "lstm_bwd_pair: null descriptor");
    const bool together = d_a->T == d_b->T && d_a->B == d_b->B && d_a->H == d_b->H && d_a->L == d_b->L && d_a->precision == d_b->precision &&
                          ws_a != ws_b && amdspeech_lstm_pair_fusable(d_a) && amdspeech_lstm_pair_fusable(d_b) && use_big1_bwd(d_a);
    if (int rc = lstm_bwd(s, d_a, static_cast<float*>(ws_a), kernels_a, kernel_stride, dkernels_a, dbiases_a, bias_stride, lengths, nullptr, together)) return rc;
    if (int rc = lstm_bwd(s, d_b, static_cast<float*>(ws_b), kernels_b, kernel_stride, dkernels_b, dbiases_b, bias_stride, lengths, nullptr, together)) return rc;
    if (!together) return AMDSPEECH_OK;
    const BigBwdStack two[2] = {{d_a, static_cast<float*>(ws_a), kernels_a, kernel_stride, dkernels_a, dbiases_a, bias_stride, lengths},
                                {d_b, static_cast<float*>(ws_b), kernels_b, kernel_stride, dkernels_b, dbiases_b, bias_stride, lengths}};
    return big1_bwd_layers(s, 2, two);
}

extern "C" int amdspeech_lstm_bwd(void* stream, const amdspeech_lstm_desc* d, void* ws, const float* kernels,
                                  long kernel_stride, float* dkernels, float* dbiases, long bias_stride,
                                  const int* lengths) {
    return lstm_bwd(static_cast<hipStream_t>(stream), d, static_cast<float*>(ws), kernels, kernel_stride,
                    dkernels, dbiases, bias_stride, lengths);
}

/* The fused CTC head (ctc_flow.h) */
extern "C" int amdspeech_lstm_ctc_fusable(const amdspeech_lstm_desc* d, int C, int U) {
    if (check_desc(d) != AMDSPEECH_OK) return 0;
    return ctc_head_plan(d, C, U) > 0 ? 1 : 0;
}
static int check_head(const amdspeech_ctc_head* h, bool bwd) {
    AS_CHECK_ARG(h != nullptr, "lstm_*_ctc: null head");
    AS_CHECK_ARG(h->w_out && h->b_out && h->logits && h->dense_labels && h->loss && h->ctc_ws && (!bwd || h->dlogits),
                 "lstm_*_ctc: null pointer in the head");
    AS_CHECK_ARG(((uintptr_t)h->w_out % 16) == 0 && ((uintptr_t)h->ctc_ws % 256) == 0, "lstm_*_ctc: W_o must be 16-byte, the CTC workspace 256-byte aligned");
    return AMDSPEECH_OK;
}
extern "C" int amdspeech_lstm_fwd_ctc(void* stream, const amdspeech_lstm_desc* d, void* ws, const float* kernels,
                                      long kernel_stride, const float* biases, long bias_stride, const int* lengths,
                                      const float* h0, const float* c0, const amdspeech_ctc_head* head) {
    if (int rc = check_head(head, false)) return rc;
    return lstm_fwd(static_cast<hipStream_t>(stream), d, static_cast<float*>(ws), kernels, kernel_stride,
                    biases, bias_stride, lengths, h0, c0, head);
}
extern "C" int amdspeech_lstm_bwd_ctc(void* stream, const amdspeech_lstm_desc* d, void* ws, const float* kernels,
                                      long kernel_stride, float* dkernels, float* dbiases, long bias_stride,
                                      const int* lengths, const amdspeech_ctc_head* head) {
    if (int rc = check_head(head, true)) return rc;
    return lstm_bwd(static_cast<hipStream_t>(stream), d, static_cast<float*>(ws), kernels, kernel_stride,
                    dkernels, dbiases, bias_stride, lengths, head);
}
