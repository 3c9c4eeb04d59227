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
static bool bf16p_layout_on(const amdspeech_lstm_desc* d) {
    static const int env = runtime_switch("AMDSPEECH_BF16_PACKED", 1);      // 0: gemm_bf16 (f32 operands converted on the way into LDS: round 4)
    return env != 0 && d->precision == 2 && d->H == 1024 && ((long)d->T * d->B) % 64 == 0 && (long)d->T * d->B >= 256;
}
struct Bf16pBufs { unsigned short *zb, *wtb, *dgb, *wb, *zht, *dgt; char* partial; size_t partial_bytes; };
static size_t bf16p_scratch_floats(const amdspeech_lstm_desc* d) {
    const size_t TB = (size_t)d->T * d->B, H = d->H;
    const size_t bytes = TB * H * 2 + 4 * H * H * 2 + TB * 4 * H * 2 + H * 4 * H * 2 + 2 * H * TB * 2 + 4 * H * TB * 2 +
                         bf16p_partial_bytes(2 * (int)H, 4 * (int)H, (int)TB) + 8 * 256;
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
    b.partial_bytes = bf16p_partial_bytes(2 * (int)H, 4 * (int)H, (int)TB);
    b.partial = take(b.partial_bytes);
    return b;
}
// G[rows][4H] = Z[rows][H] . K[0:H, :] + bias
static int bf16p_xw(hipStream_t s, const Bf16pBufs& b, int rows, int H, const float* Z, const float* K, float* G, const float* bias) {
    if (int rc = bf16p_copy(s, Z, H, rows, H, false, b.zb, H, nullptr)) return rc;
    if (int rc = bf16p_copy(s, K, 4 * H, H, 4 * H, true, b.wtb, H, nullptr)) return rc;              // [H][4H] -> [4H][H]
    return bf16p_gemm(s, rows, 4 * H, H, b.zb, H, b.wtb, H, G, 4 * H, bias, false, nullptr, 0);
}
// dX[rows][H] = dG[rows][4H] . K[0:H, :]^T
static int bf16p_dx(hipStream_t s, const Bf16pBufs& b, int rows, int H, const float* dG, const float* K, float* dX) {
    if (int rc = bf16p_copy(s, dG, 4 * H, rows, 4 * H, false, b.dgb, 4 * H, nullptr)) return rc;
    if (int rc = bf16p_copy(s, K, 4 * H, H, 4 * H, false, b.wb, 4 * H, nullptr)) return rc;
    return bf16p_gemm(s, rows, H, 4 * H, b.dgb, 4 * H, b.wb, 4 * H, dX, H, nullptr, false, nullptr, 0);
}
// A whole layer's batched backward products behind its recurrence launch (all T x B rows): ONE read of dG gives its row-major
// copy (dX), its transposed copy (dK) and the bias gradient; dX[rows][H] = dG . K[0:H, :]^T; dK[2H][4H] += [Z ; Hprev]^T . dG
static int bf16p_layer_bwd(hipStream_t s, const Bf16pBufs& b, int rows, int H, const float* Z, const float* Hp, const float* dG, const float* K,
                           float* dX, float* dK, float* dbias) {
    if (int rc = bf16p_copy(s, dG, 4 * H, rows, 4 * H, true, b.dgt, rows, dbias, b.dgb)) return rc;
    if (int rc = bf16p_copy(s, K, 4 * H, H, 4 * H, false, b.wb, 4 * H, nullptr)) return rc;
    if (int rc = bf16p_gemm(s, rows, H, 4 * H, b.dgb, 4 * H, b.wb, 4 * H, dX, H, nullptr, false, nullptr, 0)) return rc;
    if (int rc = bf16p_copy(s, Z, H, rows, H, true, b.zht, rows, nullptr)) return rc;
    if (int rc = bf16p_copy(s, Hp, H, rows, H, true, b.zht + (size_t)H * rows, rows, nullptr)) return rc;
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
    }
    // lstm_bwd_big (H = 1024), ONE layer at a time: the partial-tile rings of the two XCDs of every pair, [2 slots][batch tiles]
    // [2][32][32][256 floats], and the dG tiles that cross between them, [2 slots][batch tiles][64][1024]
    o.bigring = off;
    if (!flow_shape_ok(d) && d->precision >= 0 && d->precision <= 2 && d->H == 1024 && bp / 16 <= 4)
        o.bigring = take((size_t)2 * (bp / 16) * (2 * 32 * 32 * 256 + 64 * 1024));
    // precision = 2 at H = 1024 (gemm_bf16p.hip): bf16 copies of the batched products' operands + the split-K partial tiles
    o.bfs = off;
    if (bf16p_layout_on(d)) o.bfs = take(bf16p_scratch_floats(d));
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
    if (i < 40) sync_words[i] = 0u;              // error word (+ progress words), the per-XCD tickets, [32]: the counter of flow_fill_queue_kernel
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

// Everything lstm_fwd prepares BESIDE its whole-sequence kernel (AMDSPEECH_LSTM_ARM_NEXT: the sentinels / zeros of the backward call's
// panels and of the other set of forward panels, the backward call's transposed weight pack) as ONE work-queue kernel.  As
// hipMemsetAsync launches those fills were ordinary kernels: the dispatcher deals a kernel's workgroups to all eight XCDs, the ones
// dealt to an XCD full of recurrence workgroups start when the recurrence ends, a kernel completes with its last workgroup and the
// next one of the stream starts behind it -- five launches of ~35 us each ended up BEHIND the forward kernel (seen in the trace of
// round 5: 0.1 ms between the two recurrence kernels once the CTC stage had left that gap).  Here the workgroups that do find a CU
// pull 64 KiB chunks from a counter until the work is gone; the ones that start late find it empty.
struct FillJobs {
    float* p[6]; unsigned long long n[6]; unsigned v[6]; int count;      // regions: n dwords (multiples of 64) of value v at p (256-byte aligned)
    const float* kernels; long kstride; float* wq; int H, L;             // + pack_bwd_kernel's job (wq == nullptr: none)
};
__global__ __launch_bounds__(256) void flow_fill_queue_kernel(FillJobs j, unsigned* __restrict__ next) {
    constexpr unsigned CH = 16384;      // dwords per chunk
    __shared__ unsigned s_c;
    while (true) {
        if (threadIdx.x == 0) s_c = atomicAdd(next, 1u);
        __syncthreads();
        unsigned long long c = s_c;
        __syncthreads();
        int k = 0;
        for (; k < j.count; ++k) {
            const unsigned long long nck = (j.n[k] + CH - 1) / CH;
            if (c < nck) break;
            c -= nck;
        }
        if (k < j.count) {
            const unsigned long long base = c * CH, cnt = j.n[k] - base < CH ? j.n[k] - base : CH;
            uint4* q = reinterpret_cast<uint4*>(j.p[k] + base);
            const uint4 val = make_uint4(j.v[k], j.v[k], j.v[k], j.v[k]);
            for (unsigned i = threadIdx.x; i < cnt / 4; i += 256) q[i] = val;
            continue;
        }
        if (j.wq == nullptr) return;
        const long total = (long)j.L * 2 * j.H * 4 * j.H;
        const long o0 = (long)c * CH;
        if (o0 >= total) return;
        const int NRB = 2 * j.H / 16, NKB = 4 * j.H / 16;
        for (long o = o0 + threadIdx.x; o < o0 + CH && o < total; o += 256) {      // (pack_bwd_kernel's index map)
            const int m = o & 3, lane = (o >> 2) & 63;
            long r = o >> 8;
            const int kb = r % NKB; r /= NKB;
            const int rb = r % NRB; const int l = r / NRB;
            const int row = rb * 16 + (lane & 15), col = kb * 16 + 4 * (lane >> 4) + m;
            j.wq[o] = j.kernels[l * j.kstride + (long)row * 4 * j.H + col];
        }
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

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

}  // namespace amdspeech
#include "ctc_flow.h"      // the CTC head inside the dataflow kernels (needs FLOW_SENTINEL / flow_pending above)
namespace amdspeech {

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


template <int NTW, int PR, bool CF = false>     // 16-column N tiles (and gathered producer tiles) per wave: H/16/8 = H/128; PR: 0 f32, 1 bf16x3, 2 bf16;
                                                // CF: the instantiation with the fused CTC head's leader (see lstm_fwd_flow2)
__global__ __launch_bounds__(512) void lstm_bwd_flow2(FlowBwdArgs a_in) {
    constexpr bool BF3 = PR != 0;
    constexpr int NW = 8, H = 128 * NTW, NU = H / 16, NKB = 4 * H / 16, NRB = 2 * H / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* a_lds = smem;                                                                      // [2 (step parity)][4 m][4 kq][16 i][4 g]: dG tiles as MFMA A fragments
    float (*red_r)[256] = reinterpret_cast<float (*)[256]>(smem + 2048);                     // [NW][256] partial sums of dh
    float (*stash_lds)[256] = reinterpret_cast<float (*)[256]>(smem + 2048 + NW * 256);      // [8][256] the next epilogue's forward stash
    float* qred = smem + 2048 + 2 * NW * 256;                                                 // [NW][NTW][64][4] per-wave partial tiles of the down product
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
    f32x4 wr[NTW][4], wd[NTW][4];
    {
        const float* base = a.wq + (size_t)l * NRB * NKB * 256 + lane * 4;
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nt = wave * NTW + n, kb = g * (H / 16) + ub;
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
    constexpr unsigned SLOT_BYTES = (unsigned)NU * NU * 1024u, QSLOT_BYTES = (unsigned)NU * KS * 1024u;
    const auto rp = __builtin_amdgcn_make_buffer_rsrc(a.prec + (size_t)grp * 2 * NU * NU * 256, 0, 2u * SLOT_BYTES, 0x00020000);
    const auto rq = __builtin_amdgcn_make_buffer_rsrc(a.pdown + (size_t)grp * 4 * NU * KS * 256, 0, 4u * QSLOT_BYTES, 0x00020000);
    const auto rdg = __builtin_amdgcn_make_buffer_rsrc(a.dg + (size_t)l * T * B * 4 * H, 0, (unsigned)((size_t)T * B * 4 * H * 4), 0x00020000);
    unsigned gather_off = (unsigned)(((ub * NU + wave * NTW) * 256 + lane * 4) * 4);      // + q KiB: producer wave*NTW + q
    const unsigned store_off = (unsigned)((((wave * NTW) * NU + ub) * 256 + lane * 4) * 4);     // + n*NU KiB: consumer wave*NTW + n
    bool dead = false;
    u32x4_f gp[NTW];
    auto issue = [&](decltype(rp) rs, u32x4_f (&buf)[NTW], int slot) {
#pragma unroll
        for (int q = 0; q < NTW; ++q)
            buf[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, gather_off + (unsigned)(q * 1024), (unsigned)slot * SLOT_BYTES, FLOW2_LOAD_AUX);
    };
    auto total = [&](const u32x4_f (&buf)[NTW]) {
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NTW; ++q)
            s += (f32x4){__uint_as_float(buf[q][0]), __uint_as_float(buf[q][1]), __uint_as_float(buf[q][2]), __uint_as_float(buf[q][3])};
        return s;
    };
    // Check the gathered tiles and add them up.  The sum exists TWICE, once per path: hipcc guards every later use of a register
    // that a retry loop MAY have re-loaded with the wait count of the re-load (vmcnt(0): nothing younger in flight there), so a
    // sum behind the merge of the two paths waited, on every step, for whatever the wave had issued since the gather -- the
    // write-back stores of the Q tiles in round 2's loop.  On the straight path the tag checks have already waited for exactly
    // the gathered tiles and nothing else.
    auto settle_total = [&](decltype(rp) rs, u32x4_f (&buf)[NTW], int slot, unsigned par) __attribute__((always_inline)) -> f32x4 {
        bool again = false;
#pragma unroll
        for (int q = 0; q < NTW; ++q) again = again || flow_untagged(buf[q], par);
        if (!__any(again) || dead) return total(buf);
        while (true) {
            if (wall_clock64() - t_begin > a.limit) { dead = true; if (lane == 0) __hip_atomic_fetch_or(FLOW_G(unsigned, a.err), 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            issue(rs, buf, slot);
            again = false;
#pragma unroll
            for (int q = 0; q < NTW; ++q) again = again || flow_untagged(buf[q], par);
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
        for (int n = 0; n < NTW; ++n)
            __builtin_amdgcn_raw_buffer_store_b128(flow_tag(acc[n], par), rs,
                                                   store_off + (unsigned)(n * NU * 1024) + (unsigned)slot * SLOT_BYTES, 0, FLOW2_STORE_AUX);
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
                dcin = dcout;
#endif
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
        const bool rec_on = S || (HD ? t > t_last : t > 0);       // (HD, t <= 0: the product of a stale tile, for the hand-off's sake)
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
                    for (int n = 0; n < NTW; ++n) {
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av2[g][0]), wd[n][g][0], acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av2[g][1]), wd[n][g][1], acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av2[g][2]), wd[n][g][2], acc[n], 0, 0, 0);
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(av2[g][3]), wd[n][g][3], acc[n], 0, 0, 0);
                    }
                }
                if (FLOW2_GATHER_AT >= 4 && rec_on) issue(rp, gp, t & 1);
            }
#pragma unroll
            for (int n = 0; n < NTW; ++n)
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
};
#ifndef BIG_XGATHER_AT
#define BIG_XGATHER_AT 1         // the partner tile's first load goes out after this many quarters (0..3) of the own-tile MFMAs
#endif
#ifndef BIG_SETTLE_ALL
#define BIG_SETTLE_ALL 1         // an explicit (free) vmcnt(0) behind the settle: see the step
#endif

template <int PR>             // PR: 0 exact f32, 1 bf16x3, 2 bf16 products
__global__ __launch_bounds__(512) void lstm_bwd_big(BigBwdArgs a) {
    constexpr bool BF3 = PR != 0;
    constexpr int H = 1024, NKB = 4 * H / 16, NRB = 2 * H / 16, NTW = 4, NW = 8, NP = 32;      // NP: workgroups (= tiles) per XCD
    __shared__ __attribute__((aligned(16))) float a_lds[2][1024];            // [own | partner][4 m][4 kq][16 i][4 g]: dG tiles as MFMA A fragments
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
    const unsigned long long t_begin = wall_clock64();

    // W_hh^T fragments: output tile nt = x*32 + wave*4 + n, K = the gate columns of unit block ub (p = 0) / pub (p = 1), gate g
    f32x4 wt[NTW][2][4];
    {
        const float* base = a.wq + (size_t)l * NRB * NKB * 256 + lane * 4;
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    wt[n][p][g] = *reinterpret_cast<const f32x4*>(base + ((size_t)(H / 16 + x * NP + wave * NTW + n) * NKB + g * (H / 16) + (p ? pub : ub)) * 256);
    }
    u32x4_f wth[BF3 ? NTW : 1][2][2], wtl[BF3 ? NTW : 1][2][2];      // split precision: [tile][own | partner][gate pair]
    if (BF3) {
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int p = 0; p < 2; ++p)
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
    const unsigned gather_off = pbase + (unsigned)(((j * NP + wave * NTW) * 256 + lane * 4) * 4);        // + q KiB: producer wave*4 + q
    const unsigned store_off = pbase + (unsigned)((((wave * NTW) * NP + j) * 256 + lane * 4) * 4);       // + n*NP KiB: consumer wave*4 + n
    // X ring: [slot][mb][unit block][1024 floats]
    const unsigned xslot_stride = (unsigned)nmt * 64u * 4096u;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(a.xring, 0, 2u * xslot_stride, 0x00020000);
    const unsigned x_store_off = (unsigned)((mb * 64 + ub) * 4096 + a_slot * 4);                          // this thread's four gates (epilogue threads)
    const unsigned x_load_off = (unsigned)((mb * 64 + pub) * 4096 + (wave * 64 + lane) * 8);              // this lane's 8 bytes of the partner tile
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
    u32x2_f gx;                    // this lane's 8 bytes of the partner's dG tile
    auto issue_x = [&](int slot) {
        gx = __builtin_amdgcn_raw_buffer_load_b64(rx, x_load_off, (unsigned)slot * xslot_stride, 16);      // sc1: written by the other XCD
    };
    auto settle_x = [&](int slot, unsigned par) {
        bool again = (((gx[0] ^ par) | (gx[1] ^ par)) & 1u) != 0u;
        if (__any(again) && !dead) {
            while (true) {
                if (wall_clock64() - t_begin > a.limit) { dead = true; if (lane == 0) atomicOr(a.err, 2u); break; }
                issue_x(slot);
                again = (((gx[0] ^ par) | (gx[1] ^ par)) & 1u) != 0u;
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
            if (t > 0) __builtin_amdgcn_raw_buffer_store_b128(flow_tag(dgv, par), rx, x_store_off + (unsigned)(t & 1) * xslot_stride, 0, 16);
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
            // ---- own tile (the partner's is on its way)
            mma_half(acc, av, 0, [&]() __attribute__((always_inline)) {      // (part-way through: see BIG_XGATHER_AT)
                __builtin_amdgcn_sched_barrier(0);
                issue_x(t & 1);
                __builtin_amdgcn_sched_barrier(0);
            });
            __builtin_amdgcn_sched_barrier(0);
            // ---- the partner's tile: 8 bytes per lane -> LDS -> everybody's A fragments
            settle_x(t & 1, par);
            *reinterpret_cast<u32x2_f*>(&a_lds[1][(wave * 64 + lane) * 2]) = gx;
            lds_barrier();
#pragma unroll
            for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<const f32x4*>(&a_lds[1][(m * 64 + lane) * 4]);
            mma_half(acc, av, 1, []() {});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NTW; ++n)      // (slot offset in voffset, not soffset: see store_tiles in lstm_bwd_flow2)
                __builtin_amdgcn_raw_buffer_store_b128(flow_tag(acc[n], par), rp,
                                                       store_off + (unsigned)(n * NP * 1024) + (unsigned)(t & 1) * pslot_stride, 0, 0);
            issue(t & 1);            // the next step's operand: most of it is there when the stash loads above have come back
        }
    }
}

// ====================================================================================
// Optional split-precision ("bf16x3") variants of the two step kernels (desc.precision = 1).
// Every f32 operand x is kept as two bf16 values, hi = bf16(x) and lo = bf16(x - hi) (16 significant
// bits), and every product a.b is evaluated as hi_a.hi_b + hi_a.lo_b + lo_a.hi_b on
// v_mfma_f32_16x16x32_bf16 with f32 accumulation: 3 MFMAs of 16 passes cover K = 32 where exact f32
// needs 8 MFMAs of 32 cycles -- the MFMA phase shrinks ~5x at the same operand bytes (2+2 per value).
// Measured on the oracle (3x512, T = 1001): logits within 7e-6 relative of float64 (exact f32: 5e-7).
// It is OFF by default: the headline path computes in exact f32 like the reference.
// Layouts: a K-block is 32 k; lane (j or row = lane%16, g = lane/16) holds k = 32*kb + 8*g + e, e = 0..7,
// as one 16-byte vector of bf16; each (tile, K-block) is 1 KiB of hi followed by 1 KiB of lo.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rne(float x) {
    const unsigned u = __float_as_uint(x);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ void bf16_split(float x, unsigned short& hi, unsigned short& lo) {
    hi = bf16_rne(x);
    lo = bf16_rne(x - __uint_as_float((unsigned)hi << 16));
}
// offset (in bf16 elements) of the HI half of element (row, k) in a packed panel with K columns; LO = +512
__device__ __forceinline__ size_t packed_off3(int row, int k, int K) {
    return ((size_t)(row >> 4) * (K >> 5) + (k >> 5)) * 1024 + (((k >> 3) & 3) * 16 + (row & 15)) * 8 + (k & 7);
}

__global__ void pack_rows_bf3_kernel(const float* __restrict__ src, size_t src_stride, unsigned short* __restrict__ dst,
                                     int B, int K, int nmat) {
    const size_t per = (size_t)B * K;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per * nmat) return;
    const int mat = i / per;
    const size_t r = i % per;
    const int row = r / K, k = r % K;
    const size_t bpk2 = (size_t)((B + 15) / 16 * 16) * K * 2;       // bf16 elements per matrix (hi + lo)
    unsigned short hi, lo;
    bf16_split(src[(size_t)mat * src_stride + r], hi, lo);
    unsigned short* d = dst + (size_t)mat * bpk2 + packed_off3(row, k, K);
    d[0] = hi; d[512] = lo;
}

// forward weights: [(l, ub)][kb32][nt] -> 1 KiB hi + 1 KiB lo; local column c = g*UW + u (UW = 8)
__global__ void pack_fwd_bf3_kernel(const float* __restrict__ kernels, long kstride, unsigned short* __restrict__ wp,
                                    int H, int L) {
    constexpr int UW = 8, NT = 2;
    const int NKB = 2 * H / 32, NUB = H / UW;
    const long total = (long)L * 2 * H * 4 * H;
    long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    const int e = o & 7, lane = (o >> 3) & 63;
    long r = o >> 9;
    const int nt = r % NT; r /= NT;
    const int kb = r % NKB; r /= NKB;
    const int ub = r % NUB; const int l = r / NUB;
    const int j = lane & 15, g8 = lane >> 4;
    const int c = nt * 16 + j, g = c / UW, u = c % UW;
    const int k = kb * 32 + g8 * 8 + e;
    unsigned short hi, lo;
    bf16_split(kernels[l * kstride + (long)k * 4 * H + g * H + ub * UW + u], hi, lo);
    unsigned short* d = wp + ((((size_t)l * NUB + ub) * NKB + kb) * NT + nt) * 1024 + lane * 8 + e;
    d[0] = hi; d[512] = lo;
}

// backward weights = K^T: [(l, rb)][kb32] with row = rb*16 + lane%16, column = 32*kb + 8*(lane/16) + e
__global__ void pack_bwd_bf3_kernel(const float* __restrict__ kernels, long kstride, unsigned short* __restrict__ wq,
                                    int H, int L) {
    const int NRB = 2 * H / 16, NKB = 4 * H / 32;
    const long total = (long)L * 2 * H * 4 * H;
    long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    const int e = o & 7, lane = (o >> 3) & 63;
    long r = o >> 9;
    const int kb = r % NKB; r /= NKB;
    const int rb = r % NRB; const int l = r / NRB;
    const int row = rb * 16 + (lane & 15), col = kb * 32 + (lane >> 4) * 8 + e;
    unsigned short hi, lo;
    bf16_split(kernels[l * kstride + (long)row * 4 * H + col], hi, lo);
    unsigned short* d = wq + (((size_t)l * NRB + rb) * NKB + kb) * 1024 + lane * 8 + e;
    d[0] = hi; d[512] = lo;
}

#define BF3_MMA(ACC, AH, AL, BH, BL)                                                               \
    do {                                                                                           \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AH), __builtin_bit_cast(bf16x8, BH), ACC, 0, 0, 0); \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AH), __builtin_bit_cast(bf16x8, BL), ACC, 0, 0, 0); \
        ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AL), __builtin_bit_cast(bf16x8, BH), ACC, 0, 0, 0); \
    } while (0)

// forward step, bf16x3: workgroup = 8 units x 4 gates (2 N tiles) x 32 rows (2 M tiles), K split over NW waves
template <int NW>
__global__ __launch_bounds__(NW * 64) void lstm_fwd_step_bf3(FwdArgs a) {
    constexpr int UW = 8, NT = 2, MT = 2, UN = 4;
    const int l = blockIdx.y;
    const int t = a.d - l;
    if (t < 0 || t >= a.T) return;
    const int ub = blockIdx.x;
    const int tile0 = a.mt0 + blockIdx.z * MT;
    const int T = a.T, B = a.B, H = a.H;
    const int nkb = 2 * H / 32, nkb_x = H / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* hp = a.hs + ((size_t)l * (T + 1) + t) * B * H;
    const int nmt = (B + 15) / 16;
    const size_t bph = (size_t)nmt * 16 * H;                 // panel size in floats == (hi+lo) bf16 pairs
    const int slot = a.d & 1;
    // panels as uint4: (tile, K-block) = 128 uint4 (64 hi + 64 lo)
    const uint4* xa = reinterpret_cast<const uint4*>(l == 0 ? a.xp0 + (size_t)t * bph : a.xp + ((size_t)l * 2 + slot) * bph) + lane;
    const uint4* ha = reinterpret_cast<const uint4*>(a.hp + ((size_t)l * 2 + slot) * bph) + lane;
    const uint4* wp = reinterpret_cast<const uint4*>(a.wp) + ((size_t)(l * (H / UW) + ub) * nkb) * (NT * 128) + lane;

    const float* bias = a.bias + l * a.bias_stride;
    const float* cprev = a.cs + ((size_t)l * (T + 1) + t) * B * H;
    const int pidx = threadIdx.x % (16 * MT * UW);
    const int pbl = pidx / UW, pu = pidx % UW;
    const int pb = tile0 * 16 + pbl, punit = ub * UW + pu;
    const bool pok = threadIdx.x < 16 * MT * UW && pb < B;
    const int pbc = min(pb, B - 1);
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
    size_t tileoff[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) tileoff[i] = (size_t)min(tile0 + i, nmt - 1) * (H / 32) * 128;
    const int kb0 = wave * nkb / NW, kb1 = (wave + 1) * nkb / NW;
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    for (int kbs = kb0; kbs < kb1; kbs += UN) {
        uint4 ah[UN][MT], al[UN][MT], bh[UN][NT], bl[UN][NT];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const bool kok = kbs + u < kb1;
            const int kb = min(kbs + u, kb1 - 1);
            const bool isx = kb < nkb_x;
            const uint4* src = (isx ? xa : ha) + (size_t)(isx ? kb : kb - nkb_x) * 128;
#pragma unroll
            for (int i = 0; i < MT; ++i) { ah[u][i] = src[tileoff[i]]; al[u][i] = src[tileoff[i] + 64]; }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const uint4 h = wp[(size_t)(kb * NT + j) * 128], lo = wp[(size_t)(kb * NT + j) * 128 + 64];
                bh[u][j] = kok ? h : zero; bl[u][j] = kok ? lo : zero;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) BF3_MMA(acc[i][j], ah[u][i], al[u][i], bh[u][j], bl[u][j]);
    }

    __shared__ __attribute__((aligned(16))) float red[NW][MT * NT][256];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(&red[wave][i * NT + j][lane * 4]) = acc[i][j];
    __syncthreads();
    if (!pok) return;
    const int mt = pbl >> 4, i = pbl & 15;
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = g * UW + pu, nt = c >> 4, j = c & 15;
        const int e = ((i >> 2) * 16 + j) * 4 + (i & 3);
        float sacc = e_bias[g];
#pragma unroll
        for (int w = 0; w < NW; ++w) sacc += red[w][mt * NT + nt][e];
        pre[g] = sacc;
    }
    const float gi = sigmoidf_(pre[0]);
    const float gj = tanhf(pre[1]);
    const float gf = sigmoidf_(pre[2] + 1.0f);
    const float go = sigmoidf_(pre[3]);
    const size_t e = (size_t)pb * H + punit;
    const float cn = e_cp * gf + gi * gj;
    const float hn = tanhf(cn) * go;
    const bool live = t < e_len;
    float* gr = a.gates + ((size_t)l * T + t) * B * 4 * H + (size_t)pb * 4 * H + punit;
    gr[0] = gi; gr[H] = gj; gr[2 * H] = gf; gr[3 * H] = go;
    const float hv = live ? hn : e_hp;
    const float zv = live ? hn * zmult(a.drop, l + 1, (uint32_t)((size_t)t * B * H + e)) : 0.0f;
    a.cs[((size_t)l * (T + 1) + t + 1) * B * H + e] = live ? cn : e_cp;
    a.hs[((size_t)l * (T + 1) + t + 1) * B * H + e] = hv;
    a.z[((size_t)(l + 1) * T + t) * B * H + e] = zv;
    const size_t po = packed_off3(pb, punit, H);
    unsigned short hi, lo;
    unsigned short* hp3 = reinterpret_cast<unsigned short*>(a.hp + ((size_t)l * 2 + (slot ^ 1)) * bph);
    bf16_split(hv, hi, lo); hp3[po] = hi; hp3[po + 512] = lo;
    if (l + 1 < a.L) {
        unsigned short* xp3 = reinterpret_cast<unsigned short*>(a.xp + ((size_t)(l + 1) * 2 + (slot ^ 1)) * bph);
        bf16_split(zv, hi, lo); xp3[po] = hi; xp3[po + 512] = lo;
    }
}

// backward step, bf16x3: workgroup = 16 units x 16 rows, two product streams (rec / up), K = 4H each
template <int NW>
__global__ __launch_bounds__(NW * 64) void lstm_bwd_step_bf3(BwdArgs a) {
    constexpr int UN = 4;                         // virtual K-blocks (32 k) per burst
    const int l = blockIdx.y;
    const int T = a.T, B = a.B, H = a.H, L = a.L;
    const int t = (T - 1) - (a.d - (L - 1 - l));
    if (t < 0 || t >= T) return;
    const int ub = blockIdx.x, mb = a.mt0 + blockIdx.z;
    const int nkb = 4 * H / 32, nrb = 2 * H / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nmt = (B + 15) / 16;
    const size_t bpg = (size_t)nmt * 16 * 4 * H;
    const int slot = a.d & 1;
    const bool has_rec = t + 1 < T, has_up = l + 1 < L;

    const int bl = (threadIdx.x & 255) >> 4, u = threadIdx.x & 15;
    const int b = mb * 16 + bl;
    const int unit = ub * 16 + u;
    const bool pok = threadIdx.x < 256 && b < B;
    const int bc = min(b, B - 1);
    const size_t bec = (size_t)bc * H + unit;
    const size_t be = (size_t)b * H + unit;
    float* dcb = a.dc + (size_t)l * 2 * B * H;
    const float* gr = a.gates + ((size_t)l * T + t) * B * 4 * H + (size_t)bc * 4 * H + unit;
    const float gi = gr[0], gj = gr[H], gf = gr[2 * H], go = gr[3 * H];
    const float c = a.cs[((size_t)l * (T + 1) + t + 1) * B * H + bec];
    const float cp = a.cs[((size_t)l * (T + 1) + t) * B * H + bec];
    const float dcin_raw = dcb[(size_t)((t + 1) & 1) * B * H + bec];
    const float dtop = a.dztop[(size_t)t * B * H + bec];
    const int len = a.lengths[bc];
    const float dcin = has_rec ? dcin_raw : 0.0f;

    const uint4* a0p = reinterpret_cast<const uint4*>(a.dgp + ((size_t)l * 2 + slot) * bpg) + (size_t)mb * nkb * 128 + lane;
    const uint4* a1p = reinterpret_cast<const uint4*>(a.dgp + ((size_t)(l + 1) * 2 + slot) * bpg) + (size_t)mb * nkb * 128 + lane;
    const uint4* b0p = reinterpret_cast<const uint4*>(a.wq) + ((size_t)(l * nrb + H / 16 + ub) * nkb) * 128 + lane;
    const uint4* b1p = reinterpret_cast<const uint4*>(a.wq) + ((size_t)((l + 1) * nrb + ub) * nkb) * 128 + lane;
    const int nsrc = (has_rec ? 1 : 0) + (has_up ? 1 : 0);
    const int kb0 = wave * nkb / NW, kb1 = (wave + 1) * nkb / NW;
    const int nv = (kb1 - kb0) * nsrc;
    const int only = has_rec ? 0 : 1;
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int vs = 0; vs < nv; vs += UN) {
        uint4 ah[UN], al[UN], bh[UN], blo[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const bool ok = vs + q < nv;
            const int v = min(vs + q, nv - 1);
            const int sidx = nsrc == 2 ? (v & 1) : only;
            const int kb = kb0 + (nsrc == 2 ? (v >> 1) : v);
            const uint4* ap = (sidx ? a1p : a0p) + (size_t)kb * 128;
            const uint4* bp = (sidx ? b1p : b0p) + (size_t)kb * 128;
            ah[q] = ap[0]; al[q] = ap[64];
            const uint4 h = bp[0], lo = bp[64];
            bh[q] = ok ? h : zero; blo[q] = ok ? lo : zero;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < UN; ++q) BF3_MMA(acc[q & 1], ah[q], al[q], bh[q], blo[q]);
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
    float dgv[4];
    dgv[0] = dct * gj * gi * (1.0f - gi);
    dgv[1] = dct * gi * (1.0f - gj * gj);
    dgv[2] = dct * cp * gf * (1.0f - gf);
    dgv[3] = dh * tc * go * (1.0f - go);
    float dcout = dct * gf;
    if (!live) { dgv[0] = dgv[1] = dgv[2] = dgv[3] = 0.0f; dcout = 0.0f; }
    float* dgw = a.dg + ((size_t)l * T + t) * B * 4 * H + (size_t)b * 4 * H + unit;
    unsigned short* dgp3 = reinterpret_cast<unsigned short*>(a.dgp + ((size_t)l * 2 + (slot ^ 1)) * bpg);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        dgw[g * H] = dgv[g];
        unsigned short hi, lo;
        bf16_split(dgv[g], hi, lo);
        const size_t po = packed_off3(b, g * H + unit, 4 * H);
        dgp3[po] = hi; dgp3[po + 512] = lo;
    }
    dcb[(size_t)(t & 1) * B * H + be] = dcout;
}

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
    return env != 0 && d->precision >= 0 && d->precision <= 2 && d->H == 1024 && (d->B + 15) / 16 <= 4 && device_cus() == 256 &&
           (size_t)2 * ((d->B + 15) / 16 * 16) * d->H * 4 < (1ull << 32);
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

int lstm_fwd(hipStream_t s, const amdspeech_lstm_desc* d, float* ws, const float* kernels, long kstride,
             const float* biases, long bstride, const int* lengths, const float* h0, const float* c0,
             const amdspeech_ctc_head* head = nullptr) {
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
            // AMDSPEECH_FLOW_FILL_QUEUE=1: the fills as ONE work-queue launch that really runs beside the forward kernel.  Measured
            // (round 5, headline shape, alternating runs on one box): the forward kernel 4.46 - 4.51 -> 4.70 - 5.20 ms, the step 12.40 -
            // 12.43 -> 12.59 - 12.73 ms -- 550 MB of stores through the fabric the x-product workers and the CTC follower read their
            // operands through cost the recurrence more than the 0.1 ms the memset launches spend between the two recurrence kernels.  Off.
            static const int fillq = runtime_switch("AMDSPEECH_FLOW_FILL_QUEUE", 0);
            if (fillq) {
                // ONE work-queue launch (flow_fill_queue_kernel): what flow_fill_bwd_panels, pack_bwd_kernel and flow_fill_fwd_panels do
                const size_t bpg = bp * 4 * H;
                float* other = ws + (size_t)(1 - set) * lo.fwd_set;
                FillJobs fj{};
                int n = 0;
                auto job = [&](float* p, size_t dwords, unsigned v) { if (dwords > 0) { fj.p[n] = p; fj.n[n] = dwords; fj.v[n] = v; ++n; } };
                job(ws + lo.prec, lo.total - lo.prec, 0u);
                job(ws + lo.dxh, L > 1 ? (size_t)(L - 1) * T * (bpg / 4) : 0, FLOW_SENTINEL);
                job(ws + lo.dztop, head != nullptr ? (size_t)T * B * H : 0, FLOW_SENTINEL);
                job(other + lo.xph + (size_t)T * bph, (size_t)L * T * bph, FLOW_SENTINEL);
                job(other + lo.hph, (size_t)L * (T + 1) * bph, FLOW_SENTINEL);
                fj.count = n;
                fj.kernels = kernels; fj.kstride = kstride; fj.wq = ws + lo.wq; fj.H = H; fj.L = L;
                hipLaunchKernelGGL(flow_fill_queue_kernel, dim3(512), dim3(256), 0, g_side, fj, err + 32);
                AS_CHECK_LAUNCH();
            } else {
            if (int rc = flow_fill_bwd_panels(g_side, d, ws, lo, head != nullptr)) return rc;
            hipLaunchKernelGGL(pack_bwd_kernel, dim3(ceil_div(wtotal, 256)), dim3(256), 0, g_side, kernels, kstride, ws + lo.wq, H, L);
            AS_CHECK_LAUNCH();      // (the backward call's K^T pack: the weights do not change between the two halves of a cycle)
            if (int rc = flow_fill_fwd_panels(g_side, d, ws, lo, 1 - set)) return rc;
            }
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
        const size_t TB = (size_t)T * B, bp = (size_t)nmt * 16;
        unsigned* err = reinterpret_cast<unsigned*>(ws + lo.sync);
        BigFwdArgs ba;
        ba.wp = ws + lo.wp; ba.z = ws + lo.z; ba.hs = ws + lo.hs; ba.cs = ws + lo.cs; ba.gates = ws + lo.gates; ba.lengths = lengths;
        ba.err = err; ba.tickets = err + 16;
        ba.T = T; ba.B = B; ba.H = H; ba.L = L; ba.drop = dc;
        ba.limit = 100000000ull + (unsigned long long)T * 10000ull;
        for (int l = 0; l < L; ++l) {
            // pre-activations of ALL frames: [T*B, H] . K_l[0:H, :] + b_l -> gates[l] (replaced frame by frame by the kernel)
            if (int rc = bf16p_layout_on(d) ? bf16p_xw(s, bf16p_bufs(d, ws + lo.bfs), (int)TB, H, ws + lo.z + (size_t)l * TB * H, kernels + l * kstride,
                                                       ws + lo.gates + (size_t)l * TB * 4 * H, biases + l * bstride)
                       : bf3_gemm(d) ? gemm_reduced(d, s, false, false, (int)TB, 4 * H, H, ws + lo.z + (size_t)l * TB * H, H, kernels + l * kstride,
                                                    4 * H, ws + lo.gates + (size_t)l * TB * 4 * H, 4 * H, biases + l * bstride, false)
                                     : gemm_f32_plain(s, false, false, (int)TB, 4 * H, H, ws + lo.z + (size_t)l * TB * H, H, kernels + l * kstride,
                                                      4 * H, ws + lo.gates + (size_t)l * TB * 4 * H, 4 * H, biases + l * bstride, false)) return rc;
            // the h ring of this layer: slot 0 = the packed initial state with every word tagged 1, slot 1 = zeros (tag 0)
            float* ring = ws + lo.hp + (size_t)l * 2 * bp * H;
            AS_CHECK_HIP(hipMemsetAsync(ring, 0, 2 * bp * H * sizeof(float), s));
            hipLaunchKernelGGL(pack_rows_kernel, dim3(ceil_div(bh, 256)), dim3(256), 0, s,
                               ws + lo.hs + (size_t)l * (T + 1) * bh, bh, ring, B, H, 1);
            hipLaunchKernelGGL(tag_panel_kernel, dim3(ceil_div(bp * H, 256)), dim3(256), 0, s, ring, bp * H, 1u);
            AS_CHECK_HIP(hipMemsetAsync(ba.tickets, 0, 8 * sizeof(unsigned), s));
            ba.hring = ring; ba.layer = l;
            prof_begin(0, s, l);
            if (d->precision == 2) hipLaunchKernelGGL(lstm_fwd_big<2>, dim3(256), dim3(512), 0, s, ba);
            else if (d->precision == 1) hipLaunchKernelGGL(lstm_fwd_big<1>, dim3(256), dim3(512), 0, s, ba);
            else hipLaunchKernelGGL(lstm_fwd_big<0>, dim3(256), dim3(512), 0, s, ba);      // one workgroup per CU; each finds its place by XCC_ID
            prof_end(0, s, T * L, l);
        }
        AS_CHECK_LAUNCH();
        return AMDSPEECH_OK;
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

int lstm_bwd(hipStream_t s, const amdspeech_lstm_desc* d, float* ws, const float* kernels, long kstride,
             float* dkernels, float* dbiases, long bstride, const int* lengths, const amdspeech_ctc_head* head = nullptr) {
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
        size_t lds = ((size_t)2 * 1024 + 2 * 8 * 256 + (FLOW2_WINDOW ? 2 : 1) * 8 * (H / 128) * 256) * sizeof(float);
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
            pieces = 4; percent = 38;
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
        const int share = dev_knob_str("AMDSPEECH_FLOW_GEMM") ? percent : (d->precision != 0 ? percent * 3 / 4 : (head != nullptr ? percent - 3 : percent));
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
        b2.limit = 100000000ull + (unsigned long long)T * 10000ull;
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
        return AMDSPEECH_EHIP;
    }
    return AMDSPEECH_OK;
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
