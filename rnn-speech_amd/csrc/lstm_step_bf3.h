// Part of lstm.hip -- the launch-per-diagonal kernels in split precision (bf16x3) and their packs, DESIGN.md 4.2d.
// Not a standalone header: lstm.hip includes its kernel families in a fixed order, inside namespace amdspeech, after the helpers
// (layout, dropout multipliers, packs) they use.  Tuning macros (#ifndef ...) keep their defaults here; rnn-speech_amd/build.py
// passes overrides for development builds (AMDSPEECH_CXXFLAGS).

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

