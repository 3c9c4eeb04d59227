// Audio front end for gfx950: batch MFCC (librosa semantics) and log-mel filterbank
// + delta + delta-delta (the reference's numpy body), replacing
// /root/reference/util/audioprocessor.py:63-75 and :77-161.
//
// Design: HBM traffic is ~hop*4 B of PCM in and D*4 B out per frame -- negligible; the
// work is a 400/512-point real DFT per frame (n_fft = round(0.025*sr) is not a power of
// two, so it is a direct DFT against an LDS-resident twiddle table, 8 frames per
// workgroup so that one twiddle gather feeds 16 FMAs), the mel projection and a log.
// One kernel does PCM -> (pre-emphasis) -> frame -> window -> |DFT|^2 -> filterbank ->
// 10*log10 for 8 frames; the utterance-global statistics (top_db clamp against the
// utterance maximum for mfcc, per-filter mean for fbank) force a second pass, which
// also applies the DCT (mfcc) or the 9-tap Savitzky-Golay deltas (fbank) and writes
// the time-major [T, B, D] batch the LSTM consumes.
#include "common.h"
#include <math.h>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace amdspeech {

constexpr int FPB = 8;        // frames per workgroup
constexpr int MODE_MFCC = 0, MODE_FBANK = 1;
constexpr int N_MELS = 128, N_FILT = 40;

static inline int round_half_even(double x) {           // Python round()
    double r = nearbyint(x);                              // default FE_TONEAREST = ties to even
    return (int)r;
}

struct FrontCfg {
    int mode, sr, hop, win, n_dft, frame_len, n_bins, n_filt, center;
    float power_scale;
};

static FrontCfg make_cfg(int mode, int sr) {
    FrontCfg c;
    c.mode = mode; c.sr = sr;
    c.hop = round_half_even(sr * 0.01);
    c.win = round_half_even(sr * 0.025);
    if (mode == MODE_MFCC) {
        c.n_dft = c.win; c.frame_len = c.win; c.n_filt = N_MELS; c.center = 1; c.power_scale = 1.0f;
    } else {
        c.n_dft = 512; c.frame_len = c.win < 512 ? c.win : 512; c.n_filt = N_FILT; c.center = 0;
        c.power_scale = 1.0f / 512.0f;
    }
    c.n_bins = c.n_dft / 2 + 1;
    return c;
}

static int num_frames(const FrontCfg& c, int n) {
    if (n <= 0) return 0;
    if (c.mode == MODE_MFCC) return 1 + (n + 2 * (c.n_dft / 2) - c.n_dft) / c.hop;
    const int diff = n > c.win ? n - c.win : c.win - n;
    return (diff + c.hop - 1) / c.hop;                     // ceil(|N - win| / hop)
}

// ---- host-side tables (double precision, cached per (mode, sr, n_mfcc)) ---------
struct Tables {
    std::vector<float> twiddle;   // [n_dft][2] cos, sin
    std::vector<float> window;    // [frame_len]
    std::vector<float> filt;      // [n_bins][n_filt]  (transposed for coalesced reads)
    std::vector<float> dct;       // [n_mfcc][N_MELS]  (mfcc only)
    // the MFMA kernel's operands (frontend_frames_mfma_kernel): the folded DFT as two matrices and the filterbank transposed,
    // k contiguous, every extent padded to a multiple of 16 with zeros
    std::vector<float> cos_t;     // [nbp][kp]   cos(2 pi (k n mod N) / N), n = 0 .. N/2
    std::vector<float> sin_t;     // [nbp][kp]   sin(...)
    std::vector<float> filt_t;    // [mp][nbp]   filt^T
    int kp = 0, nbp = 0, mp = 0, half = 0;
    float* dev = nullptr;         // the tables back to back in device memory (uploaded once per configuration and device)
    size_t o_window = 0, o_filt = 0, o_dct = 0, o_cos = 0, o_sin = 0, o_filt_t = 0;      // float offsets inside `dev` (each 64-float aligned)
};

static double hz_to_mel_slaney(double f) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz_slaney(double m) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}

static void build_tables(const FrontCfg& c, int n_mfcc, Tables& t) {
    const double PI = 3.14159265358979323846;
    t.twiddle.resize((size_t)c.n_dft * 2);
    for (int i = 0; i < c.n_dft; ++i) {
        t.twiddle[2 * i] = (float)cos(2.0 * PI * i / c.n_dft);
        t.twiddle[2 * i + 1] = (float)sin(2.0 * PI * i / c.n_dft);
    }
    t.window.resize(c.frame_len);
    for (int i = 0; i < c.frame_len; ++i) {
        if (c.mode == MODE_MFCC) t.window[i] = (float)(0.5 - 0.5 * cos(2.0 * PI * i / c.win));          // periodic Hann
        else t.window[i] = (float)(0.54 - 0.46 * cos(2.0 * PI * i / (c.win - 1)));                        // np.hamming
    }
    // Folded real DFT: with e[n] = x[n] + x[N-n], o[n] = x[n] - x[N-n] (n = 1 .. ceil(N/2)-1; e[0] = x[0], e[N/2] = x[N/2]),
    // Re X[k] = sum_{n <= N/2} e[n] cos(2 pi k n / N), Im X[k] = -sum o[n] sin(2 pi k n / N): half the products of the direct sum.
    // The table entries are the SAME floats the direct kernel multiplies by (argument reduced to k n mod N first).
    t.half = c.n_dft / 2;
    t.kp = (t.half + 1 + 31) / 32 * 32;               // (two 16-sample MFMA groups per loop trip of the kernel)
    t.nbp = (c.n_bins + 15) / 16 * 16;
    t.mp = (c.n_filt + 15) / 16 * 16;
    t.cos_t.assign((size_t)t.nbp * t.kp, 0.0f);
    t.sin_t.assign((size_t)t.nbp * t.kp, 0.0f);
    for (int k = 0; k < c.n_bins; ++k)
        for (int n = 0; n <= t.half; ++n) {
            const int idx = (int)(((long)k * n) % c.n_dft);
            t.cos_t[(size_t)k * t.kp + n] = t.twiddle[2 * idx];
            t.sin_t[(size_t)k * t.kp + n] = t.twiddle[2 * idx + 1];
        }
    t.filt.assign((size_t)c.n_bins * c.n_filt, 0.0f);
    if (c.mode == MODE_MFCC) {
        // librosa.filters.mel(sr, n_fft, 128, fmin=0, fmax=sr/2, htk=False, norm='slaney')
        std::vector<double> mel_f(N_MELS + 2);
        const double m_lo = hz_to_mel_slaney(0.0), m_hi = hz_to_mel_slaney(c.sr / 2.0);
        for (int i = 0; i < N_MELS + 2; ++i) mel_f[i] = mel_to_hz_slaney(m_lo + (m_hi - m_lo) * i / (N_MELS + 1));
        for (int m = 0; m < N_MELS; ++m) {
            const double enorm = 2.0 / (mel_f[m + 2] - mel_f[m]);
            for (int k = 0; k < c.n_bins; ++k) {
                const double f = (c.sr / 2.0) * k / (c.n_bins - 1);
                const double lower = (f - mel_f[m]) / (mel_f[m + 1] - mel_f[m]);
                const double upper = (mel_f[m + 2] - f) / (mel_f[m + 2] - mel_f[m + 1]);
                double w = lower < upper ? lower : upper;
                if (w < 0) w = 0;
                t.filt[(size_t)k * c.n_filt + m] = (float)(w * enorm);
            }
        }
        t.dct.resize((size_t)n_mfcc * N_MELS);
        for (int q = 0; q < n_mfcc; ++q)
            for (int m = 0; m < N_MELS; ++m) {
                double v = cos(PI * q * (2 * m + 1) / (2.0 * N_MELS)) * sqrt(2.0 / N_MELS);
                if (q == 0) v *= sqrt(0.5);
                t.dct[(size_t)q * N_MELS + m] = (float)v;
            }
    } else {
        // util/audioprocessor.py:107-133: 40 triangles on the HTK mel scale, edges floor((nfft+1)*hz/sr)
        const int nfft = 512;
        const double high_mel = 2595.0 * log10(1.0 + (c.sr / 2.0) / 700.0);
        std::vector<double> edge(N_FILT + 2);
        for (int i = 0; i < N_FILT + 2; ++i) {
            const double mel = high_mel * i / (N_FILT + 1);
            const double hz = 700.0 * (pow(10.0, mel / 2595.0) - 1.0);
            edge[i] = floor((nfft + 1) * hz / c.sr);
        }
        for (int m = 1; m <= N_FILT; ++m) {
            const int lo = (int)edge[m - 1], ce = (int)edge[m], hi = (int)edge[m + 1];
            for (int k = lo; k < ce && k < c.n_bins; ++k)
                t.filt[(size_t)k * N_FILT + (m - 1)] = (float)((k - edge[m - 1]) / (edge[m] - edge[m - 1]));
            for (int k = ce; k < hi && k < c.n_bins; ++k)
                t.filt[(size_t)k * N_FILT + (m - 1)] = (float)((edge[m + 1] - k) / (edge[m + 1] - edge[m]));
        }
    }
    t.filt_t.assign((size_t)t.mp * t.nbp, 0.0f);
    for (int k = 0; k < c.n_bins; ++k)
        for (int m = 0; m < c.n_filt; ++m) t.filt_t[(size_t)m * t.nbp + k] = t.filt[(size_t)k * c.n_filt + m];
}

// The tables are constants of (mode, sample rate, n_mfcc): built once on the host and kept in device memory owned by the
// library (per device) -- the hot path issues no table copies.  Returns nullptr if the upload fails.
static const Tables* get_tables(const FrontCfg& c, int n_mfcc) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int>, Tables*> cache;
    std::lock_guard<std::mutex> lock(mu);
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    auto key = std::make_tuple(c.mode, c.sr, n_mfcc, device);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    Tables* t = new Tables();
    build_tables(c, n_mfcc, *t);
    auto up64 = [](size_t n) { return (n + 63) / 64 * 64; };
    t->o_window = up64(t->twiddle.size());
    t->o_filt = t->o_window + up64(t->window.size());
    t->o_dct = t->o_filt + up64(t->filt.size());
    t->o_cos = t->o_dct + up64(t->dct.size() ? t->dct.size() : 1);
    t->o_sin = t->o_cos + up64(t->cos_t.size());
    t->o_filt_t = t->o_sin + up64(t->sin_t.size());
    const size_t total = t->o_filt_t + up64(t->filt_t.size());
    if (hipMalloc(reinterpret_cast<void**>(&t->dev), total * 4) != hipSuccess) { delete t; return nullptr; }
    bool ok = hipMemcpy(t->dev, t->twiddle.data(), t->twiddle.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(t->dev + t->o_window, t->window.data(), t->window.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(t->dev + t->o_filt, t->filt.data(), t->filt.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!t->dct.empty())
        ok = ok && hipMemcpy(t->dev + t->o_dct, t->dct.data(), t->dct.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(t->dev + t->o_cos, t->cos_t.data(), t->cos_t.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(t->dev + t->o_sin, t->sin_t.data(), t->sin_t.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(t->dev + t->o_filt_t, t->filt_t.data(), t->filt_t.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { (void)hipFree(t->dev); delete t; return nullptr; }
    cache[key] = t;
    return t;
}

// ---- workspace ---------------------------------------------------------------
struct FrontLayout { size_t twiddle, window, filt, dct, logmel, d1, stat, ctl, total; int t_full; };

static FrontLayout front_layout(const FrontCfg& c, int B, int n_max, int n_mfcc_max) {
    FrontLayout o;
    o.t_full = num_frames(c, n_max);
    if (o.t_full < 1) o.t_full = 1;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t r = off; off += align_up(bytes, 256); return r; };
    o.twiddle = take((size_t)c.n_dft * 8);
    o.window = take((size_t)c.frame_len * 4);
    o.filt = take((size_t)c.n_bins * c.n_filt * 4);
    o.dct = take((size_t)n_mfcc_max * N_MELS * 4);
    o.logmel = take((size_t)B * o.t_full * c.n_filt * 4);
    o.d1 = take(c.mode == MODE_FBANK ? (size_t)B * o.t_full * c.n_filt * 4 : 4);
    o.stat = take((size_t)B * c.n_filt * 8);          // per-filter mean (fbank)
    o.ctl = take(64 + (size_t)B * 4);                 // zeroed per call: [0] the frame kernel's work-queue head, [16 ..] per-utterance max keys (mfcc)
    o.total = off;
    return o;
}

// ---- kernel 1: PCM -> log filterbank energies, FPB frames per workgroup ----------
// Order-preserving map float -> unsigned (for atomicMax on floats of either sign); 0 is below every float's key.
__device__ __forceinline__ unsigned float_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

struct FrameArgs {
    const float* pcm; const int* nsamp; const int* nframes;   // device
    const float* twiddle; const float* window; const float* filt;
    float* logmel;
    unsigned* umax;               // mfcc: per-utterance max of the log-mel energies as float_key (zeroed before the launch)
    int n_max, t_full, hop, n_dft, frame_len, n_bins, n_filt, center, preemph, mode;
    float power_scale;
    const float* cos_t; const float* sin_t; const float* filt_t;      // frontend_frames_mfma_kernel's operands (Tables)
    int half, kp, nbp, mp;
    unsigned* queue;              // frontend_frames_mfma_kernel: next work item (zeroed before the launch)
    int tiles_per_utt, n_items;   // items = (utterance, tile of FR frames)
};

__global__ __launch_bounds__(256) void frontend_frames_kernel(FrameArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;                                   // [frame_len][FPB]
    float* tw = xs + a.frame_len * FPB;               // [n_dft][2]
    float* pw = tw + a.n_dft * 2;                     // [n_bins][FPB]
    const int b = blockIdx.y, f0 = blockIdx.x * FPB;
    const int nf = a.nframes[b];
    if (f0 >= nf) return;
    const int N = a.nsamp[b];
    const float* x = a.pcm + (size_t)b * a.n_max;
    for (int i = threadIdx.x; i < a.n_dft * 2; i += 256) tw[i] = a.twiddle[i];
    for (int i = threadIdx.x; i < a.frame_len * FPB; i += 256) {
        const int f = i / a.frame_len, n = i % a.frame_len;   // consecutive threads -> consecutive samples
        int j = (f0 + f) * a.hop + n;
        float v = 0.0f;
        if (a.center) {
            j -= a.n_dft / 2;
            if (j < 0) j = -j;
            if (j >= N) j = 2 * (N - 1) - j;
            v = (j >= 0 && j < N) ? x[j] : 0.0f;
        } else if (j < N) {
            v = x[j];
            if (a.preemph && j > 0) v = v - 0.97f * x[j - 1];
        }
        xs[n * FPB + f] = v * a.window[n];
    }
    __syncthreads();
    // direct real DFT: thread = bin k, FPB frames at once
    for (int k = threadIdx.x; k < a.n_bins; k += 256) {
        float re[FPB], im[FPB];
#pragma unroll
        for (int f = 0; f < FPB; ++f) { re[f] = 0.f; im[f] = 0.f; }
        int idx = 0;
        for (int n = 0; n < a.frame_len; ++n) {
            const float2 w = *reinterpret_cast<const float2*>(tw + 2 * idx);
            const float4 x0 = *reinterpret_cast<const float4*>(xs + n * FPB);
            const float4 x1 = *reinterpret_cast<const float4*>(xs + n * FPB + 4);
            re[0] += x0.x * w.x; im[0] += x0.x * w.y;
            re[1] += x0.y * w.x; im[1] += x0.y * w.y;
            re[2] += x0.z * w.x; im[2] += x0.z * w.y;
            re[3] += x0.w * w.x; im[3] += x0.w * w.y;
            re[4] += x1.x * w.x; im[4] += x1.x * w.y;
            re[5] += x1.y * w.x; im[5] += x1.y * w.y;
            re[6] += x1.z * w.x; im[6] += x1.z * w.y;
            re[7] += x1.w * w.x; im[7] += x1.w * w.y;
            idx += k;
            if (idx >= a.n_dft) idx -= a.n_dft;
        }
#pragma unroll
        for (int f = 0; f < FPB; ++f) pw[k * FPB + f] = (re[f] * re[f] + im[f] * im[f]) * a.power_scale;
    }
    __syncthreads();
    // filterbank + log: thread = (frame, filter)
    float vmax = -__builtin_inff();
    for (int i = threadIdx.x; i < FPB * a.n_filt; i += 256) {
        const int f = i / a.n_filt, m = i % a.n_filt;
        if (f0 + f >= nf) continue;
        float acc = 0.f;
        for (int k = 0; k < a.n_bins; ++k) acc += pw[k * FPB + f] * a.filt[(size_t)k * a.n_filt + m];
        float v;
        if (a.mode == MODE_MFCC) v = 10.0f * log10f(fmaxf(acc, 1e-10f));
        else v = 10.0f * log10f(acc == 0.0f ? 2.220446049250313e-16f : acc);
        a.logmel[((size_t)b * a.t_full + f0 + f) * a.n_filt + m] = v;
        vmax = fmaxf(vmax, v);
    }
    if (a.mode == MODE_MFCC) {      // the utterance's maximum (power_to_db's top_db reference): wave max, one atomic per wave
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if ((threadIdx.x & 63) == 0 && vmax > -__builtin_inff()) atomicMax(a.umax + b, float_key(vmax));
    }
}

// ---- kernel 1 on the matrix cores ------------------------------------------------------------------------------------------
// The direct DFT of a frame IS a matrix product (frames x samples) . (samples x bins), and so is the filterbank; kernel 1 above
// does both on the vector ALUs at ~29 TFLOP/s (it is LDS-bandwidth bound: every thread re-reads every sample).  Here a workgroup
// takes FR = 32 frames of one utterance:
//   1. PCM -> (pre-emphasis, centring) -> window -> FOLDED frames e[f][n], o[f][n], n = 0 .. N/2, in LDS (see build_tables: the
//      fold halves the products);
//   2. Re = e . cos_t^T and Im = o . sin_t^T on v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulation): a wave owns every
//      fourth 16-bin tile for both 16-frame tiles; the frame fragments come from LDS (one ds_read_b128 = the operand of four MFMAs),
//      the twiddle fragments straight from L2 (the two matrices are a few hundred KiB), one 16-sample step ahead in ping-pong
//      registers; |X|^2 stays in registers until every wave is done with e / o,
//   3. then goes to LDS over them as P[f][bin], and the filterbank P . filt_t^T runs the same way (a wave owns every fourth
//      16-filter tile); log, store, per-utterance maximum as in kernel 1.
// Same numbers as kernel 1 up to the order of the f32 sums (the twiddle / window / filter floats are the same).
constexpr int FR = 32;
// FR_MAXQ (template): bin tiles per wave -- 4 for the 16 kHz mfcc front end (13 tiles over 4 waves), 5 for fbank (17), 9 for
// anything up to a 1024-point DFT; each costs ~25 registers, and 4 / 5 keep the kernel at two waves per SIMD

// one (utterance b, frames f0 .. f0 + 31) item
template <int FR_MAXQ>
__device__ __forceinline__ void frontend_frames_item(const FrameArgs& a, float* sm, int b, int f0, int nf) {
    const int ldk = a.kp + 4, ldp = a.nbp + 4;
    float* ev = sm;                                   // [FR][ldk]
    float* od = sm + FR * ldk;                        // [FR][ldk]
    float* raw = sm + 2 * FR * ldk;                   // the PCM span of the 32 frames: (FR - 1) hop + frame_len samples
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, i = lane & 15, kq = lane >> 4;
    const int N = a.nsamp[b];
    const float* x = a.pcm + (size_t)b * a.n_max;
    // 1a. the span, once: sample j of the utterance after reflect padding (mfcc) / pre-emphasis (fbank).  Branch-free (clamped
    //     index, value selected afterwards) and eight loads per thread in flight: written as "if in range, load" a thread's
    //     ~21 loads wait for each other, ~40 us of HBM latency per workgroup.
    const int span = (FR - 1) * a.hop + a.frame_len;
    const int j0 = f0 * a.hop - (a.center ? a.n_dft / 2 : 0);
    for (int e0 = 0; e0 < span; e0 += 8 * 256) {
        float v[8], pv[8];
        bool ok[8], pre[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int j = j0 + e0 + u * 256 + tid;
            if (a.center) {
                if (j < 0) j = -j;
                if (j >= N) j = 2 * (N - 1) - j;
            }
            ok[u] = j >= 0 && j < N;
            const int jc = min(max(j, 0), N - 1);
            pre[u] = a.preemph && !a.center && j > 0;
            v[u] = x[jc];
            pv[u] = x[max(jc - 1, 0)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 256 + tid;
            float r = ok[u] ? v[u] : 0.0f;
            if (pre[u] && ok[u]) r = r - 0.97f * pv[u];
            if (e < span) raw[e] = r;
        }
    }
    __syncthreads();
    // 1b. window and fold: a lane takes every 64th sample position (its two window values live in registers), a wave 8 frames
    for (int n = lane; n < a.kp; n += 64) {
        const int m2 = a.n_dft - n;
        const bool has_lo = n <= a.half && n < a.frame_len;
        const bool has_hi = n <= a.half && n > 0 && m2 != n && m2 < a.frame_len;
        const float wlo = has_lo ? a.window[n] : 0.0f;
        const float whi = has_hi ? a.window[m2] : 0.0f;
        const int nlo = has_lo ? n : 0, nhi = has_hi ? m2 : 0;
#pragma unroll
        for (int ff = 0; ff < 8; ++ff) {
            const int f = w * 8 + ff;
            const float* xr = raw + f * a.hop;
            const bool live = f0 + f < nf;
            const float lo = live ? xr[nlo] * wlo : 0.0f, hi = live ? xr[nhi] * whi : 0.0f;
            ev[f * ldk + n] = lo + hi;
            od[f * ldk + n] = lo - hi;                // (n = 0 and n = N/2 meet sin = 0 / sin(pi) in the table, as in the direct sum)
        }
    }
    __syncthreads();

    // 2. the two products, 32 samples (two MFMA k-groups) at a time; the twiddle fragments of the NEXT 32 samples are in flight
    //    while these 32 MFMAs issue (kp is a multiple of 32: no tail, no branch in the loop body -- hipcc then keeps exact vmcnt)
    const int chunks = a.kp / 32, nbt = a.nbp / 16;
    f32x4 pw[FR_MAXQ][2];
    const float* e0p = ev + i * ldk + 4 * kq;
    const float* e1p = ev + (16 + i) * ldk + 4 * kq;
    const float* o0p = od + i * ldk + 4 * kq;
    const float* o1p = od + (16 + i) * ldk + 4 * kq;
#pragma unroll
    for (int q = 0; q < FR_MAXQ; ++q) {
        const int bt = w + 4 * q;
        if (bt >= nbt) break;                          // (uniform)
        __builtin_amdgcn_sched_barrier(0);             // (one bin tile at a time: hoisting the next tiles' loads costs 25 registers each)
        const float* cp = a.cos_t + (size_t)(bt * 16 + i) * a.kp + 4 * kq;
        const float* sp = a.sin_t + (size_t)(bt * 16 + i) * a.kp + 4 * kq;
        f32x4 re0 = {0.f, 0.f, 0.f, 0.f}, re1 = re0, im0 = re0, im1 = re0;
        f32x4 c_cur[2], s_cur[2], c_nxt[2], s_nxt[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            c_cur[u] = *reinterpret_cast<const f32x4*>(cp + 16 * u);
            s_cur[u] = *reinterpret_cast<const f32x4*>(sp + 16 * u);
        }
        for (int ch = 0; ch < chunks; ++ch) {
            const int nx = ch + 1 < chunks ? ch + 1 : ch;             // (the last chunk re-reads its own fragments)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                c_nxt[u] = *reinterpret_cast<const f32x4*>(cp + 32 * nx + 16 * u);
                s_nxt[u] = *reinterpret_cast<const f32x4*>(sp + 32 * nx + 16 * u);
            }
            f32x4 e0[2], e1[2], o0[2], o1[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                e0[u] = *reinterpret_cast<const f32x4*>(e0p + 32 * ch + 16 * u);
                e1[u] = *reinterpret_cast<const f32x4*>(e1p + 32 * ch + 16 * u);
                o0[u] = *reinterpret_cast<const f32x4*>(o0p + 32 * ch + 16 * u);
                o1[u] = *reinterpret_cast<const f32x4*>(o1p + 32 * ch + 16 * u);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    re0 = __builtin_amdgcn_mfma_f32_16x16x4f32(e0[u][j], c_cur[u][j], re0, 0, 0, 0);
                    re1 = __builtin_amdgcn_mfma_f32_16x16x4f32(e1[u][j], c_cur[u][j], re1, 0, 0, 0);
                    im0 = __builtin_amdgcn_mfma_f32_16x16x4f32(o0[u][j], s_cur[u][j], im0, 0, 0, 0);
                    im1 = __builtin_amdgcn_mfma_f32_16x16x4f32(o1[u][j], s_cur[u][j], im1, 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u) { c_cur[u] = c_nxt[u]; s_cur[u] = s_nxt[u]; }
        }
        pw[q][0] = (re0 * re0 + im0 * im0) * a.power_scale;
        pw[q][1] = (re1 * re1 + im1 * im1) * a.power_scale;
    }
    __syncthreads();                                   // every wave is done with e / o: the power spectrum goes over them
    float* P = sm;                                     // [FR][ldp]; accumulator register r of a lane = frame 4 kq + r, bin 16 bt + i
#pragma unroll
    for (int q = 0; q < FR_MAXQ; ++q) {
        const int bt = w + 4 * q;
        if (bt >= nbt) break;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) P[(mt * 16 + 4 * kq + r) * ldp + bt * 16 + i] = pw[q][mt][r];
    }
    __syncthreads();
    // filterbank + log: tile of 16 filters x both frame tiles per wave turn
    float vmax = -__builtin_inff();
    const int fsteps = a.nbp / 16;
    for (int mtile = w; mtile < a.mp / 16; mtile += 4) {
        const float* fp = a.filt_t + (size_t)(mtile * 16 + i) * a.nbp + 4 * kq;
        const float* p0 = P + i * ldp + 4 * kq;
        const float* p1 = P + (16 + i) * ldp + 4 * kq;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        for (int st0 = 0; st0 < fsteps; st0 += 8) {        // eight 16-bin steps of filter fragments in flight (L2 hits)
            f32x4 fr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) fr[u] = *reinterpret_cast<const f32x4*>(fp + 16 * min(st0 + u, fsteps - 1));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (st0 + u >= fsteps) break;               // (uniform)
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(p0 + 16 * (st0 + u)), a1 = *reinterpret_cast<const f32x4*>(p1 + 16 * (st0 + u));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], fr[u][j], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], fr[u][j], acc1, 0, 0, 0);
                }
            }
        }
        const int m = mtile * 16 + i;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = f0 + mt * 16 + 4 * kq + r;
                if (f >= nf || m >= a.n_filt) continue;
                const float acc = mt == 0 ? acc0[r] : acc1[r];
                float v;
                if (a.mode == MODE_MFCC) v = 10.0f * log10f(fmaxf(acc, 1e-10f));
                else v = 10.0f * log10f(acc == 0.0f ? 2.220446049250313e-16f : acc);
                a.logmel[((size_t)b * a.t_full + f) * a.n_filt + m] = v;
                vmax = fmaxf(vmax, v);
            }
    }
    if (a.mode == MODE_MFCC) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if (lane == 0 && vmax > -__builtin_inff()) atomicMax(a.umax + b, float_key(vmax));
    }
}

// WORK QUEUE: a workgroup pulls (utterance, 32-frame tile) items until the counter runs out.  Launched beside the forward
// recurrence (amdspeech_lstm_beside_forward) only the workgroups dealt to the idle XCDs get a CU; they drain the queue, and the
// rest start when the recurrence ends, find it empty and leave.  On an idle chip every workgroup takes its share.
template <int FR_MAXQ>
__global__ __launch_bounds__(256) void frontend_frames_mfma_kernel(FrameArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ int s_item;
    for (;;) {
        __syncthreads();                              // (the previous item's LDS, s_item)
        if (threadIdx.x == 0) s_item = (int)atomicAdd(a.queue, 1u);
        __syncthreads();
        const int item = s_item;
        if (item >= a.n_items) return;
        const int b = item / a.tiles_per_utt, f0 = (item % a.tiles_per_utt) * FR;
        const int nf = a.nframes[b];
        if (f0 < nf) frontend_frames_item<FR_MAXQ>(a, sm, b, f0, nf);
    }
}

// ---- kernel 2: per-utterance statistics -----------------------------------------
// mfcc: stat[b*n_filt] = max over (t, m);  fbank: stat[b*n_filt + m] = mean over t.
__global__ __launch_bounds__(256) void frontend_stats_kernel(const float* __restrict__ logmel, const int* __restrict__ nframes,
                                                             int t_full, int n_filt, int mode, double* __restrict__ stat) {
    __shared__ double red[256];
    const int b = blockIdx.x;
    const int nf = nframes[b];
    const float* x = logmel + (size_t)b * t_full * n_filt;
    if (mode == MODE_MFCC) {
        float m = -__builtin_inff();
        for (long i = threadIdx.x; i < (long)nf * n_filt; i += 256) m = fmaxf(m, x[i]);
        red[threadIdx.x] = m;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
            __syncthreads();
        }
        if (threadIdx.x == 0) stat[(size_t)b * n_filt] = red[0];
    } else {
        // 256 threads = rows of 256/n_filt... use thread -> (rowgroup, m)
        const int groups = 256 / n_filt;                 // 6 for 40 filters
        const int m = threadIdx.x % n_filt, gidx = threadIdx.x / n_filt;
        double acc = 0.0;
        if (gidx < groups)
            for (int t = gidx; t < nf; t += groups) acc += (double)x[(size_t)t * n_filt + m];
        red[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < n_filt) {
            double s = 0.0;
            for (int g2 = 0; g2 < groups; ++g2) s += red[g2 * n_filt + threadIdx.x];
            stat[(size_t)b * n_filt + threadIdx.x] = nf > 0 ? s / nf : 0.0;
        }
    }
}

// ---- kernel 3a (mfcc): clamp + DCT-II -> feat [t_max][B][n_mfcc] --------------------
// The DCT matrix sits in LDS transposed ([m][q]: lanes = consecutive q read consecutive words); a workgroup walks
// DCT_ROWS rows (t, b), one per wave at a time.  Sum over m in ascending order, as before.
constexpr int DCT_ROWS = 32;
__global__ __launch_bounds__(256) void mfcc_dct_kernel(const float* __restrict__ logmel, const int* __restrict__ nframes,
                                                       const unsigned* __restrict__ umax, const float* __restrict__ dct,
                                                       int t_full, int t_max, int B, int n_mfcc, float* __restrict__ feat) {
    extern __shared__ float dsm[];                        // [N_MELS][n_mfcc] then [4][N_MELS]
    float* dt = dsm;
    float (*row)[N_MELS] = reinterpret_cast<float (*)[N_MELS]>(dsm + (size_t)N_MELS * n_mfcc);
    for (int i = threadIdx.x; i < N_MELS * n_mfcc; i += 256) {
        const int q = i / N_MELS, m = i % N_MELS;         // (coalesced read of dct [q][m])
        dt[m * n_mfcc + q] = dct[i];
    }
    __syncthreads();
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long rows = (long)t_max * B;
    // the wave's eight rows from memory first, all loads in flight together (one row at a time is eight HBM round trips in a row)
    float xa[DCT_ROWS / 4], xb[DCT_ROWS / 4];
#pragma unroll
    for (int it = 0; it < DCT_ROWS / 4; ++it) {
        const long r = ((long)blockIdx.x * (DCT_ROWS / 4) + it) * 4 + w;
        const long rc = r < rows ? r : rows - 1;
        const int t = rc / B, b = rc % B;
        const int tc = t < t_full ? t : t_full - 1;       // (rows past the utterance are zero-filled below; keep the address valid)
        const float* x = logmel + ((size_t)b * t_full + tc) * N_MELS;
        xa[it] = x[lane]; xb[it] = x[lane + 64];
    }
#pragma unroll
    for (int it = 0; it < DCT_ROWS / 4; ++it) {
        const long r = ((long)blockIdx.x * (DCT_ROWS / 4) + it) * 4 + w;              // row = t*B + b
        if (r >= rows) return;
        const int t = r / B, b = r % B;
        float* out = feat + r * n_mfcc;
        if (t >= nframes[b]) { for (int q = lane; q < n_mfcc; q += 64) out[q] = 0.f; continue; }
        const float floor_db = key_float(umax[b]) - 80.0f;
        row[w][lane] = fmaxf(xa[it], floor_db);
        row[w][lane + 64] = fmaxf(xb[it], floor_db);
        __builtin_amdgcn_wave_barrier();
        for (int q = lane; q < n_mfcc; q += 64) {
            float acc = 0.f;
#pragma unroll 8
            for (int m = 0; m < N_MELS; ++m) acc += dt[m * n_mfcc + q] * row[w][m];
            out[q] = acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// The same on the matrix cores (default with the MFMA frame kernel): rows x [128 mels] . [128 x n_mfcc] is 96 MFMAs per 16 rows.
// A wave owns 16 output rows (row = t * B + b); its log-mel fragments come straight from memory (one b128 per lane and 16 mels,
// all eight in flight), clamped against the utterance's floor on the way in; the DCT fragments from L1/L2 (20 KiB matrix).
// ~50 -> ~12 us for 32 x 1001 rows: it runs right behind the forward recurrence (the frame kernel's queue drains beside it), where
// the old kernel met the output Linear for 50 us.
__global__ __launch_bounds__(256) void mfcc_dct_mfma_kernel(const float* __restrict__ logmel, const int* __restrict__ nframes,
                                                            const unsigned* __restrict__ umax, const float* __restrict__ dct,
                                                            int t_full, int t_max, int B, int n_mfcc, float* __restrict__ feat) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, i = lane & 15, kq = lane >> 4;
    const long rows = (long)t_max * B;
    const long r = ((long)blockIdx.x * 4 + w) * 16 + i;              // this lane's A row
    const long rc = r < rows ? r : rows - 1;
    const int t = (int)(rc / B), b = (int)(rc % B);
    const bool live = r < rows && t < nframes[b];
    const int tc = t < t_full ? t : t_full - 1;
    const float floor_db = key_float(umax[b]) - 80.0f;
    const float* x = logmel + ((size_t)b * t_full + tc) * N_MELS + 4 * kq;
    f32x4 xa[N_MELS / 16];
#pragma unroll
    for (int s = 0; s < N_MELS / 16; ++s) xa[s] = *reinterpret_cast<const f32x4*>(x + 16 * s);
#pragma unroll
    for (int s = 0; s < N_MELS / 16; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) xa[s][j] = live ? fmaxf(xa[s][j], floor_db) : 0.0f;
    const int ntiles = (n_mfcc + 15) / 16;
    for (int nt = 0; nt < ntiles; ++nt) {
        const int q = nt * 16 + i;                                    // this lane's B column (an MFCC index)
        const float* dq = dct + (size_t)(q < n_mfcc ? q : 0) * N_MELS + 4 * kq;
        f32x4 db[N_MELS / 16];
#pragma unroll
        for (int s = 0; s < N_MELS / 16; ++s) db[s] = *reinterpret_cast<const f32x4*>(dq + 16 * s);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < N_MELS / 16; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s][j], q < n_mfcc ? db[s][j] : 0.0f, acc, 0, 0, 0);
        // accumulator register rr: output row 4 kq + rr of the wave's 16, column q = nt * 16 + i
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const long ro = ((long)blockIdx.x * 4 + w) * 16 + 4 * kq + rr;
            if (ro < rows && q < n_mfcc) feat[ro * n_mfcc + q] = acc[rr];
        }
    }
}

// ---- kernel 3b (fbank): mean-normalise, delta, delta-delta ---------------------------
// pass 0: static = logmel - (mean + 1e-8) in place.  pass 1: d1 = delta(static).
// pass 2: feat[t][b][0:40|40:80|80:120] = static | d1 | delta(d1), t < t_max.
__device__ __forceinline__ float savgol9(const float* __restrict__ x, int t, int nf, int stride) {
    // librosa>=0.6 delta: interior sum_k k*x[t+k]/60; first/last 4 frames use the end windows' slope
    int c = t;
    if (c < 4) c = 4;
    if (c > nf - 5) c = nf - 5;
    float acc = 0.f;
#pragma unroll
    for (int k = -4; k <= 4; ++k) acc += (float)k * x[(size_t)(c + k) * stride];
    return acc * (1.0f / 60.0f);
}

__global__ void fbank_pass_kernel(float* __restrict__ logmel, float* __restrict__ d1, const int* __restrict__ nframes,
                                  const double* __restrict__ stat, int t_full, int t_max, int B, int pass,
                                  float* __restrict__ feat) {
    const int b = blockIdx.y;
    const int nf = nframes[b];
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = i / N_FILT, m = i % N_FILT;
    float* st = logmel + (size_t)b * t_full * N_FILT;
    float* dd = d1 + (size_t)b * t_full * N_FILT;
    if (pass == 0) {
        if (t < nf) st[(size_t)t * N_FILT + m] = (float)((double)st[(size_t)t * N_FILT + m] - (stat[(size_t)b * N_FILT + m] + 1e-8));
    } else if (pass == 1) {
        if (t < nf) dd[(size_t)t * N_FILT + m] = savgol9(st + m, t, nf, N_FILT);
    } else {
        if (t >= t_max) return;
        float* out = feat + ((size_t)t * B + b) * (3 * N_FILT);
        if (t < nf) {
            out[m] = st[(size_t)t * N_FILT + m];
            out[N_FILT + m] = dd[(size_t)t * N_FILT + m];
            out[2 * N_FILT + m] = savgol9(dd + m, t, nf, N_FILT);
        } else {
            out[m] = 0.f; out[N_FILT + m] = 0.f; out[2 * N_FILT + m] = 0.f;
        }
    }
}

// Host-side length vectors reach the device as KERNEL ARGUMENTS of a one-block kernel (no pageable
// H2D copy, no stream synchronisation on the hot path); batches wider than META_MAX fall back to a copy.
constexpr int META_MAX = 256;
struct MetaArg { int v[2 * META_MAX]; };
__global__ void write_meta_kernel(MetaArg m, int* dst, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = m.v[i];
}

static int run_frontend(hipStream_t s, int mode, const float* pcm, const int* n_samples, int B, int n_max,
                        int sr, int n_mfcc, int t_max, float* feat, int* n_frames, void* ws) {
    AS_CHECK_ARG(pcm && n_samples && feat && n_frames && ws, "frontend: null pointer");
    AS_CHECK_ARG(B > 0 && n_max > 0 && sr >= 1000 && t_max > 0, "frontend: bad shape");
    AS_CHECK_ARG(mode == MODE_FBANK || (n_mfcc >= 1 && n_mfcc <= N_MELS), "frontend: n_mfcc out of range");
    const FrontCfg c = make_cfg(mode, sr);
    // (44.1 / 48 kHz signals -- n_fft = round(0.025 sr) = 1102 / 1200 -- go through the vector-ALU frame kernel, whose LDS need
    //  grows with the frame; 2048 points = 81.9 kHz is where its 8 frames stop fitting a CU)
    AS_CHECK_ARG(c.n_dft <= 2048, "frontend: sample rate %d gives a %d-point DFT (max 2048)", sr, c.n_dft);
    const FrontLayout lo = front_layout(c, B, n_max, mode == MODE_MFCC ? N_MELS : 1);
    const Tables* tb = get_tables(c, mode == MODE_MFCC ? n_mfcc : 0);
    AS_CHECK_ARG(tb != nullptr, "frontend: could not place the constant tables in device memory");
    char* w = static_cast<char*>(ws);
    // per-call device copies of the small tables and of the length vectors
    std::vector<int> meta(2 * B);
    for (int b = 0; b < B; ++b) {
        const int n = n_samples[b];
        AS_CHECK_ARG(n >= 0 && n <= n_max, "frontend: n_samples[%d]=%d outside [0,%d]", b, n, n_max);
        AS_CHECK_ARG(n == 0 || mode == MODE_FBANK || n > c.n_dft / 2, "frontend: utterance %d shorter than the reflect padding", b);
        meta[b] = n;
        meta[B + b] = n_frames[b] = num_frames(c, n);
        AS_CHECK_ARG(mode == MODE_MFCC || n_frames[b] == 0 || n_frames[b] >= 9, "frontend: fbank delta needs >= 9 frames (utterance %d)", b);
    }
    int* d_n = reinterpret_cast<int*>(w + lo.total);          // workspace_bytes() reserves 2*B ints past `total`
    if (B <= META_MAX) {
        MetaArg ma;
        for (int i = 0; i < 2 * B; ++i) ma.v[i] = meta[i];
        hipLaunchKernelGGL(write_meta_kernel, dim3(1), dim3(256), 0, s, ma, d_n, 2 * B);
    } else {
        AS_CHECK_HIP(hipMemcpyAsync(d_n, meta.data(), 2 * B * sizeof(int), hipMemcpyHostToDevice, s));
        AS_CHECK_HIP(hipStreamSynchronize(s));                  // `meta` is a stack-scoped staging buffer
    }
    unsigned* ctl = reinterpret_cast<unsigned*>(w + lo.ctl);
    unsigned* umax = ctl + 16;
    AS_CHECK_HIP(hipMemsetAsync(ctl, 0, 64 + (size_t)B * sizeof(unsigned), s));

    FrameArgs a;
    a.pcm = pcm; a.nsamp = d_n; a.nframes = d_n + B;
    a.twiddle = tb->dev; a.window = tb->dev + tb->o_window; a.filt = tb->dev + tb->o_filt; a.umax = umax;
    a.logmel = reinterpret_cast<float*>(w + lo.logmel);
    a.n_max = n_max; a.t_full = lo.t_full; a.hop = c.hop; a.n_dft = c.n_dft; a.frame_len = c.frame_len;
    a.n_bins = c.n_bins; a.n_filt = c.n_filt; a.center = c.center; a.preemph = mode == MODE_FBANK; a.mode = mode;
    a.power_scale = c.power_scale;
    a.cos_t = tb->dev + tb->o_cos; a.sin_t = tb->dev + tb->o_sin; a.filt_t = tb->dev + tb->o_filt_t;
    a.half = tb->half; a.kp = tb->kp; a.nbp = tb->nbp; a.mp = tb->mp;
    static const bool on_mfma = runtime_switch("AMDSPEECH_FRONTEND_MFMA", 1) != 0;
    // e / o (later the power spectrum over them: nbp <= 2 kp) and the PCM span of the 32 frames
    const size_t mfma_lds = ((size_t)2 * FR * (tb->kp + 4) + (size_t)(FR - 1) * c.hop + c.frame_len) * 4;
    constexpr int FRAMES_LDS_MAX = 160 * 1024 - 256;    // (the CU's 160 KiB less the kernel's static word)
    if (on_mfma && mfma_lds <= (size_t)FRAMES_LDS_MAX) {      // (sample rates above 32 kHz: the vector-ALU kernel)
        const size_t lds = mfma_lds;
        static unsigned long long frames_lds_seen = 0;
        if (DeviceOnce once{&frames_lds_seen}) {     // (an 800-point DFT needs 150 KiB)
            AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(frontend_frames_mfma_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, FRAMES_LDS_MAX));
            AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(frontend_frames_mfma_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, FRAMES_LDS_MAX));
            AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(frontend_frames_mfma_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, FRAMES_LDS_MAX));
            once.done();
        }
        a.queue = ctl;
        a.tiles_per_utt = ceil_div(lo.t_full, FR);
        a.n_items = a.tiles_per_utt * B;
        // two workgroups per CU's worth (what the LDS allows at 16 kHz), never more than there are items
        static const int max_wgs = dev_knob("AMDSPEECH_FRONTEND_WGS", 512);     // (dev: queue width)
        const int wgs = a.n_items < max_wgs ? a.n_items : (max_wgs > 0 ? max_wgs : 512);
        const int nbt = tb->nbp / 16;
        if (nbt <= 16) hipLaunchKernelGGL(frontend_frames_mfma_kernel<4>, dim3(wgs), dim3(256), lds, s, a);
        else if (nbt <= 20) hipLaunchKernelGGL(frontend_frames_mfma_kernel<5>, dim3(wgs), dim3(256), lds, s, a);
        else hipLaunchKernelGGL(frontend_frames_mfma_kernel<9>, dim3(wgs), dim3(256), lds, s, a);
    } else {
        const size_t lds = ((size_t)c.frame_len * FPB + (size_t)c.n_dft * 2 + (size_t)c.n_bins * FPB) * 4;
        static unsigned long long valu_lds_seen = 0;
        if (DeviceOnce once{&valu_lds_seen}) {      // (a 1200-point frame needs 67 KiB, a 2048-point one 115 KiB)
            AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(frontend_frames_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FRAMES_LDS_MAX));
            once.done();
        }
        hipLaunchKernelGGL(frontend_frames_kernel, dim3(ceil_div(lo.t_full, FPB), B), dim3(256), lds, s, a);
    }
    double* stat = reinterpret_cast<double*>(w + lo.stat);
    if (mode == MODE_MFCC) {      // (the per-utterance maximum came out of the frames kernel)
        const size_t dlds = ((size_t)N_MELS * n_mfcc + 4 * N_MELS) * sizeof(float);
        static unsigned long long dct_lds_seen = 0;
        if (DeviceOnce once{&dct_lds_seen}) {       // (n_mfcc = 128 needs 66 KiB)
            AS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mfcc_dct_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)(((size_t)N_MELS * N_MELS + 4 * N_MELS) * sizeof(float))));
            once.done();
        }
        if (on_mfma)
            hipLaunchKernelGGL(mfcc_dct_mfma_kernel, dim3(ceil_div((long)t_max * B, 64)), dim3(256), 0, s, a.logmel, a.nframes,
                               umax, tb->dev + tb->o_dct, lo.t_full, t_max, B, n_mfcc, feat);
        else
            hipLaunchKernelGGL(mfcc_dct_kernel, dim3(ceil_div((long)t_max * B, DCT_ROWS)), dim3(256), dlds, s, a.logmel, a.nframes,
                               umax, tb->dev + tb->o_dct, lo.t_full, t_max, B, n_mfcc, feat);
    } else {
        hipLaunchKernelGGL(frontend_stats_kernel, dim3(B), dim3(256), 0, s, a.logmel, a.nframes, lo.t_full, c.n_filt, mode, stat);
        float* d1 = reinterpret_cast<float*>(w + lo.d1);
        const int tt = lo.t_full > t_max ? lo.t_full : t_max;
        dim3 grid(ceil_div((long)tt * N_FILT, 256), B);
        for (int pass = 0; pass < 3; ++pass)
            hipLaunchKernelGGL(fbank_pass_kernel, grid, dim3(256), 0, s, a.logmel, d1, a.nframes, stat, lo.t_full, t_max,
                               B, pass, feat);
    }
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

}  // namespace amdspeech

using namespace amdspeech;

// ------------------------------------------------------------------- resampler
// Band-limited sinc interpolation with the "kaiser_best" filter of resampy (the resampler behind
// librosa.load(sr=22050) in the librosa versions contemporary with the reference, util/audioprocessor.py:49):
// 64 zero crossings, 512 table entries per crossing, roll-off 0.9475937167399596, Kaiser beta
// 14.769656459379492, linear interpolation between table entries, filter widened by the rate ratio when
// down-sampling.  resampy / librosa are not importable here, so this follows their published algorithm
// (parity unpinned); output length = ceil(n * ratio) as librosa.resample(fix=True) pads it.
namespace amdspeech {
constexpr int RS_ZEROS = 64, RS_PREC = 9, RS_TABLE = 1 << RS_PREC, RS_NWIN = RS_ZEROS * RS_TABLE + 1;
constexpr double RS_ROLLOFF = 0.9475937167399596, RS_BETA = 14.769656459379492;

__device__ double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    const double q = 0.25 * x * x;
    for (int k = 1; k < 200; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-17 * sum) break;
    }
    return sum;
}

// win[i] = kaiser(i) * rolloff * sinc(rolloff * i / 512), scaled by min(1, ratio); delta[i] = win[i+1] - win[i]
__global__ void resample_table_kernel(float* __restrict__ win, float* __restrict__ delta, double gain) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= RS_NWIN) return;
    auto tap = [&](int j) -> double {
        if (j >= RS_NWIN) return 0.0;
        const double r = (double)j / (double)(RS_NWIN - 1);
        const double kaiser = bessel_i0(RS_BETA * sqrt(fmax(0.0, 1.0 - r * r))) / bessel_i0(RS_BETA);
        const double z = RS_ROLLOFF * (double)j / (double)RS_TABLE * M_PI;
        const double sinc = j == 0 ? 1.0 : sin(z) / z;
        return gain * RS_ROLLOFF * sinc * kaiser;
    };
    const double w0 = tap(i), w1 = tap(i + 1);
    win[i] = (float)w0;
    delta[i] = i + 1 < RS_NWIN ? (float)(w1 - w0) : 0.0f;
}

__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, const int* __restrict__ n_in,
                                                       int in_stride, float* __restrict__ y, int out_stride,
                                                       double ratio, const float* __restrict__ win,
                                                       const float* __restrict__ delta) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= out_stride) return;
    const int n_orig = n_in[b];
    const int n_out = (int)((double)n_orig * ratio);           // samples resampy computes; the rest is padding
    float acc = 0.0f;
    if (t < n_out) {
        const float* xb = x + (size_t)b * in_stride;
        const double scale = ratio < 1.0 ? ratio : 1.0;
        const int step = (int)(scale * RS_TABLE);
        const double tr = (double)t / ratio;
        const int n = (int)tr;
        double frac = scale * (tr - (double)n);
        double idx = frac * RS_TABLE;
        int off = (int)idx;
        float eta = (float)(idx - (double)off);
        int cnt = min(n + 1, (RS_NWIN - off) / step);
        for (int i = 0; i < cnt; ++i) {                        // left wing: x[n], x[n-1], ...
            const int j = off + i * step;
            acc += (win[j] + eta * delta[j]) * xb[n - i];
        }
        frac = scale - frac;
        idx = frac * RS_TABLE;
        off = (int)idx;
        eta = (float)(idx - (double)off);
        cnt = min(n_orig - n - 1, (RS_NWIN - off) / step);
        for (int k = 0; k < cnt; ++k) {                        // right wing: x[n+1], x[n+2], ...
            const int j = off + k * step;
            acc += (win[j] + eta * delta[j]) * xb[n + k + 1];
        }
    }
    y[(size_t)b * out_stride + t] = acc;
}
}  // namespace amdspeech

extern "C" size_t amdspeech_resample_workspace_bytes(int B) {
    return B > 0 ? amdspeech::align_up((size_t)2 * amdspeech::RS_NWIN * sizeof(float), 256) + amdspeech::align_up((size_t)B * 4, 256) : 0;
}

extern "C" int amdspeech_resample_num_samples(int n_samples, int rate_in, int rate_out) {
    if (n_samples < 0 || rate_in <= 0 || rate_out <= 0) return AMDSPEECH_EINVAL;
    return (int)ceil((double)n_samples * (double)rate_out / (double)rate_in);
}

extern "C" int amdspeech_resample(void* stream, const float* pcm, const int* n_samples, int B, int n_max, int rate_in,
                                  int rate_out, float* out, int out_max, void* ws) {
    using namespace amdspeech;
    AS_CHECK_ARG(pcm && n_samples && out && ws, "resample: null pointer");
    AS_CHECK_ARG(B > 0 && n_max > 0 && out_max > 0 && rate_in > 0 && rate_out > 0, "resample: bad shape");
    const double ratio = (double)rate_out / (double)rate_in;
    for (int b = 0; b < B; ++b) {
        AS_CHECK_ARG(n_samples[b] >= 0 && n_samples[b] <= n_max, "resample: n_samples[%d] = %d exceeds n_max %d", b, n_samples[b], n_max);
        AS_CHECK_ARG((int)ceil((double)n_samples[b] * ratio) <= out_max, "resample: output row %d needs more than %d samples", b, out_max);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* win = static_cast<float*>(ws);
    float* delta = win + RS_NWIN;
    int* n_dev = reinterpret_cast<int*>(static_cast<char*>(ws) + align_up((size_t)2 * RS_NWIN * sizeof(float), 256));
    hipLaunchKernelGGL(resample_table_kernel, dim3(ceil_div(RS_NWIN, 256)), dim3(256), 0, s, win, delta,
                       ratio < 1.0 ? ratio : 1.0);
    // lengths travel as kernel arguments of a one-block kernel (stream-ordered, no host sync)
    for (int b0 = 0; b0 < B; b0 += 2 * META_MAX) {
        MetaArg m;
        const int cnt = B - b0 < 2 * META_MAX ? B - b0 : 2 * META_MAX;
        for (int i = 0; i < cnt; ++i) m.v[i] = n_samples[b0 + i];
        hipLaunchKernelGGL(write_meta_kernel, dim3(1), dim3(256), 0, s, m, n_dev + b0, cnt);
    }
    hipLaunchKernelGGL(resample_kernel, dim3(ceil_div(out_max, 256), B), dim3(256), 0, s, pcm, n_dev, n_max, out, out_max,
                       ratio, win, delta);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

extern "C" size_t amdspeech_frontend_workspace_bytes(int mode, int B, int n_max, int sample_rate) {
    if ((mode != MODE_MFCC && mode != MODE_FBANK) || B <= 0 || n_max <= 0 || sample_rate < 1000) return 0;
    const FrontCfg c = make_cfg(mode, sample_rate);
    return front_layout(c, B, n_max, mode == MODE_MFCC ? N_MELS : 1).total + align_up((size_t)2 * B * 4, 256);
}

extern "C" int amdspeech_frontend_num_frames(int mode, int n_samples, int sample_rate) {
    if ((mode != MODE_MFCC && mode != MODE_FBANK) || sample_rate < 1000) return AMDSPEECH_EINVAL;
    return num_frames(make_cfg(mode, sample_rate), n_samples);
}

extern "C" int amdspeech_frontend_mfcc(void* stream, const float* pcm, const int* n_samples, int B, int n_max,
                                       int sample_rate, int n_mfcc, int t_max, float* feat, int* n_frames, void* ws) {
    return run_frontend(static_cast<hipStream_t>(stream), MODE_MFCC, pcm, n_samples, B, n_max, sample_rate, n_mfcc,
                        t_max, feat, n_frames, ws);
}

extern "C" int amdspeech_frontend_fbank(void* stream, const float* pcm, const int* n_samples, int B, int n_max,
                                        int sample_rate, int t_max, float* feat, int* n_frames, void* ws) {
    return run_frontend(static_cast<hipStream_t>(stream), MODE_FBANK, pcm, n_samples, B, n_max, sample_rate, 0, t_max,
                        feat, n_frames, ws);
}
