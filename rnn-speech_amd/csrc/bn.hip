// Optional batch normalisation of the input-layer output (reference:
// /root/reference/models/AcousticModel.py:253-259, `batch_normalization` in config.ini, off by
// default): tf.nn.moments over the BATCH axis only (per frame t, per feature h), epsilon 1e-3, no
// scale/offset, no running statistics (the same formula in training and evaluation).
//
// HBM-bound elementwise work on [T,B,H] (read + 2 writes forward, 2 reads + write backward); one
// thread per (t, h) walks the B rows (stride H: consecutive threads -> consecutive h, coalesced).
#include "common.h"

namespace amdspeech {

__global__ __launch_bounds__(256) void bn_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                     float* __restrict__ xhat, float* __restrict__ inv_std,
                                                     int T, int B, int H, float eps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)T * H) return;
    const int t = i / H, h = i % H;
    const float* xp = x + (size_t)t * B * H + h;
    float mean = 0.f;
    for (int b = 0; b < B; ++b) mean += xp[(size_t)b * H];
    mean /= B;
    float var = 0.f;
    for (int b = 0; b < B; ++b) { const float d = xp[(size_t)b * H] - mean; var += d * d; }
    var /= B;                                   // biased variance (tf.nn.moments)
    const float is = rsqrtf(var + eps);
    inv_std[i] = is;
    for (int b = 0; b < B; ++b) {
        const float v = (xp[(size_t)b * H] - mean) * is;
        const size_t o = (size_t)t * B * H + (size_t)b * H + h;
        if (xhat) xhat[o] = v;                  // xhat before y: x and y may alias
        y[o] = v;
    }
}

// dx = inv_std * (dy - mean_b(dy) - xhat * mean_b(dy * xhat))
__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                     const float* __restrict__ inv_std, float* __restrict__ dx,
                                                     int T, int B, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)T * H) return;
    const int t = i / H, h = i % H;
    const size_t base = (size_t)t * B * H + h;
    float s1 = 0.f, s2 = 0.f;
    for (int b = 0; b < B; ++b) {
        const float g = dy[base + (size_t)b * H];
        s1 += g;
        s2 += g * xhat[base + (size_t)b * H];
    }
    s1 /= B; s2 /= B;
    const float is = inv_std[i];
    for (int b = 0; b < B; ++b) {
        const size_t o = base + (size_t)b * H;
        dx[o] = is * (dy[o] - s1 - xhat[o] * s2);
    }
}

// ---- the same normalisation when the batch axis is spread over data-parallel ranks: tf.nn.moments then spans the GLOBAL
// batch, so every sum over b becomes "local sum -> all-reduce -> finish".  Three local passes forward (sum x; sum (x-mean)^2
// -- two passes like tf.nn.moments, no E[x^2]-mean^2 cancellation; normalise), two backward.
// out[t,h] = sum_b f(x[t,b,h]); f = x (mean == nullptr) or (x - mean[t,h] * inv_n)^2 with mean holding the GLOBAL sum of x
__global__ __launch_bounds__(256) void bn_sum_kernel(const float* __restrict__ x, const float* __restrict__ gsum, float inv_n,
                                                     float* __restrict__ out, int T, int B, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)T * H) return;
    const int t = i / H, h = i % H;
    const float* xp = x + (size_t)t * B * H + h;
    const float mean = gsum ? gsum[i] * inv_n : 0.f;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) { const float d = xp[(size_t)b * H] - mean; acc += gsum ? d * d : d; }
    out[i] = acc;
}
// y = (x - mean) * inv_std with the GLOBAL sums: mean = gsum / n, var = gsq / n
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gsum,
                                                       const float* __restrict__ gsq, float inv_n, float eps,
                                                       float* __restrict__ y, float* __restrict__ xhat, float* __restrict__ inv_std,
                                                       int T, int B, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)T * H) return;
    const int t = i / H, h = i % H;
    const float mean = gsum[i] * inv_n;
    const float is = rsqrtf(gsq[i] * inv_n + eps);
    inv_std[i] = is;
    for (int b = 0; b < B; ++b) {
        const size_t o = (size_t)t * B * H + (size_t)b * H + h;
        const float v = (x[o] - mean) * is;
        if (xhat) xhat[o] = v;
        y[o] = v;
    }
}
// sums[0][t,h] = sum_b dy, sums[1][t,h] = sum_b dy * xhat (local)
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                          float* __restrict__ sums, int T, int B, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)T * H) return;
    const int t = i / H, h = i % H;
    const size_t base = (size_t)t * B * H + h;
    float s1 = 0.f, s2 = 0.f;
    for (int b = 0; b < B; ++b) {
        const float g = dy[base + (size_t)b * H];
        s1 += g;
        s2 += g * xhat[base + (size_t)b * H];
    }
    sums[i] = s1;
    sums[(size_t)T * H + i] = s2;
}
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                           const float* __restrict__ inv_std, const float* __restrict__ sums,
                                                           float inv_n, float* __restrict__ dx, int T, int B, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)T * H) return;
    const int t = i / H, h = i % H;
    const size_t base = (size_t)t * B * H + h;
    const float s1 = sums[i] * inv_n, s2 = sums[(size_t)T * H + i] * inv_n, is = inv_std[i];
    for (int b = 0; b < B; ++b) {
        const size_t o = base + (size_t)b * H;
        dx[o] = is * (dy[o] - s1 - xhat[o] * s2);
    }
}

}  // namespace amdspeech

using namespace amdspeech;

extern "C" int amdspeech_batchnorm_sum(void* stream, const float* x, const float* global_sum, int n_total, float* out,
                                       int T, int B, int H) {
    AS_CHECK_ARG(x && out && T > 0 && B > 0 && H > 0 && n_total >= B, "batchnorm_sum: bad arguments");
    hipLaunchKernelGGL(bn_sum_kernel, dim3(ceil_div((long)T * H, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, global_sum, 1.0f / (float)n_total, out, T, B, H);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_batchnorm_apply(void* stream, const float* x, const float* global_sum, const float* global_sq,
                                         int n_total, float eps, float* y, float* xhat, float* inv_std, int T, int B, int H) {
    AS_CHECK_ARG(x && global_sum && global_sq && y && inv_std && T > 0 && B > 0 && H > 0 && n_total >= B && eps > 0.f,
                 "batchnorm_apply: bad arguments");
    AS_CHECK_ARG(xhat != y, "batchnorm_apply: xhat must not alias y");
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ceil_div((long)T * H, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, global_sum, global_sq, 1.0f / (float)n_total, eps, y, xhat, inv_std, T, B, H);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_batchnorm_bwd_sums(void* stream, const float* dy, const float* xhat, float* sums, int T, int B, int H) {
    AS_CHECK_ARG(dy && xhat && sums && T > 0 && B > 0 && H > 0, "batchnorm_bwd_sums: bad arguments");
    hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(ceil_div((long)T * H, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       dy, xhat, sums, T, B, H);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_batchnorm_bwd_apply(void* stream, const float* dy, const float* xhat, const float* inv_std,
                                             const float* global_sums, int n_total, float* dx, int T, int B, int H) {
    AS_CHECK_ARG(dy && xhat && inv_std && global_sums && dx && T > 0 && B > 0 && H > 0 && n_total >= B,
                 "batchnorm_bwd_apply: bad arguments");
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ceil_div((long)T * H, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       dy, xhat, inv_std, global_sums, 1.0f / (float)n_total, dx, T, B, H);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_batchnorm_fwd(void* stream, const float* x, float* y, float* xhat, float* inv_std, int T,
                                       int B, int H, float eps) {
    AS_CHECK_ARG(x && y && inv_std && T > 0 && B > 0 && H > 0 && eps > 0.f, "batchnorm_fwd: bad arguments");
    AS_CHECK_ARG(xhat != y, "batchnorm_fwd: xhat must not alias y");
    hipLaunchKernelGGL(bn_fwd_kernel, dim3(ceil_div((long)T * H, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, y, xhat, inv_std, T, B, H, eps);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_batchnorm_bwd(void* stream, const float* dy, const float* xhat, const float* inv_std,
                                       float* dx, int T, int B, int H) {
    AS_CHECK_ARG(dy && xhat && inv_std && dx && T > 0 && B > 0 && H > 0, "batchnorm_bwd: bad arguments");
    hipLaunchKernelGGL(bn_bwd_kernel, dim3(ceil_div((long)T * H, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       dy, xhat, inv_std, dx, T, B, H);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}
