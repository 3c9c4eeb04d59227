// clip_by_global_norm + TF-flavoured Adam over the flat parameter vector
// (replaces /root/reference/models/AcousticModel.py:388,404-406).
//
// HBM-bound elementwise work: 28 B/param/step (read g,p,m,v; write p,m,v) plus one
// 4 B/param read for the norm.  Two launches, no host synchronisation: (1) per-block
// partial sums of squares, (2) every block re-reduces the <=1024 partials (L2-hot,
// fixed order => every block derives the identical scale) and applies the update
// with float4 accesses.
#include "common.h"

namespace amdspeech {

constexpr int SUMSQ_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
    __shared__ float red[4];
    float acc = 0.f;
    const long n4 = n / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = g4[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) acc += g[i] * g[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long n,
                                                        const float* __restrict__ partial, int nparts, float clip,
                                                        float lr_t, float b1, float b2, float eps,
                                                        float* __restrict__ norm_out) {
    __shared__ float red[4];
    __shared__ float scale_s;
    float acc = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float gn = sqrtf(red[0] + red[1] + red[2] + red[3]);
        scale_s = clip / fmaxf(gn, clip);
        if (blockIdx.x == 0) *norm_out = gn;
    }
    __syncthreads();
    const float scale = scale_s;
    const long n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
#define ADAM1(c)                                                  \
        { const float gc = gg.c * scale;                          \
          mm.c = b1 * mm.c + (1.f - b1) * gc;                     \
          vv.c = b2 * vv.c + (1.f - b2) * gc * gc;                \
          pp.c -= lr_t * mm.c / (sqrtf(vv.c) + eps); }
        ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            const float gc = g[i] * scale;
            const float mn = b1 * m[i] + (1.f - b1) * gc;
            const float vn = b2 * v[i] + (1.f - b2) * gc * gc;
            m[i] = mn; v[i] = vn;
            p[i] -= lr_t * mn / (sqrtf(vn) + eps);
        }
}

__global__ void axpy_kernel(float a, const float* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] += a * x[i];
}
__global__ void fill_kernel(float* __restrict__ y, float v, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = v;
}

}  // namespace amdspeech

using namespace amdspeech;

extern "C" size_t amdspeech_optim_workspace_bytes(long n) { (void)n; return SUMSQ_BLOCKS * sizeof(float); }

extern "C" int amdspeech_clip_adam(void* stream, float* params, const float* grads, float* m, float* v, long n,
                                   float clip, float lr_t, float beta1, float beta2, float eps, float* norm_out,
                                   void* ws) {
    AS_CHECK_ARG(params && grads && m && v && norm_out && ws && n > 0, "clip_adam: bad arguments");
    AS_CHECK_ARG(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) % 16 == 0,
                 "clip_adam: buffers must be 16-byte aligned");
    AS_CHECK_ARG(clip > 0.f, "clip_adam: clip must be positive");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int blocks = ceil_div(n / 4 + 1, 256);
    if (blocks > SUMSQ_BLOCKS) blocks = SUMSQ_BLOCKS;
    float* partial = static_cast<float*>(ws);
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, s, grads, n, partial);
    int ablocks = ceil_div(n / 4 + 1, 256);
    if (ablocks > 2048) ablocks = 2048;
    hipLaunchKernelGGL(clip_adam_kernel, dim3(ablocks), dim3(256), 0, s, params, grads, m, v, n, partial, blocks, clip,
                       lr_t, beta1, beta2, eps, norm_out);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_axpy(void* stream, float a, const float* x, float* y, long n) {
    AS_CHECK_ARG(x && y && n > 0, "axpy: bad arguments");
    int blocks = ceil_div(n, 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(axpy_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a, x, y, n);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

// ---- tf.reverse_sequence over the time axis of a time-major [T, B, H] tensor (what bidirectional_dynamic_rnn does around its
// backward-direction cell): out[t, b, :] = in[len_b - 1 - t, b, :] for t < len_b, 0 beyond.  The op is its own adjoint, so the
// same call carries gradients back.  float4 per thread; bandwidth bound (read + write once).
__global__ void reverse_seq_kernel(const float4* __restrict__ in, float4* __restrict__ out, const int* __restrict__ lengths,
                                   int T, int B, int H4, int accumulate) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)T * B * H4;
    if (i >= n) return;
    const int h = i % H4;
    const int b = (i / H4) % B;
    const int t = i / ((size_t)H4 * B);
    const int len = min(max(lengths[b], 0), T);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < len) v = in[((size_t)(len - 1 - t) * B + b) * H4 + h];
    if (accumulate) { const float4 o = out[i]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    out[i] = v;
}

extern "C" int amdspeech_reverse_sequences(void* stream, const float* in, float* out, const int* lengths, int T, int B, int H,
                                           int accumulate) {
    AS_CHECK_ARG(in && out && lengths && T > 0 && B > 0 && H > 0 && H % 4 == 0, "reverse_sequences: bad arguments (H %% 4 == 0)");
    AS_CHECK_ARG(in != out, "reverse_sequences: in place is not supported");
    const size_t n = (size_t)T * B * (H / 4);
    hipLaunchKernelGGL(reverse_seq_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(in), reinterpret_cast<float4*>(out), lengths, T, B, H / 4, accumulate);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}

extern "C" int amdspeech_fill(void* stream, float* y, float value, long n) {
    AS_CHECK_ARG(y && n > 0, "fill: bad arguments");
    int blocks = ceil_div(n, 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), y, value, n);
    AS_CHECK_LAUNCH();
    return AMDSPEECH_OK;
}
