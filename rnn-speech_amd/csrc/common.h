// Shared helpers for libamdspeech (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <mutex>

#include "../../include/amdspeech.h"

namespace amdspeech {

void set_error(const char* fmt, ...);

#define AS_CHECK_ARG(cond, ...)                         \
    do {                                                \
        if (!(cond)) {                                  \
            ::amdspeech::set_error(__VA_ARGS__);        \
            return AMDSPEECH_EINVAL;                    \
        }                                               \
    } while (0)

#define AS_CHECK_HIP(expr)                                                          \
    do {                                                                            \
        hipError_t e__ = (expr);                                                    \
        if (e__ != hipSuccess) {                                                    \
            ::amdspeech::set_error("%s failed: %s (%s:%d)", #expr,                  \
                                   hipGetErrorString(e__), __FILE__, __LINE__);     \
            return AMDSPEECH_EHIP;                                                  \
        }                                                                           \
    } while (0)

#define AS_CHECK_LAUNCH() AS_CHECK_HIP(hipGetLastError())

// "once per DEVICE" for function attributes (hipFuncSetAttribute applies to the current device's copy of the kernel; up to 64
// devices per process).  The library is called from several host threads (training thread, prefetch, asynchronous decoder):
// the first caller on a device sets the attributes UNDER A LOCK and marks the device only when they have succeeded -- a second
// thread either waits for the lock or sees the mark, never a half-configured kernel.
//     if (DeviceOnce once{&seen}) { AS_CHECK_HIP(hipFuncSetAttribute(...)); once.done(); }
std::mutex& device_once_mutex();
struct DeviceOnce {
    std::unique_lock<std::mutex> lock;
    unsigned long long* seen;
    unsigned long long bit = 0;
    bool first = true;
    explicit DeviceOnce(unsigned long long* flags) : seen(flags) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return;      // (unknown device: set the attributes every time)
        bit = 1ull << dev;
        if (__atomic_load_n(seen, __ATOMIC_ACQUIRE) & bit) { first = false; return; }
        lock = std::unique_lock<std::mutex>(device_once_mutex());
        first = (__atomic_load_n(seen, __ATOMIC_ACQUIRE) & bit) == 0;
    }
    explicit operator bool() const { return first; }
    void done() { if (bit) __atomic_fetch_or(seen, bit, __ATOMIC_RELEASE); }
};

// ---- environment.  A release build reads ONLY the documented run-time switches (INTEGRATION.md "Run-time switches": kernel-family
// selection and A/B fallbacks, each covered by a test) and only through runtime_switch() -- the single getenv of the library.
// Everything else that was once tunable from the environment is a compile-time constant; AMDSPEECH_DEVTRACE (development) builds
// keep those knobs readable through dev_knob() for the scripts under tools/.
int runtime_switch(const char* name, int dflt);
#ifdef AMDSPEECH_DEVTRACE
static inline int dev_knob(const char* name, int dflt) { return runtime_switch(name, dflt); }
const char* dev_knob_str(const char* name);
#else
static inline int dev_knob(const char*, int dflt) { return dflt; }
static inline const char* dev_knob_str(const char*) { return nullptr; }
#endif

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Internal entry points shared between translation units.
int gemm_f32(hipStream_t s, bool transA, bool transB, int M, int N, int K, const float* A, int lda,
             const float* B, int ldb, float* C, int ldc, const float* bias, bool accumulate,
             float* colsum = nullptr,    // colsum[n] += sum_k B[k][n] (needs !transB), fused bias gradient
             const int* gate = nullptr, int gate_need = 0,    // device word counted down by a concurrent producer
             unsigned* gate_err = nullptr);                  // error word: bit 2 = the gate wait timed out
// gemm_skinny.hip: 1 = taken (N <= 96 or K <= 80 with M >= 256, 16-byte aligned rows), 0 = not this shape, < 0 = error
int gemm_skinny(hipStream_t s, bool transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                const float* bias, bool accumulate);
// the same file's 32k-row reduction onto a narrow output, C (+)= A^T . B with min(M, N) <= 124 (the dense layers' weight gradients)
int gemm_skinny_tn(hipStream_t s, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                   bool accumulate, float* colsum);
constexpr int GEMM_GROUP_MAX = 10;
// `count` products C_i (+)= A_i^T . B_i of ONE shape (A_i [K][M], B_i [K][N], 16-byte aligned rows) in one launch
int gemm_f32_tn_group(hipStream_t s, int count, int M, int N, int K, const float* const* A, int lda, const float* const* B,
                      int ldb, float* const* C, int ldc, float* const* colsum, bool accumulate,
                      const int* gate = nullptr, int gate_need = 0, unsigned* gate_err = nullptr);
bool gemm_f32_tn_group_ok(int M, int N, int K, const float* A, int lda, const float* B, int ldb);   // alignment / 32-bit offsets
int colsum_accumulate(hipStream_t s, const float* x, int rows, int cols, int ld, float* out);
// The same product in split precision (gemm_bf3.hip): bf16 hi / lo pairs of every f32 value, hi.hi + hi.lo + lo.hi on the bf16
// MFMA, f32 accumulation and f32 operands / result in memory.  No fused column sum, no gate.
int gemm_bf3(hipStream_t s, bool transA, bool transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
             float* C, int ldc, const float* bias, bool accumulate);
// ... and with plain bf16 operands (precision = bf16): ONE bf16 per value, one MFMA per product, f32 accumulation and f32 in memory.
int gemm_bf16(hipStream_t s, bool transA, bool transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
              float* C, int ldc, const float* bias, bool accumulate);

// gemm_bf16p.hip (round 5): ... through bf16 COPIES of the operands (k contiguous) and a 256 x 256 x 64 global_load_lds kernel.
// The two steps separately -- lstm.hip shares copies between products -- and the one-call form with caller scratch.
int bf16p_copy(hipStream_t s, const float* src, long ld, long rows, int cols, bool transpose, unsigned short* dst, long ldd, float* colsum,
               unsigned short* plain = nullptr);
size_t bf16p_partial_bytes(int M, int N, int K);
int bf16p_transpose(hipStream_t s, const unsigned short* src, long rows, int cols, unsigned short* dst, long ldd);
int bf16p_gemm(hipStream_t s, int M, int N, int K, const unsigned short* Ak, long lda, const unsigned short* Bk, long ldb, float* C, long ldc,
               const float* bias, bool accumulate, void* partial, size_t partial_bytes);
bool gemm_bf16_packed_ok(bool transA, bool transB, int M, int N, int K, int lda, int ldb);
size_t gemm_bf16_packed_scratch_bytes(int M, int N, int K);
int gemm_bf16_packed(hipStream_t s, bool transA, bool transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, const float* bias, bool accumulate, float* colsum, void* scratch, size_t scratch_bytes);

// ctc.hip: the extended targets (ext / slen / valid) of a mini-batch into a CTC workspace laid out for (T, B, C, U)
int ctc_prepare_targets(hipStream_t s, const int* dense_labels, const int* lengths, int T, int B, int C, int U, void* ws);

// Counter-based dropout multiplier shared by the LSTM kernels: returns
// mask/keep for element `idx` of stream (`seed`, `tensor`).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float uniform01(uint64_t seed, uint32_t tensor, uint32_t idx) {
    uint32_t a = mix32(idx ^ (uint32_t)seed);
    uint32_t b = mix32(a + tensor * 0x9e3779b9U + (uint32_t)(seed >> 32));
    return (float)(b >> 8) * (1.0f / 16777216.0f);
}

}  // namespace amdspeech
