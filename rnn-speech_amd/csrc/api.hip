// C-ABI glue: error reporting, Linear forward/backward, raw GEMM entry point.
#include "common.h"
#include <string.h>
#include <stdlib.h>

namespace amdspeech {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
std::mutex& device_once_mutex() {
    static std::mutex m;
    return m;
}

// The library's only read of the process environment (see common.h).  Callers cache the value in a function-local static, so a
// switch is read once per process -- except where a test flips it between calls (the host beam search).
static const char* env_text(const char* name) { return getenv(name); }
int runtime_switch(const char* name, int dflt) {
    const char* e = env_text(name);
    return e ? atoi(e) : dflt;
}
#ifdef AMDSPEECH_DEVTRACE
const char* dev_knob_str(const char* name) { return env_text(name); }
#endif
}  // namespace amdspeech

using namespace amdspeech;

extern "C" int amdspeech_version(void) { return 100; }
extern "C" const char* amdspeech_last_error(void) { return g_err; }

extern "C" int amdspeech_device_cu_count(void) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        set_error("no HIP device");
        return AMDSPEECH_EHIP;
    }
    return prop.multiProcessorCount;
}

extern "C" int amdspeech_gemm_f32(void* stream, int transA, int transB, int M, int N, int K, const float* A, int lda,
                                  const float* B, int ldb, float* C, int ldc, const float* bias, int accumulate) {
    return gemm_f32(static_cast<hipStream_t>(stream), transA != 0, transB != 0, M, N, K, A, lda, B, ldb, C, ldc, bias,
                    accumulate != 0);
}

extern "C" int amdspeech_linear_fwd(void* stream, const float* x, const float* w, const float* b, float* y, int M,
                                    int K, int N) {
    return gemm_f32(static_cast<hipStream_t>(stream), false, false, M, N, K, x, K, w, N, y, N, b, false);
}

extern "C" int amdspeech_linear_bwd(void* stream, const float* x, const float* w, const float* dy, float* dx,
                                    float* dw, float* db, int M, int K, int N) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    AS_CHECK_ARG(x && w && dy && dw && db, "linear_bwd: null pointer");
    if (dx)   // dx[M,K] = dy[M,N] . w[K,N]^T
        if (int rc = gemm_f32(s, false, true, M, K, N, dy, N, w, N, dx, K, nullptr, false)) return rc;
    // dw[K,N] += x[M,K]^T . dy[M,N]   and   db[N] += column sums of dy (fused into the same GEMM)
    return gemm_f32(s, true, false, K, N, M, x, K, dy, N, dw, N, nullptr, true, db);
}
