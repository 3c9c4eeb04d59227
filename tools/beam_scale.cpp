// Dev tool: wall and CPU time of amdspeech_ctc_beam_search_host for a batch of B 1001-frame utterances of random posteriors (one decode
// thread per utterance): does the host decoder scale with threads on this box?   g++ -O2 -o /tmp/beam_scale tools/beam_scale.cpp -ldl && /tmp/beam_scale 32
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <chrono>
#include <random>
#include <sys/resource.h>
typedef int (*fn_t)(const float*, const int*, int, int, int, int, int, int*, int*, float*);
static double cpu() { rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_utime.tv_sec + r.ru_utime.tv_usec * 1e-6 + r.ru_stime.tv_sec + r.ru_stime.tv_usec * 1e-6; }
static double sys() { rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_stime.tv_sec + r.ru_stime.tv_usec * 1e-6; }
int main(int argc, char** argv) {
    void* h = dlopen("rnn-speech_amd/libamdspeech.so", RTLD_NOW);
    fn_t f = (fn_t)dlsym(h, "amdspeech_ctc_beam_search_host");
    const int T = 1001, C = 80, B = atoi(argv[1]);
    std::mt19937 g(1); std::normal_distribution<float> nd;
    std::vector<float> x((size_t)T * B * C); for (auto& v : x) v = nd(g);
    std::vector<int> len(B, T), ids((size_t)B * T), ol(B); std::vector<float> lp(B);
    f(x.data(), len.data(), T, B, C, 100, 1, ids.data(), ol.data(), lp.data());
    double c0 = cpu(), s0 = sys(); auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 5; ++r) f(x.data(), len.data(), T, B, C, 100, 1, ids.data(), ol.data(), lp.data());
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("B=%d: wall %.1f ms per batch, cpu %.1f ms (sys %.1f)\n", B, dt * 200, (cpu() - c0) * 200, (sys() - s0) * 200);
}
