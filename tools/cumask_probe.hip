// Dev tool: how does hipExtStreamCreateWithCUMask map mask bits to (XCC, SE, CU) on this box?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
#include <set>
__global__ void where(unsigned* out, long ticks) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const unsigned long long end = wall_clock64() + ticks;
    while (wall_clock64() < end) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}
int main() {
    unsigned* out; hipMalloc(&out, 4096 * 8);
    std::vector<unsigned> h(4096 * 2);
    auto run = [&](const char* name, std::vector<uint32_t> mask, int grid) {
        hipStream_t st;
        if (mask.empty()) hipStreamCreate(&st);
        else if (hipExtStreamCreateWithCUMask(&st, mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
        hipLaunchKernelGGL(where, grid, 64, 0, st, out, 1000L);   // warm
        hipStreamSynchronize(st);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, st);
        hipLaunchKernelGGL(where, grid, 64, 0, st, out, 10000L);  // 100 us spin per WG
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
        std::map<unsigned, std::set<unsigned>> per_xcc;
        for (int i = 0; i < grid; ++i) {
            const unsigned xcc = h[2 * i] & 0xF, hw = h[2 * i + 1];
            const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
        }
        int total = 0;
        printf("%-28s grid %d: %.0f us |", name, grid, ms * 1e3);
        for (auto& kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
        printf(" | distinct CUs %d\n", total);
        hipStreamDestroy(st);
    };
    run("no mask", {}, 2048);
    run("words 0-1 = ff..ff (64 bits)", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0}, 2048);
    run("every word = 0x000000ff", std::vector<uint32_t>(8, 0x000000ffu), 2048);
    run("every word = 0x11111111", std::vector<uint32_t>(8, 0x11111111u), 2048);
    run("every word = 0x00ffffff", std::vector<uint32_t>(8, 0x00ffffffu), 2048);
    run("words 0-5 = ff..ff (192 bits)", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0}, 2048);
    run("single word 0xffffffff", {0xffffffffu}, 2048);
    run("every word = 0xc0c0c0c0 (XCDs 6,7?)", std::vector<uint32_t>(8, 0xc0c0c0c0u), 2048);
    run("every word = 0x01010101 (XCD 0?)", std::vector<uint32_t>(8, 0x01010101u), 2048);
    return 0;
}
