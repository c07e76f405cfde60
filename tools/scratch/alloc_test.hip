#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now(); hipSetDevice(0); hipFree(0); double t1 = now(); printf("init %.1f ms\n", t1 - t0);
    size_t sizes[] = {1ull << 20, 64ull << 20, 256ull << 20, 1600ull << 20, 1600ull << 20, 512ull << 20};
    void *p[8];
    for (int r = 0; r < 2; r++) {
        for (int i = 0; i < 6; i++) { double a = now(); hipMalloc(&p[i], sizes[i]); double b = now(); printf("malloc %5zu MB: %.2f ms\n", sizes[i] >> 20, b - a); }
        double a = now(); hipMemset(p[3], 0, sizes[3]); hipDeviceSynchronize(); printf("memset 1.6 GB: %.2f ms\n", now() - a);
        for (int i = 0; i < 6; i++) { double a2 = now(); hipFree(p[i]); printf("free %5zu MB: %.2f ms\n", sizes[i] >> 20, now() - a2); }
    }
    void *h; double a = now(); hipHostMalloc(&h, 64 << 20); printf("hostMalloc 64 MB: %.2f ms\n", now() - a);
    a = now(); hipHostMalloc(&h, 512 << 20); printf("hostMalloc 512 MB: %.2f ms\n", now() - a);
    return 0;
}
