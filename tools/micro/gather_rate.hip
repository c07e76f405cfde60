// How many fully divergent 8-byte loads per second does a CU's vector memory path deliver?  (k_match's occupancy probes are exactly that: one random word per
// lane from an 8 MB table.)  Every lane issues `unroll` independent loads per iteration from a table of the given size; prints lane-loads/s, per CU and clock.
// build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
template <int U>
__global__ __launch_bounds__(256) void k_gather(const unsigned long long *tab, unsigned long long mask, int iters, unsigned long long *sink, int rowMode) {
    unsigned int s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    unsigned int ws = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 40503u + 7u;      // per wave: the row all its lanes probe in this step (rowMode)
    unsigned long long acc = 0;
    for (int it = 0; it < iters; it++) {
        unsigned long long v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            s = s * 1664525u + 1013904223u; ws = ws * 1664525u + 1013904223u;
            const unsigned long long anywhere = ((unsigned long long) s * 2654435761ull >> 8) & mask;
            const unsigned long long inRow = ((((unsigned long long) ws * 2654435761ull >> 8) & mask) & ~127ull) | ((s >> 9) & 127u);      // 128 words = 1 KB
            v[u] = tab[rowMode ? inRow : anywhere];
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u];
    }
    if (acc == 0x123456789ull) sink[0] = acc;
}
int main(int argc, char **argv) {
    hipDeviceProp_t p; (void) hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount; const double mhz = p.clockRate / 1e3;
    for (unsigned long long mb : {1ull, 2ull, 4ull, 8ull, 16ull, 64ull, 2048ull}) {
        const unsigned long long words = mb * (1ull << 20) / 8;
        unsigned long long *tab, *sink; (void) hipMalloc(&tab, words * 8); (void) hipMalloc(&sink, 8); (void) hipMemset(tab, 1, words * 8);
        for (int perCU : {4}) { for (int rowMode : {0, 1}) {
            const int blocks = cus * perCU, iters = 200;
            hipEvent_t a, b; (void) hipEventCreate(&a); (void) hipEventCreate(&b);
            hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(256), 0, 0, tab, words - 1, 20, sink, rowMode);
            (void) hipEventRecord(a); hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(256), 0, 0, tab, words - 1, iters, sink, rowMode); (void) hipEventRecord(b); (void) hipEventSynchronize(b);
            float ms = 0; (void) hipEventElapsedTime(&ms, a, b);
            const double loads = (double) blocks * 256 * iters * 8;
            printf("table %5llu MB, %s: %.1f G lane-loads/s = %.3f per CU per clock (%.0f MHz), %.2f TB/s of 128-byte lines if every load missed\n", mb, rowMode ? "lanes of a wave inside one 1 KB row" : "every lane anywhere            ", loads / ms / 1e6,
                   loads / (ms * 1e-3) / cus / (mhz * 1e6), mhz, loads * 128 / (ms * 1e-3) / 1e12);
        } }
        (void) hipFree(tab); (void) hipFree(sink);
    }
    return 0;
}
