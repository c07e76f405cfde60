// micro-benchmark: format 740 000 hit lines on 16 threads out of (a) ordinary memory, (b) hipHostMalloc default, (c) hipHostMalloc non-coherent
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
struct Hit { uint32_t seq; uint16_t score; int16_t diag; };
static char *u32toa(uint32_t v, char *p) { char t[12]; int n = 0; do { t[n++] = (char) ('0' + v % 10); v /= 10; } while (v); while (n) *p++ = t[--n]; return p; }
static double run(const Hit *hits, const uint32_t *ho, size_t nq, const std::vector<uint32_t> &tkeys, int T) {
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::string> blocks(T); std::vector<std::thread> pool;
    for (int t = 0; t < T; t++) pool.emplace_back([&, t]() {
        const size_t from = nq * t / T, to = nq * (t + 1) / T; std::string &block = blocks[t]; char buf[64];
        block.reserve((size_t) (ho[to] - ho[from]) * 20 + (to - from));
        for (size_t q = from; q < to; q++) {
            for (uint32_t h = ho[q]; h < ho[q + 1]; h++) {
                char *e = u32toa(tkeys[hits[h].seq], buf); *e++ = '\t'; e = u32toa(hits[h].score, e); *e++ = '\t'; e = u32toa((uint32_t) (uint16_t) hits[h].diag, e); *e++ = '\n';
                block.append(buf, (size_t) (e - buf));
            }
            block.push_back('\0');
        }
    });
    for (auto &th : pool) th.join();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main() {
    const size_t nq = 600000, nh = 740000;
    std::vector<uint32_t> tkeys(50000); for (size_t i = 0; i < tkeys.size(); i++) tkeys[i] = (uint32_t) i;
    std::vector<Hit> h(nh); std::vector<uint32_t> ho(nq + 1);
    for (size_t i = 0; i < nh; i++) h[i] = Hit{(uint32_t) (i * 2654435761u % 50000), (uint16_t) (20 + i % 40), (int16_t) (i % 190)};
    for (size_t q = 0; q <= nq; q++) ho[q] = (uint32_t) (q * nh / nq);
    for (int rep = 0; rep < 3; rep++) printf("ordinary memory, 16 threads: %.2f ms   1 thread: %.2f ms\n", run(h.data(), ho.data(), nq, tkeys, 16), run(h.data(), ho.data(), nq, tkeys, 1));
    for (unsigned flags : {(unsigned) hipHostMallocDefault, (unsigned) hipHostMallocNonCoherent}) {
        Hit *ph; uint32_t *po;
        if (hipHostMalloc((void **) &ph, nh * sizeof(Hit), flags) != hipSuccess || hipHostMalloc((void **) &po, (nq + 1) * 4, flags) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
        memcpy(ph, h.data(), nh * sizeof(Hit)); memcpy(po, ho.data(), (nq + 1) * 4);
        for (int rep = 0; rep < 3; rep++) printf("hipHostMalloc flags %u, 16 threads: %.2f ms\n", flags, run(ph, po, nq, tkeys, 16));
    }
    return 0;
}
