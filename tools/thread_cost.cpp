// how long do std::thread creation + join take here?  (sandboxed boxes: clone() can cost a millisecond)
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
int main() {
    for (int T : {1, 4, 16, 64}) {
        for (int rep = 0; rep < 3; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> pool;
            volatile long sink = 0;
            for (int t = 0; t < T; t++) pool.emplace_back([&sink]() { long s = 0; for (int i = 0; i < 1000; i++) s += i; sink = sink + s; });
            for (auto &th : pool) th.join();
            auto t1 = std::chrono::steady_clock::now();
            printf("%2d threads: create + join %.3f ms\n", T, std::chrono::duration<double, std::milli>(t1 - t0).count());
        }
    }
    // and 16 threads doing 5 ms of work each
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < 16; t++) pool.emplace_back([]() { auto s = std::chrono::steady_clock::now(); volatile double x = 1; while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - s).count() < 5.0) x = x * 1.0000001; });
    for (auto &th : pool) th.join();
    printf("16 threads x 5 ms of spinning: %.3f ms wall\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return 0;
}
