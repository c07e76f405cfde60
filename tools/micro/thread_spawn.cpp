// cost of spawning and joining N std::threads that do nothing (what every parallel section of a module call paid): g++ -O2 -pthread thread_spawn.cpp
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
int main() {
    for (int n : {4, 16, 32}) {
        double best = 1e9, sum = 0;
        for (int rep = 0; rep < 50; rep++) {
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> pool; for (int t = 0; t < n; t++) pool.emplace_back([] {}); for (auto &th : pool) th.join();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            best = ms < best ? ms : best; sum += ms;
        }
        printf("%d threads: spawn + join min %.3f ms, mean %.3f ms\n", n, best, sum / 50);
    }
}
