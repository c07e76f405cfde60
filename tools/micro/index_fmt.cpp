#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <cstdint>
struct DbEntry { uint32_t key; uint64_t offset; uint64_t length; };
static const char kDigits2[201] = "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
inline char *u32toa(uint32_t v, char *p) { char t[12]; int n = 12; while (v >= 100u) { const uint32_t r = v % 100u; v /= 100u; n -= 2; memcpy(t + n, kDigits2 + 2 * r, 2); } if (v >= 10u) { n -= 2; memcpy(t + n, kDigits2 + 2 * v, 2); } else t[--n] = (char) ('0' + v); memcpy(p, t + n, (size_t) (12 - n)); return p + (12 - n); }
int main(int argc, char **argv) {
    const size_t n = 600000; const size_t T = argc > 1 ? atoi(argv[1]) : 16;
    std::vector<DbEntry> entries(n); uint64_t at = 0; for (size_t i = 0; i < n; i++) { entries[i] = {(uint32_t) i, at, 17 + i % 40}; at += entries[i].length; }
    std::vector<std::string> texts(T);
    for (int rep = 0; rep < 6; rep++) {
        const auto t0 = std::chrono::steady_clock::now();
        auto fmt = [&](size_t t) {
            std::string text = std::move(texts[t]); const size_t from = n * t / T, to = n * (t + 1) / T;
            text.clear(); text.reserve((to - from) * 28);
            char buf[72];
            for (size_t i = from; i < to; i++) { const DbEntry &e = entries[i]; char *q = u32toa(e.key, buf); *q++ = '\t'; q = u32toa((uint32_t) e.offset, q); *q++ = '\t'; q = u32toa((uint32_t) e.length, q); *q++ = '\n'; text.append(buf, (size_t) (q - buf)); }
            texts[t] = std::move(text);
        };
        std::vector<std::thread> pool; for (size_t t = 0; t < T; t++) pool.emplace_back(fmt, t); for (auto &th : pool) th.join();
        printf("%zu threads: %.3f ms\n", T, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
}
