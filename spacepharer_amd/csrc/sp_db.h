// MMseqs2 DB triples, module command lines and small formatting helpers shared by the module drivers
// (sp_modules.cpp: prefilter / align / rescorediagonal; sp_sets.cpp: the set-level steps around them).
#pragma once
#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <string>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <thread>
#include <vector>

#include "../../include/spacepharer_amd.h"
#include "sp_host.h"
#include "sp_modules.h"

namespace sp {

// ---------------------------------------------------------------------------------------------------------------
// MMseqs2 DB triple: <p> (or <p>.0 .. <p>.N), <p>.index, <p>.dbtype     (commons/DBReader.cpp, DBWriter.cpp)
// ---------------------------------------------------------------------------------------------------------------
struct DbEntry { uint32_t key; uint64_t offset; uint64_t length; };   // length includes the terminating NUL

// a file mapped read-only (sequence and result DBs are read once, front to back: no copy into a vector)
struct MappedFile {
    const char *p = nullptr; size_t n = 0;
    MappedFile() = default;
    MappedFile(const MappedFile &) = delete; MappedFile &operator=(const MappedFile &) = delete;
    ~MappedFile() { if (p && n) munmap(const_cast<char *>(p), n); }
    bool open(const std::string &path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st; if (fstat(fd, &st) != 0) { ::close(fd); return false; }
        n = (size_t) st.st_size;
        if (n) { void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0); if (m == MAP_FAILED) { ::close(fd); n = 0; return false; } p = static_cast<const char *>(m); }
        ::close(fd);
        return true;
    }
};

struct Db {
    std::vector<DbEntry> entries;     // sorted by data offset (DBReader::LINEAR_ACCCESS)
    int dbtype = -1;
    bool compressed() const { return dbtype >= 0 && (dbtype & (1 << 31)); }
    int type() const { return dbtype & 0x7FFFFFFF; }
    size_t dataSize() const { return size_; }

    bool load(const std::string &path, std::string &err) {
        {   // key \t offset \t length \n  -- parsed by hand and in parallel: result DBs of 10^5 spacers have 6 x 10^5 lines per index
            MappedFile idx;
            if (!idx.open(path + ".index")) { err = "cannot open " + path + ".index"; return false; }
            const char *text = idx.p; const size_t sz = idx.n;
            const size_t T = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(16, (size_t) defaultThreads()), sz / (1 << 20) + 1));
            std::vector<std::vector<DbEntry>> part(T); std::vector<int> bad(T, 0);
            auto parse = [&](size_t t) {
                size_t from = sz * t / T, to = sz * (t + 1) / T;
                if (t > 0) { while (from < sz && text[from - 1] != '\n') from++; }      // a chunk starts behind a line end ...
                while (to < sz && to > 0 && text[to - 1] != '\n') to++;                 // ... and runs to the end of its last line
                const char *p = text + from, *end = text + to;
                part[t].reserve((to - from) / 12 + 1);
                while (p < end) {
                    if (*p == '\n' || *p == '\r') { p++; continue; }
                    unsigned long long v[3];
                    for (int f = 0; f < 3; f++) {
                        if (p >= end || *p < '0' || *p > '9') { bad[t] = 1; return; }
                        unsigned long long x = 0;
                        while (p < end && *p >= '0' && *p <= '9') x = x * 10 + (unsigned long long) (*p++ - '0');
                        v[f] = x;
                        if (f < 2) { if (p >= end || *p != '\t') { bad[t] = 1; return; } p++; }
                    }
                    while (p < end && *p != '\n') p++;
                    DbEntry e; e.key = (uint32_t) v[0]; e.offset = v[1]; e.length = v[2];
                    part[t].push_back(e);
                }
            };
            if (T == 1) parse(0);
            else runTasks(T, parse);
            for (size_t t = 0; t < T; t++) if (bad[t]) { err = "malformed index line in " + path + ".index"; return false; }
            size_t total = 0; for (auto &v : part) total += v.size();
            entries.reserve(total);
            for (auto &v : part) entries.insert(entries.end(), v.begin(), v.end());
        }
        std::vector<std::string> files;
        struct stat st;
        if (stat(path.c_str(), &st) == 0) files.push_back(path);
        else for (int i = 0;; i++) { std::string f = path + "." + std::to_string(i); if (stat(f.c_str(), &st) != 0) break; files.push_back(f); }
        if (files.empty()) { err = "cannot open data file of " + path; return false; }
        if (files.size() == 1) {
            if (!map_.open(files[0])) { err = "cannot read " + files[0]; return false; }
            base_ = map_.p; size_ = map_.n;
        } else {      // split data files <p>.0 .. <p>.N are one logical file (DBWriter.cpp:117-123): read back to back
            for (const std::string &f : files) {
                MappedFile m;
                if (!m.open(f)) { err = "cannot read " + f; return false; }
                owned_.insert(owned_.end(), m.p, m.p + m.n);
            }
            base_ = owned_.data(); size_ = owned_.size();
        }
        std::ifstream t(path + ".dbtype", std::ios::binary);
        if (t.good()) { int32_t v = 0; t.read((char *) &v, 4); dbtype = v; }
        for (const DbEntry &e : entries) if (e.length == 0 || e.length > size_ || e.offset > size_ - e.length) { err = "index of " + path + " points past the data file"; return false; }      // (no offset + length: a length of 2^64 - 1 wraps it -- found by tests/test_host_fuzz.py under ASan)
        bool sorted = true;
        for (size_t i = 1; i < entries.size() && sorted; i++) sorted = entries[i - 1].offset <= entries[i].offset;
        if (!sorted) std::sort(entries.begin(), entries.end(), [](const DbEntry &a, const DbEntry &b) { return a.offset < b.offset; });
        return true;
    }
    // sequence entry = residues + '\n' + '\0'  (DBReader.h:191-211)
    uint64_t seqLen(size_t i) const { return entries[i].length >= 2 ? entries[i].length - 2 : 0; }
    const char *ptr(size_t i) const { return base_ + entries[i].offset; }
private:
    MappedFile map_; std::vector<char> owned_; const char *base_ = nullptr; size_t size_ = 0;
};

// One module call can be spread over the GPUs of a node (SP_GPUS=N): the parent starts N worker processes, worker r takes slice r of the entries (DBReader::decomposeDomainByAminoAcid arithmetic) on GPU r and writes
// the data file <out>.<r> -- the split-data-file form MMseqs2 DBs have anyway (DBWriter.cpp:117-123) -- plus a private index;
// the parent merges the indexes into <out>.index (offsets shifted by the sizes of the preceding parts) and writes .dbtype last.
struct Part { int rank = 0, count = 1; };
inline Part currentPart() {
    Part p;
    if (const char *e = getenv("SP_PART")) { int r = 0, n = 1; if (sscanf(e, "%d/%d", &r, &n) == 2 && n >= 1 && r >= 0 && r < n) { p.rank = r; p.count = n; } }
    return p;
}
// contiguous slice of `lengths.size()` entries for this part, balanced by data size (commons/DBReader.cpp:1171-1213)
inline void partRange(const std::vector<uint64_t> &lengths, const Part &p, size_t &from, size_t &to) {
    const size_t n = lengths.size();
    from = 0; to = n;
    if (p.count <= 1) return;
    uint64_t total = 0; for (uint64_t l : lengths) total += l;
    if (n <= (size_t) p.count) { from = std::min<size_t>(n, (size_t) p.rank); to = std::min<size_t>(n, from + 1); return; }
    const uint64_t chunk = (total + (uint64_t) p.count - 1) / (uint64_t) p.count;
    std::vector<size_t> per((size_t) p.count, 0);
    size_t cur = 0; uint64_t acc = 0;
    for (uint64_t l : lengths) { if (acc >= chunk && cur + 1 < (size_t) p.count) { acc = 0; cur++; } acc += l; per[cur]++; }
    for (int r = 0; r < p.rank; r++) from += per[(size_t) r];
    to = from + per[(size_t) p.rank];
}

// Buffers a module call formats its output into, kept (with their capacity) from call to call on the calling thread.  A fresh process does not
// care; a resident server would otherwise first-touch the same tens of megabytes again in every call, and page faults of several threads of one
// process serialise (on the virtualised MI355X boxes: 30 of the 75 ms of a served 100 000-spacer `prefilter` call).
struct BufferPool {
    std::vector<std::string> strings; std::vector<std::vector<DbEntry>> entryLists;
    static BufferPool &mine() { static thread_local BufferPool p; return p; }
    std::string takeString() { if (strings.empty()) return std::string(); std::string s = std::move(strings.back()); strings.pop_back(); s.clear(); return s; }
    void give(std::string &&s) { if (s.capacity() >= (1u << 16) && strings.size() < 256) { s.clear(); strings.push_back(std::move(s)); } }
    std::vector<DbEntry> takeEntries() { if (entryLists.empty()) return {}; std::vector<DbEntry> v = std::move(entryLists.back()); entryLists.pop_back(); v.clear(); return v; }
    void give(std::vector<DbEntry> &&v) { if (v.capacity() >= 4096 && entryLists.size() < 8) { v.clear(); entryLists.push_back(std::move(v)); } }
};

inline char *u32toa(uint32_t v, char *p);
inline char *u64toa(uint64_t v, char *p);
struct DbWriter {
    std::string path; int dbtype;
    Part part = currentPart();
    std::vector<char> data; std::vector<DbEntry> entries = BufferPool::mine().takeEntries();
    // strings for addBlock(), one per chunk, from the calling thread's pool (call before the workers start)
    static std::vector<std::string> pooledBlocks(size_t n) { std::vector<std::string> v(n); for (std::string &s : v) s = BufferPool::mine().takeString(); return v; }
    void add(uint32_t key, const std::string &payload) {
        DbEntry e; e.key = key; e.offset = blockBytes + data.size(); e.length = payload.size() + 1;
        data.insert(data.end(), payload.begin(), payload.end());
        data.push_back('\0');
        entries.push_back(e);
    }
    // a block of consecutive entries formatted elsewhere (payloads concatenated, each followed by its NUL); the block is kept as it
    // is and written behind what came before it (no copy into one buffer)
    std::vector<std::string> blocks; uint64_t blockBytes = 0;
    const char *extIndex = nullptr; size_t extIndexBytes = 0, extIndexEntries = 0;      // (see close())
    void setExternalIndex(const char *text, size_t bytes, size_t nEntries) { extIndex = text; extIndexBytes = bytes; extIndexEntries = nEntries; }
    std::vector<std::pair<const char *, size_t>> blockView;      // per block: (nullptr, 0) = blocks[i] itself, else memory of the caller's that outlives close() (addExternal)
    // a block formatted ELSEWHERE and left where it is -- the device-side result text in the caller's page-locked buffer (sp_ctx_hits_text / sp_ctx_align_hits_text):
    // entry k has key keys[k] and lengths[k] bytes (its NUL included); `text` must stay valid until close() has returned
    void addExternal(const uint32_t *keys, const uint32_t *lengths, size_t n, const char *text, size_t bytes) {
        if (!data.empty()) { blocks.emplace_back(data.begin(), data.end()); blockView.emplace_back(nullptr, 0); blockBytes += data.size(); data.clear(); }
        uint64_t at = blockBytes;
        const size_t first = entries.size(); entries.resize(first + n);
        DbEntry *e = entries.data() + first;
        for (size_t i = 0; i < n; i++) { e[i].key = keys[i]; e[i].offset = at; e[i].length = lengths[i]; at += lengths[i]; }
        blockBytes += bytes;
        blocks.emplace_back(); blockView.emplace_back(text, bytes);
    }
    void addBlock(const std::vector<uint32_t> &keys, const std::vector<uint64_t> &lengths, std::string &&block) {
        if (!data.empty()) { blocks.emplace_back(data.begin(), data.end()); blockView.emplace_back(nullptr, 0); blockBytes += data.size(); data.clear(); }
        uint64_t at = blockBytes;
        const size_t first = entries.size(); entries.resize(first + keys.size());
        DbEntry *e = entries.data() + first;
        for (size_t i = 0; i < keys.size(); i++) { e[i].key = keys[i]; e[i].offset = at; e[i].length = lengths[i]; at += lengths[i]; }
        blockBytes += block.size();
        blocks.push_back(std::move(block)); blockView.emplace_back(nullptr, 0);
    }
    static bool writeAll(int fd, const char *p, size_t n) { while (n) { const ssize_t k = ::write(fd, p, n); if (k <= 0) return false; p += k; n -= (size_t) k; } return true; }
    static bool writeFile(const std::string &path, const char *p, size_t n) {
        const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) return false;
        const bool ok = writeAll(fd, p, n);
        return (::close(fd) == 0) && ok;
    }
    bool close(std::string &err) {
        const std::string dataPath = part.count > 1 ? path + "." + std::to_string(part.rank) : path;
        const std::string indexPath = part.count > 1 ? dataPath + ".idxpart" : path + ".index";
        const std::string typePath = part.count > 1 ? dataPath + ".typepart" : path + ".dbtype";
        if (part.count <= 1)      // split parts left by an earlier fan-out run would be read in front of / behind this data file
            for (int i = 0;; i++) { const std::string dp = path + "." + std::to_string(i); struct stat st; if (stat(dp.c_str(), &st) != 0) break; remove(dp.c_str()); }
        // the data file is written by a few threads (one block each, at its offset) WHILE the index text below is formatted and written: copying 10-20 MB into the
        // page cache from one thread was 7 ms of a served module call, the index another few
        const bool timed = getenv("SP_MODULE_TIMES") != nullptr;
        auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
        const double tw0 = timed ? now() : 0.0; double tw1 = 0, tw2 = 0, tw3 = 0;
        const int dataFd = ::open(dataPath.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        std::atomic<bool> dataOk{dataFd >= 0};
        std::vector<std::thread> dataWriters;
        if (dataFd >= 0) {
            // segments of at most 4 MB (a block formatted on the device is ONE piece of tens of megabytes), dealt to the writers
            struct Seg { const char *p; size_t n; uint64_t o; };
            std::vector<Seg> segs;
            uint64_t o = 0;
            for (size_t i = 0; i <= blocks.size(); i++) {
                const char *p = i < blocks.size() ? (blockView[i].first ? blockView[i].first : blocks[i].data()) : data.data();
                size_t n = i < blocks.size() ? (blockView[i].first ? blockView[i].second : blocks[i].size()) : data.size();
                while (n) { const size_t k = std::min<size_t>(n, (size_t) 4 << 20); segs.push_back({p, k, o}); p += k; n -= k; o += k; }
            }
            const size_t W = blockBytes + data.size() >= (4u << 20) ? std::min<size_t>(blockBytes + data.size() >= (64u << 20) ? 8 : 4, std::max<size_t>(1, segs.size())) : 1;      // (eight writers for a data file of 64 MB and more: the copy into the page cache is what takes the time)
            auto job = [dataFd, segs, W, &dataOk](size_t w) {
                for (size_t i = w; i < segs.size(); i += W) {
                    const char *p = segs[i].p; size_t n = segs[i].n; uint64_t o = segs[i].o;
                    while (n) { const ssize_t k = ::pwrite(dataFd, p, n, (off_t) o); if (k <= 0) { dataOk = false; return; } p += k; n -= (size_t) k; o += (uint64_t) k; }
                }
            };
            if (W == 1) job(0); else for (size_t w = 0; w < W; w++) dataWriters.emplace_back(job, w);
        }
        auto finishData = [&]() -> bool {
            for (std::thread &t : dataWriters) t.join();
            dataWriters.clear();
            bool ok = dataOk;
            if (dataFd >= 0) ok = (::close(dataFd) == 0) && ok;
            if (!ok) err = "cannot write " + dataPath;
            return ok;
        };
        if (timed) tw1 = now();
        bool sorted = true;
        for (size_t i = 1; i < entries.size() && sorted; i++) sorted = entries[i - 1].key <= entries[i].key;
        if (!sorted) std::sort(entries.begin(), entries.end(), [](const DbEntry &a, const DbEntry &b) { return a.key < b.key; });   // DBWriter.cpp:645-660
        // an index text that came with the external block (formatted on the device, sp_ctx_text_index) stands for the whole DB when that block is all there is and its
        // keys ascend -- the order the index is written in (a part of a fan-out run writes its own .idxpart the same way)
        const bool useExtIndex = extIndex && sorted && entries.size() == extIndexEntries && blocks.size() == 1 && blockView[0].first && data.empty();
        if (useExtIndex) {
            if (timed) tw2 = now();
            if (!writeFile(indexPath, extIndex, extIndexBytes)) { finishData(); err = "cannot write " + indexPath; return false; }
        } else {   // index text: formatted by a few threads over contiguous ranges, written with one call
            const size_t n = entries.size();
            const size_t T = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(32, (size_t) defaultThreads()), n / 16384 + 1));
            std::vector<std::string> texts = pooledBlocks(T);
            auto fmt = [&](size_t t) {
                // (the text as a LOCAL of the worker, moved into its slot at the end: neighbouring std::string headers share a cache line, and every append of one
                // thread would pull it out of its neighbour's cache -- the lesson of round 3's formatters, missed here: 3.9 ms for 6 x 10^5 lines on 16 threads)
                std::string text = std::move(texts[t]); const size_t from = n * t / T, to = n * (t + 1) / T;
                text.clear(); text.reserve((to - from) * 28);
                char buf[72];
                for (size_t i = from; i < to; i++) {
                    const DbEntry &e = entries[i];
                    char *q = u32toa(e.key, buf); *q++ = '\t';
                    q = u64toa(e.offset, q); *q++ = '\t';
                    q = u64toa(e.length, q); *q++ = '\n';
                    text.append(buf, (size_t) (q - buf));
                }
                texts[t] = std::move(text);
            };
            if (T == 1) fmt(0);
            else runTasks(T, fmt);
            if (timed) tw2 = now();
            bool wrote = false;
            {   // the parts one after the other, no concatenating copy
                const int fd = ::open(indexPath.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
                wrote = fd >= 0;
                for (const std::string &t : texts) wrote = wrote && writeAll(fd, t.data(), t.size());
                if (fd >= 0) wrote = (::close(fd) == 0) && wrote;
            }
            BufferPool &pool = BufferPool::mine();
            for (std::string &t : texts) pool.give(std::move(t));
            if (!wrote) { finishData(); err = "cannot write " + indexPath; return false; }
        }
        if (timed) tw3 = now();
        if (!finishData()) return false;
        if (timed) fprintf(stderr, "[sp]   write: data writers started %.3f, index formatted %.3f, index written %.3f, data written %.3f ms (%zu entries, %.1f MB)\n", tw1 - tw0, tw2 - tw1, tw3 - tw2, now() - tw3, entries.size(), (blockBytes + data.size()) / 1e6);
        // .dbtype is written LAST so that a crashed step is re-run by the workflow scripts (DBWriter.cpp:243)
        { const int32_t v = dbtype; if (!writeFile(typePath, (const char *) &v, 4)) { err = "cannot write " + typePath; return false; } }
        {   BufferPool &pool = BufferPool::mine();      // everything is on disk: the buffers go back to the pool with their capacity
            for (std::string &b : blocks) pool.give(std::move(b));
            blocks.clear(); blockView.clear();
            pool.give(std::move(entries)); }
        return true;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// flags (commons/Parameters.cpp:338-437): every flag of the module's list is accepted; values this implementation
// cannot honour are rejected loudly rather than ignored
// ---------------------------------------------------------------------------------------------------------------
struct Flags {
    std::vector<std::string> positional;
    std::map<std::string, std::string> kv;
    bool has(const std::string &k) const { return kv.count(k) != 0; }
    std::string str(const std::string &k, const std::string &d) const { auto it = kv.find(k); return it == kv.end() ? d : it->second; }
    long num(const std::string &k, long d) const { auto it = kv.find(k); return it == kv.end() ? d : strtol(it->second.c_str(), NULL, 10); }
    double real(const std::string &k, double d) const { auto it = kv.find(k); return it == kv.end() ? d : strtod(it->second.c_str(), NULL); }
};

static const char *kBoolFlags[] = {"-a", "--add-self-matches", "--realign", "--wrapped-scoring", "--filter-hits", "--simple-best-hit", "--translate", "--use-all-table-starts",
                                   "--create-lookup", "--add-orf-stop", "--report-fdr", "--db-output", NULL};

inline bool parseFlags(int argc, const char **argv, const std::set<std::string> &allowed, size_t nPositional, Flags &f, std::string &err) {
    for (int i = 0; i < argc; i++) {
        std::string a = argv[i];
        if (a.size() >= 2 && a[0] == '-' && !(a.size() > 1 && (isdigit((unsigned char) a[1])))) {
            if (!allowed.count(a)) { err = "Unrecognized parameter \"" + a + "\""; return false; }      // fatal in the reference too (Parameters.cpp:1663-1666)
            bool isBool = false;
            for (const char **b = kBoolFlags; *b; b++) if (a == *b) isBool = true;
            if (isBool) {
                if (i + 1 < argc && (!strcmp(argv[i + 1], "0") || !strcmp(argv[i + 1], "1") || !strcmp(argv[i + 1], "true") || !strcmp(argv[i + 1], "false"))) {
                    std::string v = argv[++i]; f.kv[a] = (v == "1" || v == "true") ? "1" : "0";
                } else f.kv[a] = "1";
            } else {
                if (i + 1 >= argc) { err = "Missing argument for " + a; return false; }
                f.kv[a] = argv[++i];
            }
        } else f.positional.push_back(a);
    }
    if (f.positional.size() != nPositional) { err = "expected " + std::to_string(nPositional) + " database arguments, got " + std::to_string(f.positional.size()); return false; }
    return true;
}

// MultiParam "nucl:X,aa:Y" (commons/MultiParam.cpp:22-95): the amino-acid part
inline std::string aaPart(const std::string &v) {
    size_t p = v.find("aa:");
    if (p == std::string::npos) return v;
    size_t e = v.find(',', p);
    return v.substr(p + 3, e == std::string::npos ? std::string::npos : e - p - 3);
}

inline bool requireValue(const Flags &f, const char *flag, const std::vector<std::string> &ok, std::string &err) {
    if (!f.has(flag)) return true;
    std::string v = aaPart(f.str(flag, ""));
    for (const std::string &o : ok) if (v == o) return true;
    err = std::string("unsupported value for ") + flag + ": " + v + " (the MI355X path implements SpacePHARER's search configuration only)";
    return false;
}

// decimal text, two digits per division (result lines and index lines are three to eleven integers each, 10^6 of them per module call)
static const char kDigits2[201] = "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
inline char *u32toa(uint32_t v, char *p) {
    char t[12]; int n = 12;
    while (v >= 100u) { const uint32_t r = v % 100u; v /= 100u; n -= 2; memcpy(t + n, kDigits2 + 2 * r, 2); }
    if (v >= 10u) { n -= 2; memcpy(t + n, kDigits2 + 2 * v, 2); } else t[--n] = (char) ('0' + v);
    memcpy(p, t + n, (size_t) (12 - n));
    return p + (12 - n);
}
inline char *u64toa(uint64_t v, char *p) {
    if (v <= 0xFFFFFFFFull) return u32toa((uint32_t) v, p);
    char t[20]; int n = 20;
    while (v >= 100u) { const uint64_t r = v % 100u; v /= 100u; n -= 2; memcpy(t + n, kDigits2 + 2 * r, 2); }
    if (v >= 10u) { n -= 2; memcpy(t + n, kDigits2 + 2 * v, 2); } else t[--n] = (char) ('0' + v);
    memcpy(p, t + n, (size_t) (20 - n));
    return p + (20 - n);
}
inline char *i32toa(int32_t v, char *p) { if (v < 0) { *p++ = '-'; return u32toa((uint32_t) (-(int64_t) v), p); } return u32toa((uint32_t) v, p); }

// DB key -> position: keys are usually dense (0 .. n-1), so a flat table; a sorted list otherwise
struct KeyMap {
    std::vector<uint32_t> lut; std::vector<std::pair<uint32_t, uint32_t>> sorted; bool dense = false;
    void build(const std::vector<uint32_t> &keys) {
        uint32_t mx = 0; for (uint32_t k : keys) mx = std::max(mx, k);
        dense = !keys.empty() && (uint64_t) mx < 4ull * keys.size() + 1024;
        if (dense) { lut.assign((size_t) mx + 1, 0xFFFFFFFFu); for (size_t i = 0; i < keys.size(); i++) lut[keys[i]] = (uint32_t) i; }
        else { sorted.resize(keys.size()); for (size_t i = 0; i < keys.size(); i++) sorted[i] = {keys[i], (uint32_t) i}; std::sort(sorted.begin(), sorted.end()); }
    }
    bool find(uint32_t key, uint32_t &pos) const {
        if (dense) { if (key >= lut.size() || lut[key] == 0xFFFFFFFFu) return false; pos = lut[key]; return true; }
        auto it = std::lower_bound(sorted.begin(), sorted.end(), std::make_pair(key, 0u));
        if (it == sorted.end() || it->first != key) return false;
        pos = it->second; return true;
    }
};

template <typename F>
inline void parallelChunks(size_t n, int threads, F f) {      // f(chunkIndex, from, to), chunks in order
    const size_t T = (size_t) std::max(1, std::min<int>(threads, (int) std::min<size_t>(n / 2048 + 1, 64)));
    if (T == 1) { f(0, 0, n); return; }
    const size_t per = (n + T - 1) / T;
    runTasks(T, [&](size_t t) { f(t, std::min(n, t * per), std::min(n, (t + 1) * per)); });
}
inline size_t numChunks(size_t n, int threads) { return (size_t) std::max(1, std::min<int>(threads, (int) std::min<size_t>(n / 2048 + 1, 64))); }

inline int fail(const std::string &e) { setLastError(e); fprintf(stderr, "%s\n", e.c_str()); return 1; }

// "%.3E" of a double (Matcher::resultToBuffer prints E-values that way): std::to_chars with a precision is defined to give exactly
// what printf gives, several times faster -- the formatting of 10^6 alignment lines is a visible part of a module run
inline char *fmtE3(double x, char *p) {
    if (!std::isfinite(x)) return p + sprintf(p, "%.3E", x);
    const std::to_chars_result r = std::to_chars(p, p + 32, x, std::chars_format::scientific, 3);
    for (char *c = p; c < r.ptr; c++) if (*c == 'e') { *c = 'E'; break; }
    return r.ptr;
}

// "%.3E" through a small cache keyed by the double's bits: E-values are a function of (score, query length) and repeat by the thousand in a result DB
inline char *fmtE3Cached(double x, char *p) {
    struct Slot { uint64_t bits; uint8_t len; char text[15]; };
    static thread_local Slot cache[1024] = {};
    uint64_t bits; memcpy(&bits, &x, 8);
    Slot &s = cache[(bits * 0x9E3779B97F4A7C15ull) >> 54];
    if (s.len == 0 || s.bits != bits) {
        char tmp[40]; char *e = fmtE3(x, tmp);
        if ((size_t) (e - tmp) > sizeof(s.text)) { memcpy(p, tmp, (size_t) (e - tmp)); return p + (e - tmp); }      // (inf / nan / exponents of three digits: not cached)
        s.bits = bits; s.len = (uint8_t) (e - tmp); memcpy(s.text, tmp, s.len);
    }
    memcpy(p, s.text, s.len);
    return p + s.len;
}

inline void appendAlnLine(std::string &out, const sp_alignment &r, const char *cigar, bool addBacktrace) {
    // the whole line in a local buffer, one append (flushed in between only for a backtrace of hundreds of runs)
    char buf[512]; char *p = buf;
    p = u32toa(r.db_key, p); *p++ = '\t';
    p = i32toa(r.bit_score, p); *p++ = '\t';
    // Util::fastSeqIdToBuffer (Util.cpp:278-306): for 1.0 the caller overwrites the third '0' with the tab
    if (r.seq_id == 1.0) { memcpy(p, "1.00", 4); p += 4; }
    else { *p++ = '0'; *p++ = '.'; if (r.seq_id < 0.10) *p++ = '0'; if (r.seq_id < 0.01) *p++ = '0'; p = i32toa((int) (r.seq_id * 1000), p); }
    *p++ = '\t';
    p = fmtE3Cached(r.evalue, p); *p++ = '\t';
    p = i32toa(r.q_start, p); *p++ = '\t'; p = i32toa(r.q_end, p); *p++ = '\t'; p = i32toa((int) r.q_len, p); *p++ = '\t';
    p = i32toa(r.db_start, p); *p++ = '\t'; p = i32toa(r.db_end, p); *p++ = '\t'; p = i32toa((int) r.db_len, p);
    if (addBacktrace) {          // Matcher::compressAlignment (Matcher.cpp:211-230)
        *p++ = '\t';
        char state = 'M'; uint32_t counter = 0;
        for (const char *c = cigar; *c; c++) {
            if (*c != state) {
                p = u32toa(counter, p); *p++ = state; state = *c; counter = 1;
                if (p > buf + sizeof(buf) - 32) { out.append(buf, (size_t) (p - buf)); p = buf; }
            } else counter++;
        }
        p = u32toa(counter, p); *p++ = state;
    }
    *p++ = '\n';
    out.append(buf, (size_t) (p - buf));
}

inline bool alnLess(const sp_alignment &a, const sp_alignment &b) {      // Matcher::compareHits, Matcher.h:156-167
    if (a.evalue != b.evalue) return a.evalue < b.evalue;
    if (a.bit_score != b.bit_score) return a.bit_score > b.bit_score;
    if (a.db_len != b.db_len) return a.db_len < b.db_len;
    return a.db_key < b.db_key;
}


}  // namespace sp
