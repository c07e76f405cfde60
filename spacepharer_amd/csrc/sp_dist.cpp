// Multi-GPU exchange (see sp_dist.h).  The reference spreads a search over MPI ranks that each write files and lets rank 0 merge them
// (lib/mmseqs/src/prefiltering/Prefiltering.cpp:258-362, commons/DBReader.cpp:1171-1213); here the ranks are the GPUs of one node and what they exchange -- per-shard
// candidate lists, per-slice hit lists -- stays in HBM: RCCL all-gather over xGMI.  RCCL is loaded when the first communicator with that transport is made
// (dlopen: a one-GPU module run never pays for it).
#include "sp_dist.h"

#include <arpa/inet.h>
#include <dlfcn.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <vector>
#include <thread>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

namespace sp {

namespace {

struct IdLayout { char magic[8]; uint16_t port; uint16_t pad; uint32_t pad2; uint64_t token; char addr[64]; };
static_assert(sizeof(IdLayout) <= kDistIdBytes, "id layout");

bool sendAll(int fd, const void *p, size_t n) {
    const char *c = static_cast<const char *>(p);
    while (n) { const ssize_t w = ::send(fd, c, n, MSG_NOSIGNAL); if (w <= 0) { if (w < 0 && errno == EINTR) continue; return false; } c += w; n -= (size_t) w; }
    return true;
}
bool recvAll(int fd, void *p, size_t n) {
    char *c = static_cast<char *>(p);
    while (n) { const ssize_t r = ::recv(fd, c, n, 0); if (r <= 0) { if (r < 0 && errno == EINTR) continue; return false; } c += r; n -= (size_t) r; }
    return true;
}
void noDelay(int fd) { int one = 1; (void) setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one)); }

// the RCCL entry points used, resolved from librccl.so
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*getUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*commInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*commDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*allGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*errorString)(ncclResult_t) = nullptr;
    ncclResult_t (*send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;      // (point to point: the all-to-all of the probe-list cut; optional)
    ncclResult_t (*recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*groupStart)() = nullptr;
    ncclResult_t (*groupEnd)() = nullptr;
    bool load(std::string &err) {
        if (lib) return true;
        // the RCCL that belongs to the HIP runtime this library is linked against -- by PATH: a process may already hold another copy under the same soname (PyTorch
        // bundles its own, bound to its own runtime: device pointers of this library's runtime mean nothing to it)
        std::vector<std::string> names;
        if (const char *e = getenv("SP_RCCL_LIB")) names.push_back(e);
        if (const char *r = getenv("ROCM_PATH")) names.push_back(std::string(r) + "/lib/librccl.so.1");
        names.push_back("/opt/rocm/lib/librccl.so.1"); names.push_back("librccl.so.1"); names.push_back("librccl.so");
        for (const std::string &n : names) { lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) { err = std::string("RCCL is not available: ") + dlerror(); return false; }
        getUniqueId = reinterpret_cast<decltype(getUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        commInitRank = reinterpret_cast<decltype(commInitRank)>(dlsym(lib, "ncclCommInitRank"));
        commDestroy = reinterpret_cast<decltype(commDestroy)>(dlsym(lib, "ncclCommDestroy"));
        allGather = reinterpret_cast<decltype(allGather)>(dlsym(lib, "ncclAllGather"));
        errorString = reinterpret_cast<decltype(errorString)>(dlsym(lib, "ncclGetErrorString"));
        send = reinterpret_cast<decltype(send)>(dlsym(lib, "ncclSend")); recv = reinterpret_cast<decltype(recv)>(dlsym(lib, "ncclRecv"));
        groupStart = reinterpret_cast<decltype(groupStart)>(dlsym(lib, "ncclGroupStart")); groupEnd = reinterpret_cast<decltype(groupEnd)>(dlsym(lib, "ncclGroupEnd"));
        if (!getUniqueId || !commInitRank || !commDestroy || !allGather || !errorString) { err = "librccl.so lacks an entry point (ncclGetUniqueId / ncclCommInitRank / ncclAllGather ...)"; return false; }
        return true;
    }
};
Rccl g_rccl;

// sockets bound by uniqueId() and not yet taken by a rank 0 of this process (a handful at most: an id nobody uses keeps its port until the process ends or eight newer ones exist)
std::mutex g_probeMu;
std::map<uint64_t, int> g_probes;
std::vector<uint64_t> g_probeOrder;
void keepProbe(uint64_t token, int fd) {
    std::lock_guard<std::mutex> lk(g_probeMu);
    g_probes[token] = fd; g_probeOrder.push_back(token);
    while (g_probeOrder.size() > 8) { auto it = g_probes.find(g_probeOrder.front()); if (it != g_probes.end()) { close(it->second); g_probes.erase(it); } g_probeOrder.erase(g_probeOrder.begin()); }
}
int takeProbe(uint64_t token) {
    std::lock_guard<std::mutex> lk(g_probeMu);
    auto it = g_probes.find(token);
    if (it == g_probes.end()) return -1;
    const int fd = it->second; g_probes.erase(it);
    for (size_t k = 0; k < g_probeOrder.size(); k++) if (g_probeOrder[k] == token) { g_probeOrder.erase(g_probeOrder.begin() + (long) k); break; }
    return fd;
}
// wait until fd is readable or the deadline has passed
bool readableBefore(int fd, std::chrono::steady_clock::time_point deadline) {
    for (;;) {
        const auto left = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - std::chrono::steady_clock::now()).count();
        if (left <= 0) return false;
        pollfd p; p.fd = fd; p.events = POLLIN; p.revents = 0;
        const int r = poll(&p, 1, (int) (left > 1000 ? 1000 : left));
        if (r > 0) return true;
        if (r < 0 && errno != EINTR) return false;
    }
}

}  // namespace

bool Comm::uniqueId(unsigned char *id, std::string &err) {
    memset(id, 0, kDistIdBytes);
    IdLayout l; memset(&l, 0, sizeof(l));
    memcpy(l.magic, "SPDIST1", 8);
    const char *addr = getenv("SP_DIST_ADDR");
    snprintf(l.addr, sizeof(l.addr), "%s", addr && *addr ? addr : "127.0.0.1");
    // a free port ON THE ID'S ADDRESS (loopback unless SP_DIST_ADDR says otherwise -- a one-node job has no business listening on every interface): bind to port 0,
    // read it back, and KEEP the socket: the process that made the id is the one that becomes rank 0 in every caller this library has, and Comm::create() listens on
    // this very socket (no window in which somebody else can take the port).  An id that travels to another process' rank 0 is bound again there (SO_REUSEADDR).
    const int fd = socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) { err = "sp_dist_unique_id: socket() failed"; return false; }
    int one = 1; (void) setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in sa; memset(&sa, 0, sizeof(sa)); sa.sin_family = AF_INET; sa.sin_port = 0;
    if (inet_pton(AF_INET, l.addr, &sa.sin_addr) != 1) { close(fd); err = std::string("sp_dist_unique_id: bad address ") + l.addr; return false; }
    socklen_t len = sizeof(sa);
    if (bind(fd, reinterpret_cast<sockaddr *>(&sa), sizeof(sa)) != 0 || getsockname(fd, reinterpret_cast<sockaddr *>(&sa), &len) != 0) { close(fd); err = std::string("sp_dist_unique_id: no free port on ") + l.addr; return false; }
    l.port = ntohs(sa.sin_port);
    std::random_device rd; l.token = ((uint64_t) rd() << 32) ^ rd() ^ (uint64_t) getpid();
    keepProbe(l.token, fd);
    memcpy(id, &l, sizeof(l));
    return true;
}

Comm *Comm::create(const unsigned char *id, int rank, int world, int transport, int timeoutSec, std::string &err) {
    IdLayout l; memcpy(&l, id, sizeof(l));
    if (memcmp(l.magic, "SPDIST1", 8) != 0) { err = "sp_dist_init: not an id made by sp_dist_unique_id"; return nullptr; }
    if (world < 1 || rank < 0 || rank >= world) { err = "sp_dist_init: rank " + std::to_string(rank) + " of " + std::to_string(world); return nullptr; }
    if (transport != kTransportRccl && transport != kTransportHost) { err = "sp_dist_init: unknown transport"; return nullptr; }
    Comm *c = new Comm; c->rank_ = rank; c->world_ = world; c->transport_ = transport;
    auto failed = [&](const std::string &why) { err = why; delete c; return (Comm *) nullptr; };
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(timeoutSec > 0 ? timeoutSec : 120);
    if (world > 1) {
        if (rank == 0) {
            // the socket uniqueId() bound (same process), or a fresh one on the id's address and port
            c->listenFd_ = takeProbe(l.token);
            if (c->listenFd_ < 0) {
                c->listenFd_ = socket(AF_INET, SOCK_STREAM, 0);
                int one = 1; (void) setsockopt(c->listenFd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
                sockaddr_in sa; memset(&sa, 0, sizeof(sa)); sa.sin_family = AF_INET; sa.sin_port = htons(l.port);
                if (inet_pton(AF_INET, l.addr, &sa.sin_addr) != 1) return failed(std::string("sp_dist_init: bad address ") + l.addr);
                if (c->listenFd_ < 0 || bind(c->listenFd_, reinterpret_cast<sockaddr *>(&sa), sizeof(sa)) != 0)
                    return failed("sp_dist_init: rank 0 cannot bind " + std::string(l.addr) + ":" + std::to_string(l.port) + ": " + strerror(errno));
            }
            if (listen(c->listenFd_, world) != 0) return failed("sp_dist_init: rank 0 cannot listen on " + std::string(l.addr) + ":" + std::to_string(l.port) + ": " + strerror(errno));
            c->fds_.assign((size_t) world, -1);
            for (int k = 1; k < world; k++) {
                // the deadline bounds accept() and the hello only (poll): the accepted sockets carry NO receive timeout, so a peer that reaches a collective minutes
                // after rank 0 -- a target shard still being read from a cold disk -- is waited for, as on every other rank
                if (!readableBefore(c->listenFd_, deadline)) return failed("sp_dist_init: rank 0 waited in vain for " + std::to_string(world - k) + " of " + std::to_string(world - 1) + " peers");
                const int fd = accept(c->listenFd_, nullptr, nullptr);
                if (fd < 0) { k--; continue; }
                noDelay(fd);
                uint64_t hello[2] = {0, 0};
                { timeval tv; tv.tv_sec = 5; tv.tv_usec = 0; (void) setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv)); }      // (the 16 bytes of the hello only ...
                const bool greeted = readableBefore(fd, deadline) && recvAll(fd, hello, sizeof(hello));
                { timeval tv; tv.tv_sec = 0; tv.tv_usec = 0; (void) setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv)); }      // ... the data path waits without a limit)
                if (!greeted || hello[0] != l.token || hello[1] == 0 || hello[1] >= (uint64_t) world || c->fds_[hello[1]] >= 0) { close(fd); k--; continue; }      // (not one of ours)
                c->fds_[hello[1]] = fd;
            }
        } else {
            sockaddr_in sa; memset(&sa, 0, sizeof(sa)); sa.sin_family = AF_INET; sa.sin_port = htons(l.port);
            if (inet_pton(AF_INET, l.addr, &sa.sin_addr) != 1) return failed(std::string("sp_dist_init: bad address ") + l.addr);
            int fd = -1;
            for (;;) {
                fd = socket(AF_INET, SOCK_STREAM, 0);
                if (fd >= 0 && connect(fd, reinterpret_cast<sockaddr *>(&sa), sizeof(sa)) == 0) break;
                if (fd >= 0) close(fd);
                fd = -1;
                if (std::chrono::steady_clock::now() > deadline) return failed("sp_dist_init: rank " + std::to_string(rank) + " cannot reach rank 0 at " + l.addr + ":" + std::to_string(l.port));
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
            noDelay(fd);
            const uint64_t hello[2] = {l.token, (uint64_t) rank};
            if (!sendAll(fd, hello, sizeof(hello))) { close(fd); return failed("sp_dist_init: handshake with rank 0 failed"); }
            c->fds_.assign(1, fd);
        }
    }
    if (!c->barrier(err)) { delete c; return nullptr; }
    if (transport == kTransportRccl) {
        // Every step of the RCCL set-up may fail on ONE rank only (a library that does not load, a device another rank already holds, a fabric that is not there).  The
        // ranks therefore agree on the outcome over the bootstrap sockets: all of them hold an RCCL communicator, or all of them fall back to the host transport -- a rank
        // that returned an error here while the others went on would leave them waiting in their first collective.  SP_DIST_STRICT=1: no fallback, the error comes back.
        std::string why;
        bool ok = g_rccl.load(why);
        ncclUniqueId nid; memset(&nid, 0, sizeof(nid));
        if (ok && rank == 0) { const ncclResult_t r = g_rccl.getUniqueId(&nid); if (r != ncclSuccess) { ok = false; why = std::string("ncclGetUniqueId: ") + g_rccl.errorString(r); } }
        std::vector<unsigned char> all((size_t) world * (sizeof(nid) + 1)), mine(sizeof(nid) + 1);
        memcpy(mine.data(), &nid, sizeof(nid)); mine[sizeof(nid)] = ok ? 1 : 0;
        if (!c->hostAllgather(mine.data(), mine.size(), all.data(), err)) { delete c; return nullptr; }
        bool allOk = true; for (int k = 0; k < world; k++) allOk = allOk && all[(size_t) k * mine.size() + sizeof(nid)] != 0;
        ncclComm_t comm = nullptr; hipStream_t s = nullptr;
        if (allOk) {
            memcpy(&nid, all.data(), sizeof(nid));      // rank 0's
            const ncclResult_t r = g_rccl.commInitRank(&comm, world, nid, rank);
            if (r != ncclSuccess) { ok = false; comm = nullptr; why = std::string("ncclCommInitRank: ") + g_rccl.errorString(r) + " (one rank per GPU; ranks that share a device need the host transport)"; }
            else if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { ok = false; why = "sp_dist_init: cannot create the exchange stream"; }
            std::vector<unsigned char> flags((size_t) world); const unsigned char f = ok ? 1 : 0;
            if (!c->hostAllgather(&f, 1, flags.data(), err)) { if (comm) (void) g_rccl.commDestroy(comm); if (s) (void) hipStreamDestroy(s); delete c; return nullptr; }
            for (int k = 0; k < world; k++) allOk = allOk && flags[(size_t) k] != 0;
        }
        if (allOk) { c->nccl_ = comm; c->stream_ = s; }
        else {
            if (comm) (void) g_rccl.commDestroy(comm);
            if (s) (void) hipStreamDestroy(s);
            if (why.empty()) why = "another rank could not set up RCCL";
            const char *strict = getenv("SP_DIST_STRICT");
            if (strict && *strict && *strict != '0') return failed(why);
            fprintf(stderr, "[sp] rank %d: RCCL is not available to this job (%s) -- the exchange goes through host memory and the bootstrap sockets instead\n", rank, why.c_str());
            c->transport_ = kTransportHost;
        }
    }
    return c;
}

Comm::~Comm() {
    if (nccl_ && g_rccl.commDestroy) (void) g_rccl.commDestroy(static_cast<ncclComm_t>(nccl_));
    if (stream_) (void) hipStreamDestroy(static_cast<hipStream_t>(stream_));
    for (int fd : fds_) if (fd >= 0) close(fd);
    if (listenFd_ >= 0) close(listenFd_);
}

// star over the bootstrap sockets: everybody sends to rank 0, rank 0 sends the concatenation to everybody
bool Comm::hostAllgather(const void *send, size_t bytes, void *recv, std::string &err) {
    if (world_ == 1) { if (recv != send) memmove(recv, send, bytes); return true; }
    unsigned char *out = static_cast<unsigned char *>(recv);
    if (rank_ == 0) {
        memmove(out, send, bytes);
        for (int k = 1; k < world_; k++) if (!recvAll(fds_[(size_t) k], out + (size_t) k * bytes, bytes)) { err = "sp_dist: rank " + std::to_string(k) + " is gone"; return false; }
        for (int k = 1; k < world_; k++) if (!sendAll(fds_[(size_t) k], out, (size_t) world_ * bytes)) { err = "sp_dist: rank " + std::to_string(k) + " is gone"; return false; }
    } else {
        if (!sendAll(fds_[0], send, bytes) || !recvAll(fds_[0], out, (size_t) world_ * bytes)) { err = "sp_dist: rank 0 is gone"; return false; }
    }
    bytes_ += (uint64_t) world_ * bytes;
    return true;
}

bool Comm::allgather(const void *send, size_t bytes, void *recv, bool onDevice, std::string &err) {
    if (bytes == 0) return true;
    if (!onDevice) return hostAllgather(send, bytes, recv, err);
    if (transport_ == kTransportRccl) {
        const ncclResult_t r = g_rccl.allGather(send, recv, bytes, ncclChar, static_cast<ncclComm_t>(nccl_), static_cast<hipStream_t>(stream_));
        if (r != ncclSuccess) { err = std::string("ncclAllGather: ") + g_rccl.errorString(r); return false; }
        if (hipStreamSynchronize(static_cast<hipStream_t>(stream_)) != hipSuccess) { err = "sp_dist: the exchange stream failed"; return false; }
        bytes_ += (uint64_t) world_ * bytes;
        return true;
    }
    // ranks that share a GPU: through host memory and the sockets
    stage_.resize((size_t) (world_ + 1) * bytes);
    if (hipMemcpy(stage_.data(), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) { err = "sp_dist: device -> host copy failed"; return false; }
    if (!hostAllgather(stage_.data(), bytes, stage_.data() + bytes, err)) return false;
    if (hipMemcpy(recv, stage_.data() + bytes, (size_t) world_ * bytes, hipMemcpyHostToDevice) != hipSuccess) { err = "sp_dist: host -> device copy failed"; return false; }
    return true;
}

// Every rank sends a block of its own size to every rank (sendBytes[r] to rank r, recvBytes[r] from rank r; the sizes were agreed on by the caller).  Device buffers over
// RCCL: one group of point-to-point sends and receives, HBM to HBM over the xGMI links of the pairs themselves.  Otherwise through rank 0 over the bootstrap sockets.
bool Comm::alltoallv(const void *const *send, const size_t *sendBytes, void *const *recv, const size_t *recvBytes, bool onDevice, std::string &err) {
    const size_t me = (size_t) rank_, W = (size_t) world_;
    if (sendBytes[me] != recvBytes[me]) { err = "sp_dist alltoallv: a rank's block to itself has two sizes"; return false; }
    auto selfCopy = [&]() -> bool {
        if (!sendBytes[me] || send[me] == recv[me]) return true;
        if (onDevice) return hipMemcpy(recv[me], send[me], sendBytes[me], hipMemcpyDeviceToDevice) == hipSuccess;
        memmove(recv[me], send[me], sendBytes[me]); return true;
    };
    if (world_ == 1) { if (!selfCopy()) { err = "sp_dist alltoallv: copy failed"; return false; } return true; }
    // The sizes first, over the bootstrap sockets, on EVERY transport: each rank publishes what it sends and what it expects, every rank checks the whole matrix and so reaches the
    // same verdict -- a mismatch fails all ranks together here instead of leaving the others in a receive without a time-out (or inside ncclGroupEnd).
    std::vector<uint64_t> mineSz(2 * W), both(2 * W * W), mat(W * W);
    for (size_t k = 0; k < W; k++) { mineSz[k] = sendBytes[k]; mineSz[W + k] = recvBytes[k]; }
    if (!hostAllgather(mineSz.data(), 2 * W * sizeof(uint64_t), both.data(), err)) return false;
    for (size_t src = 0; src < W; src++) for (size_t dst = 0; dst < W; dst++) {
        mat[src * W + dst] = both[src * 2 * W + dst];
        if (both[src * 2 * W + dst] != both[dst * 2 * W + W + src]) {
            err = "sp_dist alltoallv: rank " + std::to_string(src) + " sends " + std::to_string(both[src * 2 * W + dst]) + " bytes to rank " + std::to_string(dst) + ", which expects " + std::to_string(both[dst * 2 * W + W + src]);
            return false;
        }
    }
    if (onDevice && transport_ == kTransportRccl && g_rccl.send && g_rccl.recv && g_rccl.groupStart && g_rccl.groupEnd) {
        ncclComm_t nc = static_cast<ncclComm_t>(nccl_); hipStream_t st = static_cast<hipStream_t>(stream_);
        ncclResult_t r = g_rccl.groupStart();
        for (size_t k = 0; k < W && r == ncclSuccess; k++) {
            if (k == me) continue;
            if (sendBytes[k]) r = g_rccl.send(send[k], sendBytes[k], ncclChar, (int) k, nc, st);
            if (r == ncclSuccess && recvBytes[k]) r = g_rccl.recv(recv[k], recvBytes[k], ncclChar, (int) k, nc, st);
        }
        const ncclResult_t e = g_rccl.groupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) { err = std::string("ncclSend/ncclRecv: ") + g_rccl.errorString(r); return false; }
        if (!selfCopy() || hipStreamSynchronize(st) != hipSuccess) { err = "sp_dist: the exchange stream failed"; return false; }
        for (size_t k = 0; k < W; k++) if (k != me) bytes_ += recvBytes[k];
        return true;
    }
    // through rank 0: everybody hands it all its blocks (sizes first), it hands everybody the blocks addressed to them, in the order of the senders
    size_t sendTotal = 0, recvTotal = 0;
    for (size_t k = 0; k < W; k++) { sendTotal += sendBytes[k]; recvTotal += recvBytes[k]; }
    std::vector<unsigned char> out(sendTotal + 1), in(recvTotal + 1);
    {   size_t at = 0;
        for (size_t k = 0; k < W; k++) {
            if (sendBytes[k]) { if (onDevice) { if (hipMemcpy(out.data() + at, send[k], sendBytes[k], hipMemcpyDeviceToHost) != hipSuccess) { err = "sp_dist: device -> host copy failed"; return false; } } else memcpy(out.data() + at, send[k], sendBytes[k]); }
            at += sendBytes[k];
        } }
    if (rank_ == 0) {
        std::vector<std::vector<unsigned char>> from(W);      // rank k's blocks back to back
        for (size_t k = 1; k < W; k++) {
            size_t tot = 0; for (size_t j = 0; j < W; j++) tot += (size_t) mat[k * W + j];
            from[k].resize(tot + 1);
            if (tot && !recvAll(fds_[k], from[k].data(), tot)) { err = "sp_dist: rank " + std::to_string(k) + " is gone"; return false; }
        }
        auto blockOf = [&](size_t src, size_t dst) -> const unsigned char * {
            size_t at = 0; for (size_t j = 0; j < dst; j++) at += (size_t) mat[src * W + j];
            return (src == 0 ? out.data() : from[src].data()) + at;
        };
        for (size_t dst = 0; dst < W; dst++) {
            std::vector<unsigned char> cat; size_t tot = 0;
            for (size_t src = 0; src < W; src++) tot += (size_t) mat[src * W + dst];
            cat.reserve(tot + 1);
            for (size_t src = 0; src < W; src++) cat.insert(cat.end(), blockOf(src, dst), blockOf(src, dst) + (size_t) mat[src * W + dst]);
            if (dst == 0) memcpy(in.data(), cat.data(), tot);
            else if (tot && !sendAll(fds_[dst], cat.data(), tot)) { err = "sp_dist: rank " + std::to_string(dst) + " is gone"; return false; }
        }
    } else {
        if ((sendTotal && !sendAll(fds_[0], out.data(), sendTotal)) || (recvTotal && !recvAll(fds_[0], in.data(), recvTotal))) { err = "sp_dist: rank 0 is gone"; return false; }
    }
    {   size_t at = 0;
        for (size_t k = 0; k < W; k++) {
            if (recvBytes[k]) { if (onDevice) { if (hipMemcpy(recv[k], in.data() + at, recvBytes[k], hipMemcpyHostToDevice) != hipSuccess) { err = "sp_dist: host -> device copy failed"; return false; } } else memcpy(recv[k], in.data() + at, recvBytes[k]); }
            at += recvBytes[k];
        } }
    bytes_ += recvTotal;
    return true;
}

bool Comm::barrier(std::string &err) {
    unsigned char one = 1; std::vector<unsigned char> all((size_t) world_);
    return hostAllgather(&one, 1, all.data(), err);
}

bool Comm::allreduce(double *v, size_t n, int op, std::string &err) {
    if (world_ == 1 || n == 0) return true;
    std::vector<double> all((size_t) world_ * n);
    if (!hostAllgather(v, n * sizeof(double), all.data(), err)) return false;
    for (size_t i = 0; i < n; i++) {
        double acc = all[i];
        for (int k = 1; k < world_; k++) { const double x = all[(size_t) k * n + i]; acc = op == 0 ? acc + x : (op == 1 ? (x > acc ? x : acc) : (x < acc ? x : acc)); }
        v[i] = acc;
    }
    return true;
}

bool decomposeByResidues(const uint64_t *lengths, size_t n, int rank, int world, size_t &start, size_t &count) {
    uint64_t dataSize = 0;
    for (size_t i = 0; i < n; i++) dataSize += lengths[i];
    if ((uint64_t) world > (dataSize > 1 ? dataSize : 1)) return false;
    if (world == 1) { start = 0; count = n; return true; }
    if (n <= (size_t) world) { start = (size_t) rank < n ? (size_t) rank : 0; count = (size_t) rank < n ? 1 : 0; return true; }
    const uint64_t chunk = (uint64_t) std::ceil((double) dataSize / (double) world);
    std::vector<size_t> per((size_t) world, 0);
    size_t cur = 0; uint64_t acc = 0;
    for (size_t i = 0; i < n; i++) {
        if (acc >= chunk) { acc = 0; cur++; }
        acc += lengths[i];
        per[cur] += 1;
    }
    start = 0; count = per[(size_t) rank];
    for (int j = 0; j < rank; j++) start += per[(size_t) j];
    return true;
}

}  // namespace sp
