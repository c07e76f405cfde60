// Front-end binary: `spacepharer-amd <module> <args...>` with module in {prefilter, align, rescorediagonal, search} and the set-level
// steps around them (extractorfs ... summarizeresults).
// Point $MMSEQS at a wrapper that sends these module names here and every other one to the stock spacepharer binary, and the
// unmodified workflow scripts (lib/mmseqs/data/workflow/blastp.sh:33,41,48) land in the MI355X implementation.
//
// `spacepharer-amd serve <socket>` keeps ONE process with the HIP runtime up and the last target sets resident in HBM and runs
// module calls for clients: with SP_SERVER=<socket> in the environment `spacepharer-amd <module> <args...>` forwards its argv
// (and working directory) there and exits with the module's status.  A module call then costs what it costs in a warm process
// -- a fresh process pays ~0.12 s of runtime start-up and tables before its first kernel, more than the whole search of 10^5 spacers
// takes.  Without a reachable server the call runs in this process as before.  SP_SERVER=auto[:<idle seconds>] uses a per-user
// server on a default socket and STARTS it when nobody answers (it leaves by itself after 120 s without a call): the workflow
// scripts' back-to-back module processes (blastp.sh:33,41 for targets and again for the control set) then pay the start-up once,
// and `align` finds the hits its `prefilter` left resident (sp_modules.cpp: PrefilterMemo).
#include <cerrno>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <exception>
#include <fcntl.h>
#include <malloc.h>
#include <poll.h>
#include <sys/file.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/un.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include <vector>

#include <dlfcn.h>

// The front end does NOT link the library: a client that forwards its call to the resident server is a few system calls, and a process that links
// libspacepharer_amd.so maps the HIP runtime and its dependencies before main() -- 10-25 ms, a third of a served module call at 10^5 spacers.
// The library (next to this binary) is loaded when a module runs in THIS process.
static void *g_lib = nullptr;
static void *libSym(const char *name) {
    if (!g_lib) {
        char exe[4096]; const ssize_t n = readlink("/proc/self/exe", exe, sizeof(exe) - 1);
        if (n > 0) {
            exe[n] = 0;
            std::string dir(exe); const size_t cut = dir.rfind('/'); dir = cut == std::string::npos ? "." : dir.substr(0, cut);
            const char *lib = getenv("SP_LIB_NAME");
            g_lib = dlopen((dir + "/" + (lib ? lib : "libspacepharer_amd.so")).c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (!g_lib && lib) g_lib = dlopen((dir + "/csrc/build/" + lib).c_str(), RTLD_NOW | RTLD_GLOBAL);      // (instrumented / sanitized builds: make variant | asan)
        }
        if (!g_lib) g_lib = dlopen("libspacepharer_amd.so", RTLD_NOW | RTLD_GLOBAL);
        if (!g_lib) { fprintf(stderr, "spacepharer-amd: cannot load libspacepharer_amd.so: %s\n", dlerror()); exit(1); }
    }
    void *f = dlsym(g_lib, name);
    if (!f) { fprintf(stderr, "spacepharer-amd: libspacepharer_amd.so has no %s\n", name); exit(1); }
    return f;
}
static int sp_init_async(int mask) { return ((int (*)(int)) libSym("sp_init_async"))(mask); }
static void sp_shutdown() { ((void (*)()) libSym("sp_shutdown"))(); }
static const char *sp_last_error() { return ((const char *(*)()) libSym("sp_last_error"))(); }
constexpr int SP_OK = 0;

struct Module {
    const char *name; const char *symbol; bool device;
    int run(int argc, const char **argv) const { return ((int (*)(int, const char **)) libSym(symbol))(argc, argv); }
};
static const Module kModules[] = {
    {"prefilter", "sp_prefilter_run", true}, {"align", "sp_align_run", true}, {"rescorediagonal", "sp_rescorediagonal_run", true}, {"search", "sp_search_run", true},
    // the set-level steps are host-only: no device is touched for them
    {"extractorfs", "sp_extractorfs_run", false}, {"translatenucs", "sp_translatenucs_run", false}, {"reverseseqbycodon", "sp_reverseseqbycodon_run", false},
    {"proteinaln2nucl", "sp_proteinaln2nucl_run", false}, {"combineprotnuclaln", "sp_combineprotnuclaln_run", false}, {"besthitperset", "sp_besthitperset_run", false},
    {"mergeresultsbyset", "sp_mergeresultsbyset_run", false}, {"combinescore", "sp_combinescore_run", false}, {"filtermatchbyfdr", "sp_filtermatchbyfdr_run", false},
    {"summarizeresults", "sp_summarizeresults_run", false}};

static const Module *findModule(const char *name) { for (const Module &m : kModules) if (!strcmp(name, m.name)) return &m; return nullptr; }

static bool sendAll(int fd, const void *p, size_t n) { const char *c = (const char *) p; while (n) { const ssize_t k = write(fd, c, n); if (k <= 0) return false; c += k; n -= (size_t) k; } return true; }
static bool recvAll(int fd, void *p, size_t n) { char *c = (char *) p; while (n) { const ssize_t k = read(fd, c, n); if (k <= 0) return false; c += k; n -= (size_t) k; } return true; }
static bool sendStr(int fd, const std::string &s) { const uint32_t n = (uint32_t) s.size(); return sendAll(fd, &n, 4) && sendAll(fd, s.data(), n); }
static bool recvStr(int fd, std::string &s) { uint32_t n = 0; if (!recvAll(fd, &n, 4) || n > (1u << 26)) return false; s.resize(n); return n == 0 || recvAll(fd, &s[0], n); }

static int connectTo(const char *path) {
    const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (fd < 0) return -1;
    sockaddr_un a; memset(&a, 0, sizeof(a)); a.sun_family = AF_UNIX; strncpy(a.sun_path, path, sizeof(a.sun_path) - 1);
    if (connect(fd, (sockaddr *) &a, sizeof(a)) != 0) { close(fd); return -1; }
    return fd;
}

// request: cwd, argc, argv[0..argc) (argv[0] = module name); reply: int32 status, message
static int forward(int fd, int argc, const char **argv) {
    char cwd[4096]; if (!getcwd(cwd, sizeof(cwd))) cwd[0] = 0;
    bool ok = sendStr(fd, cwd);
    const uint32_t n = (uint32_t) argc; ok = ok && sendAll(fd, &n, 4);
    for (int i = 0; i < argc && ok; i++) ok = sendStr(fd, argv[i]);
    int32_t rc = 1; std::string msg;
    ok = ok && recvAll(fd, &rc, 4) && recvStr(fd, msg);
    close(fd);
    if (!ok) { fprintf(stderr, "spacepharer-amd: lost the connection to the server\n"); return 1; }
    if (!msg.empty()) fprintf(stderr, "%s\n", msg.c_str());
    return rc;
}

// default socket of the per-user resident server (SP_SERVER=auto)
// ($XDG_RUNTIME_DIR is the user's own; without one the socket lives in a directory this user owns, mode 0700, checked: anybody may create
// /tmp/spacepharer-amd-<uid> first, and a socket in somebody else's directory would hand that somebody the client's argv and the say over its exit status)
static std::string defaultSocket() {
    const char *dir = getenv("XDG_RUNTIME_DIR");
    struct stat st;
    if (dir && *dir && stat(dir, &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == getuid()) return std::string(dir) + "/spacepharer-amd.sock";
    const std::string own = "/tmp/spacepharer-amd-" + std::to_string((long) getuid());
    (void) mkdir(own.c_str(), 0700);
    if (lstat(own.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & 077) != 0) return std::string();      // not ours: no default server
    return own + "/server.sock";
}
// the peer of a connected socket is this user (SO_PEERCRED): a client never talks to somebody else's server
static bool peerIsMe(int fd) {
    ucred cr; socklen_t len = sizeof(cr);
    return getsockopt(fd, SOL_SOCKET, SO_PEERCRED, &cr, &len) == 0 && cr.uid == getuid();
}

// idleSeconds > 0: leave when no client has called for that long (the auto-started server)
static int serve(const char *path, int idleSeconds) {
    signal(SIGPIPE, SIG_IGN);
    // A module call formats tens of megabytes into fresh buffers.  glibc hands every large request straight to mmap and returns it on free, so
    // a resident process would first-touch the same pages again in every call -- and a first touch is slow on a virtualised box (a third
    // of a served call went there).  Keep freed memory in the heap instead: the next call finds it mapped.
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
    // the socket first, the runtime second: clients that connect while the HIP runtime and the tables come up simply wait in the backlog
    {   struct stat st;      // a leftover SOCKET nobody answers on is replaced; anything else at that path is not ours to remove
        if (lstat(path, &st) == 0) {
            if (!S_ISSOCK(st.st_mode)) { fprintf(stderr, "spacepharer-amd serve: %s exists and is not a socket\n", path); return 1; }
            const int probe = connectTo(path);
            if (probe >= 0) { close(probe); fprintf(stderr, "spacepharer-amd serve: another server answers on %s\n", path); return 0; }
            unlink(path);
        } }
    const int ls = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un a; memset(&a, 0, sizeof(a)); a.sun_family = AF_UNIX; strncpy(a.sun_path, path, sizeof(a.sun_path) - 1);
    const mode_t old = umask(077);      // the server runs module calls (chdir, DB writes) as this user for whoever may connect: owner only
    const bool bound = ls >= 0 && bind(ls, (sockaddr *) &a, sizeof(a)) == 0 && listen(ls, 64) == 0;
    umask(old);
    if (!bound) { fprintf(stderr, "spacepharer-amd serve: cannot listen on %s: %s\n", path, strerror(errno)); return 1; }
    // (the runtime comes up on a second thread while the first client's call -- already waiting in the backlog of an auto-started server -- reads its DBs; a machine without a
    // device shows in that call's error)
    if (sp_init_async(0) != SP_OK) { fprintf(stderr, "%s\n", sp_last_error()); close(ls); unlink(path); return 1; }
    if (!getenv("SP_CONTEXT_CACHE")) setenv("SP_CONTEXT_CACHE", "4", 1);
    fprintf(stderr, "spacepharer-amd: serving on %s\n", path);
    for (;;) {      // one call at a time: a module call owns the GPU while it runs
        if (idleSeconds > 0) {
            pollfd pf{ls, POLLIN, 0};
            const int pr = poll(&pf, 1, idleSeconds * 1000);
            if (pr == 0) break;                       // idle: leave, the next client starts a new one
            if (pr < 0) { if (errno == EINTR) continue; break; }
        }
        const int fd = accept(ls, nullptr, nullptr);
        if (fd < 0) { if (errno == EINTR) continue; break; }
        { timeval tv{30, 0}; setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv)); }      // a client that connects and stalls does not hold the server
        std::string cwd; uint32_t argc = 0; std::vector<std::string> args;
        bool ok = recvStr(fd, cwd) && recvAll(fd, &argc, 4) && argc >= 1 && argc < 4096;
        for (uint32_t i = 0; i < argc && ok; i++) { std::string s; ok = recvStr(fd, s); args.push_back(s); }
        if (!ok) { close(fd); continue; }
        if (args[0] == "shutdown") { const int32_t rc = 0; sendAll(fd, &rc, 4); sendStr(fd, ""); close(fd); break; }
        int32_t rc = 1; std::string msg;
        const Module *m = findModule(args[0].c_str());
        if (!m) msg = "unknown module " + args[0];
        else {
            if (!cwd.empty() && chdir(cwd.c_str()) != 0) msg = "cannot enter " + cwd;
            else {
                std::vector<const char *> av; for (size_t i = 1; i < args.size(); i++) av.push_back(args[i].c_str());
                try { rc = m->run((int) av.size(), av.data()); if (rc != 0) msg = sp_last_error(); }
                catch (const std::exception &e) { rc = 1; msg = std::string(args[0]) + ": " + e.what(); }      // (bad_alloc on a huge DB: the call fails, the server stays)
            }
        }
        sendAll(fd, &rc, 4); sendStr(fd, msg);
        close(fd);
    }
    close(ls); unlink(path);
    sp_shutdown();
    return 0;
}

// SP_SERVER=auto: connect to this user's resident server, starting it first when nobody answers.  The client process never touches the GPU, so
// the server is a plain child of it: detached (own session, stdio closed), this binary again with `serve <socket> --idle <seconds>`.
static int connectOrStart(const std::string &path, const char *self, int idleSeconds) {
    int fd = connectTo(path.c_str());
    if (fd >= 0) return fd;
    const pid_t pid = fork();
    if (pid < 0) return -1;
    if (pid == 0) {
        if (setsid() < 0) _exit(1);
        const int nul = open("/dev/null", O_RDWR);
        const char *log = getenv("SP_SERVER_LOG");
        const int err = log ? open(log, O_WRONLY | O_CREAT | O_APPEND, 0600) : nul;
        if (nul >= 0) { dup2(nul, 0); dup2(nul, 1); }
        if (err >= 0) dup2(err, 2);
        const std::string idle = std::to_string(idleSeconds);
        char exe[4096]; const ssize_t n = readlink("/proc/self/exe", exe, sizeof(exe) - 1);
        if (n > 0) { exe[n] = 0; execl(exe, exe, "serve", path.c_str(), "--idle", idle.c_str(), (char *) nullptr); }
        execlp(self, self, "serve", path.c_str(), "--idle", idle.c_str(), (char *) nullptr);
        _exit(1);
    }
    for (int waited = 0; waited < 20000; waited += 2) {      // the socket appears within milliseconds; requests wait in its backlog while the runtime starts
        fd = connectTo(path.c_str());
        if (fd >= 0) return fd;
        int st; if (waitpid(pid, &st, WNOHANG) == pid) break;      // the server could not start (no device, socket path not writable)
        usleep(2000);
    }
    return connectTo(path.c_str());
}

int main(int argc, const char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s <module> <args>   with module one of:", argv[0]);
        for (const Module &m : kModules) fprintf(stderr, " %s", m.name);
        fprintf(stderr, "\n       %s serve <socket> [--idle <seconds>]   run module calls for clients that set SP_SERVER=<socket>\n"
                        "       SP_SERVER=auto[:<idle seconds>] %s <module> <args>   forward to this user's resident server, starting it when nobody answers\n", argv[0], argv[0]);
        return 1;
    }
    if (!strcmp(argv[1], "serve")) {
        if (argc != 3 && !(argc == 5 && !strcmp(argv[3], "--idle"))) { fprintf(stderr, "usage: %s serve <socket> [--idle <seconds>]\n", argv[0]); return 1; }
        return serve(argv[2], argc == 5 ? atoi(argv[4]) : 0);
    }
    if (const char *srv = getenv("SP_SERVER")) {
        const bool known = !strcmp(argv[1], "shutdown") || findModule(argv[1]);
        if (known && !strncmp(srv, "auto", 4) && (srv[4] == 0 || srv[4] == ':')) {
            const std::string path = defaultSocket();
            const Module *m = findModule(argv[1]);
            int fd = -1;
            if (!path.empty()) {
                // two clients starting at once must not each unlink the other's freshly bound socket: the start is serialised by a lock file next to it
                const int lock = open((path + ".lock").c_str(), O_CREAT | O_RDWR, 0600);
                if (lock >= 0) (void) flock(lock, LOCK_EX);
                fd = (m && m->device) ? connectOrStart(path, argv[0], srv[4] == ':' ? std::max(1, atoi(srv + 5)) : 120) : connectTo(path.c_str());      // host-only steps and `shutdown` never start one
                if (lock >= 0) { (void) flock(lock, LOCK_UN); close(lock); }
                if (fd >= 0 && !peerIsMe(fd)) { close(fd); fd = -1; }      // (somebody else's process answers there: run the module in this process instead)
            }
            if (fd >= 0) return forward(fd, argc - 1, argv + 1);
        } else if (known) { const int fd = connectTo(srv); if (fd >= 0) return forward(fd, argc - 1, argv + 1); }
        if (!strcmp(argv[1], "shutdown")) return 0;
    }
    const Module *m = findModule(argv[1]);
    if (!m) { fprintf(stderr, "unknown module %s\n", argv[1]); return 1; }
    // SP_MODULE_TIMES=1: where a fresh process spends its time AROUND the module's own laps -- since exec (the kernel's start time of this process), loading the library,
    // bringing the runtime up, the module, and (by the caller's clock) whatever the exit costs
    const bool times = getenv("SP_MODULE_TIMES") != nullptr;
    auto nowMs = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    auto sinceExecMs = [&]() -> double {      // /proc/self/stat field 22: start time in clock ticks since boot (CLOCK_BOOTTIME's origin)
        FILE *f = fopen("/proc/self/stat", "r"); if (!f) return -1;
        char buf[2048]; const size_t n = fread(buf, 1, sizeof(buf) - 1, f); fclose(f); buf[n] = 0;
        const char *p = strrchr(buf, ')'); if (!p) return -1;
        unsigned long long start = 0; int field = 2;
        for (p++; *p && field < 22; p++) if (*p == ' ') field++;
        start = strtoull(p, nullptr, 10);
        timespec ts; clock_gettime(CLOCK_BOOTTIME, &ts);
        return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6 - (double) start * 1e3 / (double) sysconf(_SC_CLK_TCK);
    };
    const double tMain = nowMs(), preMain = times ? sinceExecMs() : 0;
    // SP_GPUS=N: the module starts one worker process per GPU (this binary again, with SP_PART set); the parent only merges
    // the parts and has no use for a device
    const char *gpus = getenv("SP_GPUS");
    const bool fanOut = gpus && atoi(gpus) > 1;
    if (m->device && !fanOut) setenv("SP_ONESHOT", "1", 0);      // (before the library loads: it reads the variable once)
    (void) libSym("sp_last_error");
    const double tLib = nowMs();
    // this process runs the module itself and ends with it: the runtime comes up on a second thread while the module reads its DBs (the module's context waits for it), result
    // buffers are ordinary memory and nothing is torn down in order at the end (SP_ONESHOT: sp_modules.cpp)
    if (m->device && !fanOut) { if (sp_init_async(0) != SP_OK) { fprintf(stderr, "%s\n", sp_last_error()); return 1; } }
    const double tInit = nowMs();
    const int rc = m->run(argc - 2, argv + 2);
    if (rc != 0) fprintf(stderr, "%s\n", sp_last_error());
    if (times) fprintf(stderr, "[sp] process: exec -> main %.1f ms (10 ms resolution), library loaded %.1f ms, runtime started on a second thread %.1f ms, module %.1f ms; since exec %.1f ms at exit\n",
                       preMain, tLib - tMain, tInit - tLib, nowMs() - tInit, sinceExecMs());
    // The outputs are on disk (write + close, nothing buffered).  Tearing the HIP runtime down in order -- contexts, 0.5 GB of tables,
    // code objects -- takes ~50 ms of a module call whose GPU work takes less; the kernel driver reclaims everything at exit anyway.
    // SP_ORDERLY_EXIT=1 keeps the orderly path (leak checkers).
    if (m->device && !getenv("SP_ORDERLY_EXIT")) { fflush(NULL); _exit(rc); }
    if (m->device) sp_shutdown();
    return rc;
}
