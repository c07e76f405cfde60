// Front-end binary: `spacepharer-amd <module> <args...>` with module in {prefilter, align, rescorediagonal, search} and the set-level
// steps around them (extractorfs ... summarizeresults).
// Point $MMSEQS at a wrapper that sends these module names here and every other one to the stock spacepharer binary, and the
// unmodified workflow scripts (lib/mmseqs/data/workflow/blastp.sh:33,41,48) land in the MI355X implementation.
//
// `spacepharer-amd serve <socket>` keeps ONE process with the HIP runtime up and the last target sets resident in HBM and runs
// module calls for clients: with SP_SERVER=<socket> in the environment `spacepharer-amd <module> <args...>` forwards its argv
// (and working directory) there and exits with the module's status.  A module call then costs what it costs in a warm process
// -- a fresh process pays ~0.2 s of runtime start-up, tables and first-touch page mapping before its first kernel, more than the
// whole search of 10^5 spacers takes.  Without a reachable server the call runs in this process as before.
#include <cerrno>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>
#include <vector>

#include "../../include/spacepharer_amd.h"

struct Module { const char *name; int (*run)(int, const char **); bool device; };
static const Module kModules[] = {
    {"prefilter", sp_prefilter_run, true}, {"align", sp_align_run, true}, {"rescorediagonal", sp_rescorediagonal_run, true}, {"search", sp_search_run, true},
    // the set-level steps are host-only: no device is touched for them
    {"extractorfs", sp_extractorfs_run, false}, {"translatenucs", sp_translatenucs_run, false}, {"reverseseqbycodon", sp_reverseseqbycodon_run, false},
    {"proteinaln2nucl", sp_proteinaln2nucl_run, false}, {"combineprotnuclaln", sp_combineprotnuclaln_run, false}, {"besthitperset", sp_besthitperset_run, false},
    {"mergeresultsbyset", sp_mergeresultsbyset_run, false}, {"combinescore", sp_combinescore_run, false}, {"filtermatchbyfdr", sp_filtermatchbyfdr_run, false},
    {"summarizeresults", sp_summarizeresults_run, false}};

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

static int serve(const char *path) {
    signal(SIGPIPE, SIG_IGN);
    if (sp_init(0) != SP_OK) { fprintf(stderr, "%s\n", sp_last_error()); return 1; }
    if (!getenv("SP_CONTEXT_CACHE")) setenv("SP_CONTEXT_CACHE", "4", 1);
    const int ls = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un a; memset(&a, 0, sizeof(a)); a.sun_family = AF_UNIX; strncpy(a.sun_path, path, sizeof(a.sun_path) - 1);
    unlink(path);
    if (ls < 0 || bind(ls, (sockaddr *) &a, sizeof(a)) != 0 || listen(ls, 64) != 0) { fprintf(stderr, "spacepharer-amd serve: cannot listen on %s: %s\n", path, strerror(errno)); return 1; }
    fprintf(stderr, "spacepharer-amd: serving on %s\n", path);
    for (;;) {      // one call at a time: a module call owns the GPU while it runs
        const int fd = accept(ls, nullptr, nullptr);
        if (fd < 0) { if (errno == EINTR) continue; break; }
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
                rc = m->run((int) av.size(), av.data());
                if (rc != 0) msg = sp_last_error();
            }
        }
        sendAll(fd, &rc, 4); sendStr(fd, msg);
        close(fd);
    }
    close(ls); unlink(path);
    sp_shutdown();
    return 0;
}

int main(int argc, const char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s <module> <args>   with module one of:", argv[0]);
        for (const Module &m : kModules) fprintf(stderr, " %s", m.name);
        fprintf(stderr, "\n       %s serve <socket>          run module calls for clients that set SP_SERVER=<socket>\n", argv[0]);
        return 1;
    }
    if (!strcmp(argv[1], "serve")) { if (argc != 3) { fprintf(stderr, "usage: %s serve <socket>\n", argv[0]); return 1; } return serve(argv[2]); }
    if (const char *srv = getenv("SP_SERVER")) {
        if (!strcmp(argv[1], "shutdown") || findModule(argv[1])) { const int fd = connectTo(srv); if (fd >= 0) return forward(fd, argc - 1, argv + 1); }
        if (!strcmp(argv[1], "shutdown")) return 0;
    }
    const Module *m = findModule(argv[1]);
    if (!m) { fprintf(stderr, "unknown module %s\n", argv[1]); return 1; }
    // SP_GPUS=N: the module starts one worker process per GPU (this binary again, with SP_PART set); the parent only merges
    // the parts and has no use for a device
    const char *gpus = getenv("SP_GPUS");
    const bool fanOut = gpus && atoi(gpus) > 1;
    if (m->device && !fanOut && sp_init(0) != SP_OK) { fprintf(stderr, "%s\n", sp_last_error()); return 1; }
    const int rc = m->run(argc - 2, argv + 2);
    if (rc != 0) fprintf(stderr, "%s\n", sp_last_error());
    // The outputs are on disk (write + close, nothing buffered).  Tearing the HIP runtime down in order -- contexts, 0.5 GB of tables,
    // code objects -- takes ~50 ms of a module call whose GPU work takes less; the kernel driver reclaims everything at exit anyway.
    // SP_ORDERLY_EXIT=1 keeps the orderly path (leak checkers).
    if (m->device && !getenv("SP_ORDERLY_EXIT")) { fflush(NULL); _exit(rc); }
    if (m->device) sp_shutdown();
    return rc;
}
