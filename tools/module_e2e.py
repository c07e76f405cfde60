"""End-to-end timing of the module drop-ins on MMseqs2 DB triples ("DBs on disk" -> "result DB written"): `prefilter` then `align`
with the workflow's command lines, and the fused `search`, on the bench.py workload.  Wall clock per call =
entry into the module function to result DB written (reading the DBs, target set preparation on the device, search, formatting,
writing).  `cold` = the first call of a fresh process (pays the HIP runtime start-up and the similar-3-mer table), `warm` = the
same call again in the same process.  One JSON line; bench.py embeds it under "module_e2e".
usage: python tools/module_e2e.py [spacers ...]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(n_spacers, d, mode, n_targets=50000, light=False):
    import bench
    import spacepharer_amd as sa
    from test_gpu_modules import ALN_FLAGS, PREF_FLAGS, write_seq_db
    threads = str(bench.host_info()["cores_for_cpu_baseline"])      # what this container may use (affinity mask, cgroup quota)
    pf = [x if x != "8" or PREF_FLAGS.split()[i - 1] != "--threads" else threads for i, x in enumerate((PREF_FLAGS % 300).split())]
    af = [x if x != "8" or ALN_FLAGS.split()[i - 1] != "--threads" else threads for i, x in enumerate(ALN_FLAGS.split())]
    sf = pf + [x for i, x in enumerate(af) if not (x in pf and x.startswith("-")) and not (i > 0 and af[i - 1] in pf and af[i - 1].startswith("-"))]
    tres, toff, qres, qoff = bench.make_workload(n_spacers, n_targets, 200, 12345)
    T = [tres[int(toff[i]):int(toff[i + 1])].tobytes().decode() for i in range(len(toff) - 1)]
    Q = [qres[int(qoff[i]):int(qoff[i + 1])].tobytes().decode() for i in range(len(qoff) - 1)]
    q, t, pref, aln = (os.path.join(d, x) for x in ("qdb", "tdb", "pref", "aln"))
    write_seq_db(q, Q)
    write_seq_db(t, T)
    out = {"spacers": n_spacers, "targets": n_targets, "target_residues": int(toff[-1]), "threads": int(threads), "mode": mode, "runs": []}
    for rep in range(2 if light else 3):
        for f in os.listdir(d):
            if f.startswith("pref") or f.startswith("aln"):
                os.remove(os.path.join(d, f))
        t0 = time.perf_counter()
        if mode == "two_step":
            sa.module_prefilter([q, t, pref] + pf)
            t1 = time.perf_counter()
            sa.module_align([q, t, pref, aln] + af)
            out["runs"].append({"prefilter_s": round(t1 - t0, 4), "align_s": round(time.perf_counter() - t1, 4), "total_s": round(time.perf_counter() - t0, 4)})
        else:
            sa.module_search([q, t, aln, d] + sf)
            out["runs"].append({"total_s": round(time.perf_counter() - t0, 4)})
    # the same calls as fresh PROCESSES of the front-end binary -- what a workflow script pays per module call: process start, HIP
    # runtime start-up, similar-3-mer table, first touch of every device buffer
    exe = os.path.join(ROOT, "spacepharer_amd", "spacepharer-amd")
    procs = []
    for rep in range(2):      # (the first fresh process on a box also pays for the driver's own warm-up: the better of two counts)
        for f in os.listdir(d):
            if f.startswith("pref") or f.startswith("aln"):
                os.remove(os.path.join(d, f))
        t0 = time.perf_counter()
        if mode == "two_step":
            subprocess.run([exe, "prefilter", q, t, pref] + pf, check=True, capture_output=True)
            subprocess.run([exe, "align", q, t, pref, aln] + af, check=True, capture_output=True)
        else:
            subprocess.run([exe, "search", q, t, aln, d] + sf, check=True, capture_output=True)
        procs.append(round(time.perf_counter() - t0, 4))
    out["process_total_s"] = procs
    out["process_spacer_queries_per_s"] = n_spacers / min(procs)
    # where a fresh process of the fused call spends its time (SP_MODULE_TIMES=1: the front end's and the module's own laps), one more run
    if mode != "two_step":
        import re
        for f in os.listdir(d):
            if f.startswith("aln"):
                os.remove(os.path.join(d, f))
        t0 = time.perf_counter()
        p = subprocess.run([exe, "search", q, t, aln, d] + sf, check=True, capture_output=True, text=True, env=dict(os.environ, SP_MODULE_TIMES="1"))
        wall = time.perf_counter() - t0
        laps = {"wall_s": round(wall, 4)}
        for ln in p.stderr.splitlines():
            m = re.match(r"\[sp\] search: (.+?)\s+([0-9.]+) ms$", ln)
            if m:
                laps[m.group(1).strip()] = float(m.group(2))
            m = re.search(r"process: exec -> main ([0-9.]+) ms.*library loaded ([0-9.]+) ms.*module ([0-9.]+) ms; since exec ([0-9.]+) ms at exit", ln)
            if m:
                laps["exec_to_main_ms"], laps["library_loaded_ms"], laps["module_ms"], laps["since_exec_at_exit_ms"] = (float(x) for x in m.groups())
                laps["exit_and_teardown_ms"] = round(wall * 1e3 - laps["since_exec_at_exit_ms"], 1)
            m = re.search(r"runtime \(second thread\): .*code object \+ first kernel ([0-9.]+) ms", ln)
            if m:
                laps["code_object_and_first_kernel_ms"] = float(m.group(1))
        laps["note"] = "`context (matrices, tables)` is mostly the wait for the HIP runtime, which comes up on a second thread from the process's first millisecond while the DBs are read"
        out["fresh_process_laps"] = laps
    # the same client processes against a resident server (`spacepharer-amd serve`): runtime, tables and the target set stay up
    sock = os.path.join(d, "sp.sock")
    srv = subprocess.Popen([exe, "serve", sock], stderr=subprocess.DEVNULL)
    for _ in range(600):
        if os.path.exists(sock):
            break
        time.sleep(0.01)
    served = []
    env = dict(os.environ, SP_SERVER=sock)
    for rep in range(3 if light else 4):
        for f in os.listdir(d):
            if f.startswith("pref") or f.startswith("aln"):
                os.remove(os.path.join(d, f))
        t0 = time.perf_counter()
        if mode == "two_step":
            subprocess.run([exe, "prefilter", q, t, pref] + pf, check=True, capture_output=True, env=env)
            subprocess.run([exe, "align", q, t, pref, aln] + af, check=True, capture_output=True, env=env)
        else:
            subprocess.run([exe, "search", q, t, aln, d] + sf, check=True, capture_output=True, env=env)
        served.append(round(time.perf_counter() - t0, 4))
    subprocess.run([exe, "shutdown"], env=env, capture_output=True)
    srv.wait(timeout=60)
    out["served_total_s"] = served
    out["served_spacer_queries_per_s"] = n_spacers / min(served[1:])
    # ... and what a workflow gets WITHOUT setting anything up: SP_SERVER=auto -- the first module process finds no server, starts one (detached,
    # leaves after an idle period) and forwards; the following processes (align, then the control set's prefilter + align) are served warm, and
    # `align` aligns the hits its `prefilter` left resident.  first = the pair that starts the server, next = the pairs after it.
    rt = os.path.join(d, "rt")
    os.makedirs(rt, exist_ok=True)
    env = dict(os.environ, SP_SERVER="auto:30", XDG_RUNTIME_DIR=rt)
    auto = []
    for rep in range(3 if light else 4):
        for f in os.listdir(d):
            if f.startswith("pref") or f.startswith("aln"):
                os.remove(os.path.join(d, f))
        t0 = time.perf_counter()
        if mode == "two_step":
            subprocess.run([exe, "prefilter", q, t, pref] + pf, check=True, capture_output=True, env=env)
            subprocess.run([exe, "align", q, t, pref, aln] + af, check=True, capture_output=True, env=env)
        else:
            subprocess.run([exe, "search", q, t, aln, d] + sf, check=True, capture_output=True, env=env)
        auto.append(round(time.perf_counter() - t0, 4))
    subprocess.run([exe, "shutdown"], env=env, capture_output=True)
    out["auto_server_total_s"] = auto
    out["auto_server_first_spacer_queries_per_s"] = n_spacers / auto[0]
    out["auto_server_next_spacer_queries_per_s"] = n_spacers / min(auto[1:])
    import zlib
    out["aln_db_bytes"] = os.path.getsize(aln)
    out["aln_db_crc"] = "%08x" % zlib.crc32(open(aln + ".index", "rb").read(), zlib.crc32(open(aln, "rb").read()))
    out["cold_spacer_queries_per_s"] = n_spacers / out["runs"][0]["total_s"]
    out["warm_spacer_queries_per_s"] = n_spacers / min(r["total_s"] for r in out["runs"][1:])
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else 50000, len(sys.argv) > 6 and sys.argv[6] == "light")
        sys.exit(0)
    # --targets N: the target set (default 50 000 x 200 aa = the bench proxy; 1 000 000 = the survey's estimate of GenBank_phage_2018_09); --light: fewer repetitions
    n_targets = int(sys.argv[sys.argv.index("--targets") + 1]) if "--targets" in sys.argv else 50000
    light = "--light" in sys.argv
    sizes = [int(a) for i, a in enumerate(sys.argv[1:], 1) if a.isdigit() and sys.argv[i - 1] != "--targets"] or [10000, 100000]
    res = []
    for n in sizes:      # a fresh process per size and mode: its first call is the cold one
        for mode in ("two_step", "fused"):
            with tempfile.TemporaryDirectory() as d:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(n), d, mode, str(n_targets)] + (["light"] if light else []), capture_output=True, text=True, timeout=1800)
                if p.returncode != 0:
                    sys.exit(p.stderr[-3000:])
                res.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]))
                sys.stderr.write(p.stderr[-6000:] if os.environ.get("SP_MODULE_TIMES") else "")
    for a, b in zip(res[0::2], res[1::2]):
        assert a["aln_db_crc"] == b["aln_db_crc"], "the fused search wrote a different alignment DB than prefilter + align"
    print(json.dumps({"module_e2e": res}))
