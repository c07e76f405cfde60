# GPU box: wall-clock laps (SP_MODULE_TIMES=1, from the server's log) of the fused `search` client process against the auto-started server
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import os, sys, subprocess, tempfile, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from test_gpu_modules import write_seq_db, PREF_FLAGS, ALN_FLAGS
n = int(os.environ.get("SPACERS", "100000"))
tres, toff, qres, qoff = bench.make_workload(n, int(os.environ.get("TARGETS", "50000")), 200, 12345)
T = [tres[int(toff[i]):int(toff[i + 1])].tobytes().decode() for i in range(len(toff) - 1)]
Q = [qres[int(qoff[i]):int(qoff[i + 1])].tobytes().decode() for i in range(len(qoff) - 1)]
d = tempfile.mkdtemp()
write_seq_db(d + "/q", Q); write_seq_db(d + "/t", T)
th = str(bench.host_info()["cores_for_cpu_baseline"])
pf = (PREF_FLAGS % 300).replace("--threads 8", "--threads " + th).split()
af = ALN_FLAGS.replace("--threads 8", "--threads " + th).split()
sf = pf + [x for i, x in enumerate(af) if not (x in pf and x.startswith("-")) and not (i > 0 and af[i - 1] in pf and af[i - 1].startswith("-"))]
env = dict(os.environ, SP_MODULE_TIMES="1", SP_SERVER="auto:20", XDG_RUNTIME_DIR=d, SP_SERVER_LOG=d + "/server.log")
exe = "spacepharer_amd/spacepharer-amd"
for rep in range(4):
    mark = os.path.getsize(d + "/server.log") if os.path.exists(d + "/server.log") else 0
    t0 = time.perf_counter()
    p1 = subprocess.run([exe, "search", d + "/q", d + "/t", d + "/aln%d" % rep, d] + sf, env=env, capture_output=True, text=True)
    t1 = time.perf_counter()
    print("== %d spacers, call %d: search client %.4f s (rc %d) %s" % (n, rep, t1 - t0, p1.returncode, p1.stderr[-200:] if p1.returncode else ""))
    if rep in (0, 3):
        print(open(d + "/server.log").read()[mark:])
subprocess.run([exe, "shutdown"], env=env)
PY
