# GPU box: wall-clock laps (SP_MODULE_TIMES=1, from the server's log) of prefilter + align client processes against the auto-started server
cd $GRAFT_REPO_ROOT
python - <<'PY'
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
env = dict(os.environ, SP_MODULE_TIMES="1", SP_SERVER="auto:20", XDG_RUNTIME_DIR=d, SP_SERVER_LOG=d + "/server.log")
exe = "spacepharer_amd/spacepharer-amd"
for rep in range(3):
    mark = os.path.getsize(d + "/server.log") if os.path.exists(d + "/server.log") else 0
    t0 = time.perf_counter()
    p1 = subprocess.run([exe, "prefilter", d + "/q", d + "/t", d + "/pref%d" % rep] + pf, env=env, capture_output=True, text=True)
    t1 = time.perf_counter()
    p2 = subprocess.run([exe, "align", d + "/q", d + "/t", d + "/pref%d" % rep, d + "/aln%d" % rep] + af, env=env, capture_output=True, text=True)
    t2 = time.perf_counter()
    print("== %d spacers, pair %d: prefilter client %.3f s (rc %d), align client %.3f s (rc %d), both %.3f s" % (n, rep, t1 - t0, p1.returncode, t2 - t1, p2.returncode, t2 - t0))
    print(open(d + "/server.log").read()[mark:])
subprocess.run([exe, "shutdown"], env=env)
PY
