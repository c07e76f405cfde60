# GPU box: wall-clock laps of ONE fresh `spacepharer-amd search` process on the bench workload (SP_MODULE_TIMES=1)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, subprocess, tempfile, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from test_gpu_modules import write_seq_db, PREF_FLAGS, ALN_FLAGS
n = int(os.environ.get("SPACERS", "100000"))
tres, toff, qres, qoff = bench.make_workload(n, 50000, 200, 12345)
T = [tres[int(toff[i]):int(toff[i + 1])].tobytes().decode() for i in range(len(toff) - 1)]
Q = [qres[int(qoff[i]):int(qoff[i + 1])].tobytes().decode() for i in range(len(qoff) - 1)]
d = tempfile.mkdtemp()
write_seq_db(d + "/q", Q); write_seq_db(d + "/t", T)
pf = (PREF_FLAGS % 300).replace("--threads 8", "--threads 64").split()
sf = pf + ["-a", "1", "--alignment-mode", "2", "-e", "200", "--gap-open", "nucl:10,aa:16", "--gap-extend", "2"]
for rep in range(2):
    t0 = time.perf_counter()
    p = subprocess.run(["spacepharer_amd/spacepharer-amd", "search", d + "/q", d + "/t", d + "/aln%d" % rep, d] + sf, env=dict(os.environ, SP_MODULE_TIMES="1"), capture_output=True, text=True)
    print("process wall %.3f s rc=%d" % (time.perf_counter() - t0, p.returncode)); print(p.stderr[-1500:])
PY
