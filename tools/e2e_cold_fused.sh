# GPU box: laps (SP_MODULE_TIMES=1) of fresh fused `search` processes at 100 000 spacers, TARGETS targets; extra environment in EXTRA ("A=1 B=2"), REPS runs
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, subprocess, tempfile, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from test_gpu_modules import write_seq_db, PREF_FLAGS
n = int(os.environ.get("SPACERS", "100000"))
tres, toff, qres, qoff = bench.make_workload(n, int(os.environ.get("TARGETS", "1000000")), 200, 12345)
T = [tres[int(toff[i]):int(toff[i + 1])].tobytes().decode() for i in range(len(toff) - 1)]
Q = [qres[int(qoff[i]):int(qoff[i + 1])].tobytes().decode() for i in range(len(qoff) - 1)]
d = tempfile.mkdtemp()
write_seq_db(d + "/q", Q); write_seq_db(d + "/t", T)
th = str(bench.host_info()["cores_for_cpu_baseline"])
pf = (PREF_FLAGS % 300).replace("--threads 8", "--threads " + th).split()
sf = pf + ["-a", "1", "--alignment-mode", "2", "-e", "200", "--gap-open", "nucl:10,aa:16", "--gap-extend", "2"]
exe = "spacepharer_amd/spacepharer-amd"
for extra in os.environ.get("EXTRAS", "").split(";"):
    env = dict(os.environ, SP_MODULE_TIMES="1")
    for kv in extra.split():
        k, v = kv.split("="); env[k] = v
    for rep in range(int(os.environ.get("REPS", "2"))):
        t0 = time.perf_counter()
        p = subprocess.run([exe, "search", d + "/q", d + "/t", d + "/alnf%d" % rep, d] + sf, env=env, capture_output=True, text=True)
        import zlib
        crc = "%08x" % zlib.crc32(open(d + "/alnf%d.index" % rep, "rb").read(), zlib.crc32(open(d + "/alnf%d" % rep, "rb").read())) if p.returncode == 0 else "-"
        print("== [%s] fused search process %.3f s rc=%d aln_db_crc %s" % (extra, time.perf_counter() - t0, p.returncode, crc)); print("\n".join(l for l in p.stderr.splitlines() if "search:" in l or "process:" in l or "ctx_create" in l or "runtime" in l or ("MEM_SUMMARY" in os.environ and ("hipMalloc" in l or "recycled" in l))))
PY
