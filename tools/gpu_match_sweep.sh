cd $GRAFT_REPO_ROOT
for n in 256 512 1024 2048 4096 6144 8192 10000 12288; do
  timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --stage prefilter --spacers $n 2>&1 | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print($n, d['kernels_ms_per_step']['match'], d['counters_per_step']['similar_kmers'])"
done
