cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_full
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_full -o full -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 3 --cpu-sample 0 2>&1 | tail -1 | cut -c1-300
cat $GRAFT_REPO_ROOT/gpurun_out/prof_full/full_kernel_stats.csv
