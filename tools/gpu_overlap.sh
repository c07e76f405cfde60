# GPU box: how much slower is every kernel of a step when it runs beside the other context's k_match?  k_match at 1 / 2 / 3 blocks per CU
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mb in 1 2 3; do
  rm -rf $R/gpurun_out/ov$mb
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ov$mb -o tl -- python3 $R/bench.py --steps 30 --warmup 3 --regions 2 --cpu-sample 0 --no-module-e2e --pipeline 2 --match-blocks $mb > /dev/null 2>&1
  echo "== device-driven x2, k_match blocks per CU: $mb"; python3 $R/tools/timeline_overlap.py $R/gpurun_out/ov$mb
done
rm -rf $R/gpurun_out/ovh
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ovh -o tl -- python3 $R/bench.py --steps 40 --warmup 3 --regions 2 --cpu-sample 0 --no-module-e2e --pipeline 4 --host-driven > /dev/null 2>&1
echo "== host-driven x4, k_match blocks per CU: 3"; python3 $R/tools/timeline_overlap.py $R/gpurun_out/ovh
