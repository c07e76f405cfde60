#!/bin/bash
# configs[4]'s chain (tools/predictmatch_full.py): a small case first (results of the pipelined chain against the sequential one), then the full size
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r05
timeout 600 python3 tools/predictmatch_full.py 90 20000 10 500 20 16 > gpurun_out/r05/pm_small.json 2> gpurun_out/r05/pm_small.err; tail -3 gpurun_out/r05/pm_small.err
python3 -c "
import json; d=json.load(open('gpurun_out/r05/pm_small.json')); print(d['chain_s_without_python_input_writing'], d['spacers_per_s_whole_chain'], d.get('pipelined'))"
timeout 900 python3 tools/predictmatch_full.py ${FULL_ARGS:-} > gpurun_out/r05/pm_full.json 2> gpurun_out/r05/pm_full.err; tail -30 gpurun_out/r05/pm_full.err
python3 -c "
import json; d=json.load(open('gpurun_out/r05/pm_full.json')); print(d['seconds']); print(d['chain_s_without_python_input_writing'], d['spacers_per_s_whole_chain'], d.get('pipelined'))"
