#!/bin/bash
# final round-3 evidence: kernel traces + PMC passes for both workloads, the default bench line, the 2-rank gloo line, tool logs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/profile_round.sh r03 bio-synth
bash tools/profile_round.sh r03 reddit-synth
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
timeout 900 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 > gpurun_out/r03_bench_2rank_gloo.json 2> gpurun_out/r03_bench_2rank_gloo.err
timeout 600 python tools/kbench.py > gpurun_out/r03_tool_kbench.log 2>&1
timeout 600 python tools/kbench.py --durations 4096 > gpurun_out/r03_tool_kbench_durations_B4096.log 2>&1
timeout 600 python tools/kbench.py --durations 512 --workload reddit-synth --dim 256 > gpurun_out/r03_tool_kbench_durations_reddit.log 2>&1
rm -rf gpurun_out/prof_r03_* gpurun_out/pmc_r03_*
