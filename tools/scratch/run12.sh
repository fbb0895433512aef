mkdir -p gpurun_out
bash tools/profile_round.sh r03 bio-synth
bash tools/profile_round.sh r03 reddit-synth
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-reddit > gpurun_out/r03_bench_2rank_gloo.json 2> gpurun_out/r03_bench_2rank.err
tail -2 gpurun_out/r03_bench_2rank.err | cut -c1-300
(GQE_SHARD_PROFILE=1 timeout 300 python tools/shard_overhead_bench.py 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl") > gpurun_out/r03_tool_shard_overhead_bench.log
cat gpurun_out/r03_tool_shard_overhead_bench.log | cut -c1-200
for a in "" "--workload reddit-synth" "--decoder bilinear"; do timeout 300 python tools/eval_bench.py $a 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r03_tool_eval_bench.log
(timeout 300 python tools/train_bench.py 2>&1 | grep -v amdgpu.ids | tail -12) > gpurun_out/r03_tool_train_bench.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03_shard -o shard -- python $GRAFT_REPO_ROOT/tools/shard_overhead_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find gpurun_out/prof_r03_shard -name "*.db" | head -1) > gpurun_out/r03_shard_kernel_stats.txt
head -14 gpurun_out/r03_shard_kernel_stats.txt | cut -c1-150
python tools/collect_profiles.py r03
