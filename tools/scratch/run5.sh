mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_shard.py tests/test_gpu_rccl.py tests/test_gpu_bench.py tests/test_gpu_parity.py -k "shard or rccl or bench or feeder or adam_three or bag" -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r3_tests5.log 2>&1; echo "rc $?" >> gpurun_out/r3_tests5.log)
tail -8 gpurun_out/r3_tests5.log | cut -c1-250
(GQE_SHARD_PROFILE=1 timeout 300 python tools/shard_overhead_bench.py > gpurun_out/r3_shard_overhead.log 2>&1)
grep "us/step\|shard profile" gpurun_out/r3_shard_overhead.log || tail -5 gpurun_out/r3_shard_overhead.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_shard -o shard -- python $GRAFT_REPO_ROOT/tools/shard_overhead_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r3_prof_shard.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_shard -name "*kernel_stats*" | head -2 | xargs -r head -30 | cut -c1-220
