mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -k "candidate or eval or perc or auc" -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r3_tests6.log 2>&1; echo "rc $?" >> gpurun_out/r3_tests6.log)
tail -8 gpurun_out/r3_tests6.log | cut -c1-250
(GQE_SHARD_PROFILE=1 timeout 300 python tools/shard_overhead_bench.py > gpurun_out/r3_shard_overhead.log 2>&1)
grep "us/step\|shard profile" gpurun_out/r3_shard_overhead.log || tail -5 gpurun_out/r3_shard_overhead.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/r3_sq_counters.txt
for B in 512 8192; do
  for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS"; do
    tag=$(echo $pass | cut -d' ' -f1)
    timeout 600 rocprofv3 --pmc $pass -d $R/gpurun_out/pmc_f_${B}_$tag -o p -- python $R/bench.py --only-main --batch-size $B --steps 20 --warmup 5 --min-seconds 0 > $R/gpurun_out/r3_pmc_${B}_$tag.log 2>&1
  done
done
cd $R
ls gpurun_out | grep pmc_f_
