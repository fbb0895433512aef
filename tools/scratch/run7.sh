mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_parity.py -k "golden_scores or baseline_configs or eight_wave or full_size" -m gpu -q -x --no-header -p no:cacheprovider > gpurun_out/r3_tests7.log 2>&1; echo "rc $?" >> gpurun_out/r3_tests7.log)
tail -4 gpurun_out/r3_tests7.log | cut -c1-250
for B in 512 8192; do
  python bench.py --only-main --batch-size $B --steps 20 --warmup 5 > gpurun_out/r3_b7_$B.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r3_b7_$B.json"))
print("B=$B", d["value"], d["ms_per_step"], {k:v["avg_launch_ms"] for k,v in d["kernels"].items()}, d["roofline"]["avg_launch_ms"])
PY
done
cd /tmp && export TMPDIR=/tmp
for B in 512 8192; do
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc7_$B -o p -- python $R/bench.py --only-main --batch-size $B --steps 20 --warmup 5 --min-seconds 0 > $R/gpurun_out/r3_pmc7_$B.log 2>&1
done
cd $R
python - <<'PY'
import sqlite3
for B in (512, 8192):
    db=sqlite3.connect("gpurun_out/pmc7_%d/p_results.db"%B); cur=db.cursor()
    for r in cur.execute("select kernel_name,counter_name,avg(value),count(*) from counters_collection where kernel_name like '%fused%' or kernel_name like '%pair_gemm%' group by kernel_name,counter_name"):
        print("B=%d"%B, r[0][:30], r[1], "%.0f"%r[2], r[3])
PY
