for kg in 0 2 4 8; do
  if [ $kg = 0 ]; then export GQE_GEMM_BIG_MIN_UNITS=100000000; unset GQE_GEMM_KG; else unset GQE_GEMM_BIG_MIN_UNITS; export GQE_GEMM_KG=$kg; fi
  python bench.py --only-main --batch-size 8192 --steps 20 --warmup 5 > gpurun_out/r3_b10.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r3_b10.json"))
print("kg=$kg B=8192", d["value"], d["ms_per_step"], {k:(v["avg_launch_ms"], v.get("mfma_TFs")) for k,v in d["kernels"].items()})
PY
done
