mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rccl.py -k "pair_gemm or eight_wave or rccl or feeder" -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r3_tests10.log 2>&1; echo "rc $?" >> gpurun_out/r3_tests10.log)
tail -6 gpurun_out/r3_tests10.log | cut -c1-250
for kg in 0 1 2 4; do
  if [ $kg = 0 ]; then export GQE_GEMM_BIG_MIN_UNITS=100000000; unset GQE_GEMM_KG; else unset GQE_GEMM_BIG_MIN_UNITS; export GQE_GEMM_KG=$kg; fi
  python bench.py --only-main --batch-size 8192 --steps 20 --warmup 5 > gpurun_out/r3_b10.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r3_b10.json"))
print("kg=$kg B=8192", d["value"], d["ms_per_step"], {k:(v["avg_launch_ms"], v.get("mfma_TFs")) for k,v in d["kernels"].items()})
PY
done
unset GQE_GEMM_BIG_MIN_UNITS GQE_GEMM_KG
(GQE_SHARD_PROFILE=1 timeout 300 python tools/shard_overhead_bench.py > gpurun_out/r3_shard_overhead.log 2>&1)
grep "us/step\|host time" gpurun_out/r3_shard_overhead.log
python bench.py --only-main --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('main', d['value'], d['ms_per_step'])"
python - <<'PY'
import subprocess, json, sys
out = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-configs", "--no-reddit", "--no-lazy", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True).stdout
d = json.loads(out)
h = d["host_fed"]; print("host_fed", d["value"], h["value"], h["pinned_hipMemcpyAsync"]["value"], h["pinned_hipMemcpyAsync"]["timing"], h["lazy_exact_adam"]["value"])
PY
