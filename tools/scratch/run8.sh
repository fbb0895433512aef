mkdir -p gpurun_out
(timeout 1800 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r3_tests8.log 2>&1; echo "rc $?" >> gpurun_out/r3_tests8.log)
tail -6 gpurun_out/r3_tests8.log | cut -c1-250
(timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench8.json 2> gpurun_out/r3_bench8.err)
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3_bench8.json"))
print("main", d["value"], d["ms_per_step"], {k:v["avg_launch_ms"] for k,v in d["kernels"].items()}, d["roofline"]["avg_launch_ms"])
print("lazy", d["lazy_exact_adam"]["value"], d["lazy_exact_adam"]["ms_per_step"])
h=d["host_fed"]; print("host_fed", h["value"], h["pinned_hipMemcpyAsync"]["value"], h["pinned_hipMemcpyAsync"]["timing"], h["lazy_exact_adam"]["value"])
print("reddit", d["reddit_synth"]["value"], d["reddit_synth"]["ms_per_step"], d["reddit_synth"]["kernels_ms"])
for k,v in d["configs"].items(): print(k, v["value"], v["ms_per_step"], v["kernels_ms"])
PY
for a in "" "--workload reddit-synth" "--decoder bilinear"; do timeout 300 python tools/eval_bench.py $a 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r3_eval_bench.log
