mkdir -p gpurun_out
for rep in 1 2 3; do
for t in old new seq; do
  if [ $t = old ]; then D=build/old_tree; else D=.; fi
  if [ $t = seq ]; then export GQE_LIB=$GRAFT_REPO_ROOT/build/ab/seq/libgqe.so; else unset GQE_LIB; fi
  (cd $D && python bench.py --only-main --steps 20 --warmup 5 2>/dev/null) > gpurun_out/r3_ab_$t.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r3_ab_$t.json"))
print("$t", d["value"], d["ms_per_step"], {k:v["avg_launch_ms"] for k,v in d["kernels"].items()}, d["roofline"]["avg_launch_ms"])
PY
done
done
