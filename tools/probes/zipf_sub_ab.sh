#!/bin/bash
# A / B of the hot word rows' sub-lists (GQE_HOT_SUB, include/gqe.h) on reddit-synth with Zipf(1) words: the tests that reach the
# path, then the step probe with the library's event brackets (0 = fused launch + gather, 2 = optimiser pass) -> stdout
cd "$(dirname "$0")/../.."
python -m pytest tests/test_gpu_parity.py -q -x -k "zipf or lazy_adam_with_a_bag or embedding_bag" -p no:cacheprovider 2>&1 | tail -2
python -m pytest tests/test_gpu_limits.py tests/test_gpu_split.py -q -x -k "hot_rows or bag" -p no:cacheprovider 2>&1 | tail -2
for sub in 1 0; do
  echo "== GQE_HOT_SUB=$sub, Zipf(1) words"
  GQE_HOT_SUB=$sub STEP_PROBE_TIMING=1 python tools/probes/step_probe.py --workload reddit-synth --dim 256 --zipf 1.0 2>&1 | tail -3
done
echo "== uniform words"
STEP_PROBE_TIMING=1 python tools/probes/step_probe.py --workload reddit-synth --dim 256 2>&1 | tail -2
