mkdir -p gpurun_out/s2b
for cfg in "96 4" "64 4" "64 8" "64 12" "48 12" "80 6"; do set -- $cfg
 echo "=== LEAD=$1 SHARE=$2" >> gpurun_out/s2b/timeline.txt
 GQE_SPLIT_LEAD=$1 GQE_SPLIT_SHARE=$2 python tools/probes/split_timeline.py >> gpurun_out/s2b/timeline.txt 2>&1
 echo "--- step_probe LEAD=$1 SHARE=$2" >> gpurun_out/s2b/probe.txt
 GQE_SPLIT_LEAD=$1 GQE_SPLIT_SHARE=$2 python tools/probes/step_probe.py --train-step 2>&1 | tail -3 >> gpurun_out/s2b/probe.txt
done
