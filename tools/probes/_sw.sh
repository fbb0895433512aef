mkdir -p gpurun_out/s2d
run() { echo "--- $*" >> gpurun_out/s2d/sweep.txt; env $1 python tools/probes/step_probe.py --train-step ${@:2} 2>&1 | grep "us/step" >> gpurun_out/s2d/sweep.txt; }
run "X=1" --batch 8192
run "GQE_NO_LEAN=1" --batch 8192
for L in 0 96 256; do for T in 256 512; do run "GQE_SPLIT_MANY_TILES=1 GQE_SPLIT_SHAPE=8 GQE_SPLIT_LEAD=$L GQE_SPLIT_TAIL=$T" --batch 8192; done; done
run "GQE_SPLIT_MANY_TILES=1 GQE_SPLIT_LEAD=96" --batch 8192
run "X=1" --batch 2048
run "GQE_SPLIT_MANY_TILES=1 GQE_SPLIT_SHAPE=8 GQE_SPLIT_LEAD=96" --batch 2048
run "GQE_SPLIT_MANY_TILES=1 GQE_SPLIT_LEAD=96" --batch 2048
run "X=1" --batch 1024
run "GQE_SPLIT_MANY_TILES=1 GQE_SPLIT_LEAD=96" --batch 1024
run "GQE_SPLIT_MANY_TILES=1 GQE_SPLIT_LEAD=64" --batch 1024
