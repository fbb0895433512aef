mkdir -p gpurun_out/s2h
for pf in "GQE_LAZY_SLICES=1" "GQE_LAZY_SLICES=2"; do
 echo "--- lazy defer $pf" >> gpurun_out/s2h/lazy.txt
 env STEP_PROBE_TIMING=1 $pf python tools/probes/step_probe.py --lazy --defer 2>&1 | grep -v amdgpu >> gpurun_out/s2h/lazy.txt
 env $pf python tools/probes/step_probe.py --lazy --defer --workload reddit-synth --dim 256 2>&1 | grep -v amdgpu >> gpurun_out/s2h/lazy.txt
done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py -x -q -k "lazy" > gpurun_out/s2h/tests.log 2>&1; tail -3 gpurun_out/s2h/tests.log >> gpurun_out/s2h/lazy.txt
