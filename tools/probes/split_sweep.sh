#!/bin/bash
# A / B sweep of the split step's launch shape on one box: GQE_SPLIT_SHAPE (tile waves) x GQE_SPLIT_LEAD x GQE_SPLIT_TAIL x GQE_SPLIT_SHARE.
# usage: tools/probes/split_sweep.sh "16:96:256:4 8:96:256:4 ..."   (prints one step_probe line per configuration)
cd "$(dirname "$0")/../.."
for cfg in $1; do
  IFS=: read shape lead tail share <<< "$cfg"
  echo -n "shape=$shape lead=$lead tail=$tail share=$share :: "
  GQE_SPLIT_SHAPE=$shape GQE_SPLIT_LEAD=$lead GQE_SPLIT_TAIL=$tail GQE_SPLIT_SHARE=$share STEP_PROBE_TIMING=1 python tools/probes/step_probe.py --train-step 2>&1 | grep -v amdgpu.ids | tr '\n' ' '
  echo
done
