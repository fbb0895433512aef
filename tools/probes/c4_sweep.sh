#!/bin/bash
# A / B sweep of the split step on BASELINE config 4 (full Bilinear decoder, d = 128): lead riders x share x queries per pair-GEMM unit.
# usage: tools/probes/c4_sweep.sh "lead:share:kmul ..."   (kmul 0 = default)
cd "$(dirname "$0")/../.."
for cfg in $1; do
  IFS=: read lead share kmul <<< "$cfg"
  echo -n "lead=$lead share=$share kmul=$kmul :: "
  extra=""
  if [ "$kmul" != "0" ]; then extra="GQE_DEBUG_GEMM_KMUL=$kmul"; fi
  env GQE_SPLIT_LEAD=$lead GQE_SPLIT_SHARE=$share $extra STEP_PROBE_TIMING=1 python tools/probes/step_probe.py --train-step --decoder bilinear 2>&1 | grep -v amdgpu.ids | tr '\n' ' '
  echo
done
