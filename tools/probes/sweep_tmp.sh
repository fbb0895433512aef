run() { echo "$1: $(env $1 python tools/probes/step_probe.py --train-step 2>&1 | tail -1 | sed 's/bio-synth d=128 bilinear-diag B=512: //; s/ (median.*//')"; }
B="GQE_SPLIT_STOP=0 GQE_SPLIT_LEAD=64 GQE_SPLIT_TAIL=64"
run "$B"
run "$B GQE_SPLIT_DEBUG_B=3"
run "$B GQE_SPLIT_DEBUG_NORIDE=1"
run "$B GQE_SPLIT_DEBUG_NORIDE=1 GQE_SPLIT_DEBUG_B=3"
run "GQE_SPLIT_STOP=0 GQE_SPLIT_LEAD=0 GQE_SPLIT_TAIL=64 GQE_SPLIT_DEBUG_NORIDE=1 GQE_SPLIT_DEBUG_B=3"
