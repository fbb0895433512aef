#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "random_schema_vs_oracle" 2>&1 | grep -v amdgpu | tail -3 > gpurun_out/r38.log
timeout 300 python -c "
from graphqembed_amd.engine import load_library
lib = load_library(); print('supported', lib.gqe_dim_supported(0, 0, 128))
import numpy as np, torch
from graphqembed_amd.engine import Engine, ArenaLayout
L = ArenaLayout(); L.add('enc.a', (100, 64)); L.add('dec.x', (64,))
e = Engine(64, 'bilinear-diag', 'min', L); print('engine ok'); e.close()" >> gpurun_out/r38.log 2>&1
