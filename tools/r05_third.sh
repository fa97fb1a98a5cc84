#!/bin/bash
# round 5, third pass: conv_dx image-tile form A/B + the conv / DQN parity tests on it + DQN bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
T=${T:-r05c}
timeout 300 python tools/ab_convdx.py 2>&1 | tail -4 | tee $O/${T}_ab_convdx.txt
timeout 600 python -m pytest tests/test_dqn_gpu.py -x -q 2>&1 | tail -4
timeout 300 python -m pytest tests/test_fullsize_offpolicy_gpu.py -x -q -k cfg5 2>&1 | tail -3
timeout 200 python tools/bench_dqn.py --epochs 8 2>/dev/null | tail -1 | tee $O/${T}_dqn_bench.json | cut -c1-400
timeout 200 python tools/bench_dqn.py --epochs 8 --quantiles 200 2>/dev/null | tail -1 | tee $O/${T}_qrdqn_bench.json | cut -c1-400
TRL_DX_CLASS_FORM=1 timeout 200 python tools/bench_dqn.py --epochs 8 2>/dev/null | tail -1 | cut -c1-300
timeout 120 python tools/bench_noise.py 2>&1 | tail -1 | tee $O/${T}_bench_noise.json
