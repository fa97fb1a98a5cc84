#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_dqn_gpu.py tests/test_fullsize_offpolicy_gpu.py tests/test_deferred_updates_gpu.py tests/test_frame_dedup_gpu.py tests/test_dist_gpu.py -x -q -k "not eight and not abi" 2>&1 | tail -3
for i in 1 2 3; do
timeout 200 python tools/bench_dqn.py --epochs 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dqn', d['ms_per_update'], d['ms_per_vector_step'])"
done
timeout 200 python tools/bench_dqn.py --epochs 8 --quantiles 200 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('qrdqn', d['ms_per_update'])"
rm -rf $O/prof_dqn; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_dqn -- python tools/bench_dqn.py --epochs 4 > /dev/null 2>&1
python tools/kstats.py $(find $O/prof_dqn -name "*.db" | head -1) | grep -E "conv1_" | cut -c1-140
rm -rf $O/prof_dqn
