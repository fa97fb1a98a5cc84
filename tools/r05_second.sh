#!/bin/bash
# round 5, second pass: new parity tests (a21 alias, NaN keys, a17 per-element bound), the 8-rank layout on one GPU, a steady-state timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
T=${T:-r05b}
timeout 600 python -m pytest tests/test_host_env_gpu.py -x -q 2>&1 | tail -5
timeout 600 python -m pytest tests/test_generic_shapes_gpu.py tests/test_fullsize_offpolicy_gpu.py -x -q -k "fused_ppo_update or cfg3_update" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_dist_gpu.py -x -q -k "eight or abi_collectives" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_bench_multirank_gpu.py -x -q -k "cfg4 or two_ranks_over" 2>&1 | tail -8
TRL_BENCH_DEVICE_MAP=0,0,0,0,0,0,0,0 timeout 400 python bench.py --gpus 8 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/${T}_bench_8ranks_one_gpu.json 2> $O/${T}_bench_8ranks.log
echo "8 ranks rc=$?"; tail -1 $O/${T}_bench_8ranks_one_gpu.json | cut -c1-1500
rm -rf $O/prof_t; timeout 400 rocprofv3 --kernel-trace -d $O/prof_t -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/ktimeline.py $(find $O/prof_t -name "*.db" | head -1) 140 > $O/${T}_ppo_iteration_timeline_reference_noise.csv
rm -rf $O/prof_t
grep -v "ppo_grad_wave\|ppo_reduce_adam" $O/${T}_ppo_iteration_timeline_reference_noise.csv | cut -c1-120 | tail -30
cp $O/parity_errors.json $O/${T}_parity_errors.json 2>/dev/null
