#!/bin/bash
# round 5, first evidence pass: cfg 4's layout (8 ranks) on one GPU + SQ counters over the PPO iteration at HEAD
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
T=${T:-r05a}
TRL_BENCH_DEVICE_MAP=0,0,0,0,0,0,0,0 timeout 600 python bench.py --gpus 8 --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > $O/${T}_bench_8ranks_one_gpu.json 2> $O/${T}_bench_8ranks.log
echo "8 ranks rc=$?"; tail -1 $O/${T}_bench_8ranks_one_gpu.json | cut -c1-600; tail -5 $O/${T}_bench_8ranks.log
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"
B3="python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-secondary"
rm -rf $O/pmc_ppo; timeout 600 rocprofv3 --pmc $CNT --output-format csv -d $O/pmc_ppo -- $B3 > /dev/null 2>&1
python tools/summarize_pmc.py $(find $O/pmc_ppo -name "*counter_collection.csv") > $O/${T}_ppo_pmc_per_kernel_mean.csv
cat $O/${T}_ppo_pmc_per_kernel_mean.csv | cut -c1-250
rm -rf $O/pmc_ppo
