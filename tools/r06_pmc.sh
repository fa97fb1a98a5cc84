cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
T=r06z
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
rm -rf $O/prof_b; TRL_PPO_CHAINS=joint rocprofv3 --kernel-trace --stats -d $O/prof_b -- $B > /dev/null 2>&1
python tools/kstats.py $(find $O/prof_b -name "*.db" | head -1) > $O/${T}_bench_kernel_stats_joint.csv
head -4 $O/${T}_bench_kernel_stats_joint.csv | cut -c1-150
export TRL_PPO_CHAINS=joint
B3="python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-secondary"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c; rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- $B3 > /dev/null 2>&1
done
python tools/summarize_pmc.py $(find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*counter_collection.csv") > $O/${T}_pmc_traffic_per_kernel_mean.csv
grep -E "kernel,|ppo_grad|reduce_adam|gae|adv_stats|value_pass|rollout" $O/${T}_pmc_traffic_per_kernel_mean.csv | cut -c1-200
CNT0="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"
rm -rf $O/pmc_ppo; rocprofv3 --pmc $CNT0 --output-format csv -d $O/pmc_ppo -- $B3 > /dev/null 2>&1
python tools/summarize_pmc.py $(find $O/pmc_ppo -name "*counter_collection.csv") > $O/${T}_ppo_pmc_per_kernel_mean.csv
grep -E "kernel,|ppo_grad|reduce_adam" $O/${T}_ppo_pmc_per_kernel_mean.csv | cut -c1-220
rm -rf $O/prof_* $O/pmc_*
