#!/bin/bash
# The end-of-round measurement set (profiles/r06z_*; r05z_* / r04z_* were the same script one and two rounds earlier): run on the MI355X box from the repo root, e.g.
#   gpurun --timeout 2400 -- 'bash tools/measure_round.sh'
# Writes into gpurun_out/ (scratch); copy what is to be kept into profiles/.  Counter passes are runs of their own with no
# tracing domain besides the kernel trace.  T = file prefix.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
T=${T:-r06z}
# 1. the default bench line (cpu baseline, secondary workloads, peaks)
python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.log; tail -1 $O/${T}_bench.json | cut -c1-400
# 2. kernel trace of the same command (short: no cpu baseline / secondary)
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
rm -rf $O/prof_b; rocprofv3 --kernel-trace --stats -d $O/prof_b -- $B > $O/${T}_prof_bench.json 2>/dev/null
python tools/kstats.py $(find $O/prof_b -name "*.db" | head -1) > $O/${T}_bench_kernel_stats.csv
head -12 $O/${T}_bench_kernel_stats.csv | cut -c1-150
# 2b. the same with the two networks in ONE launch per minibatch (TRL_PPO_CHAINS=joint): what the roofline entry times
rm -rf $O/prof_b; TRL_PPO_CHAINS=joint rocprofv3 --kernel-trace --stats -d $O/prof_b -- $B > /dev/null 2>&1
python tools/kstats.py $(find $O/prof_b -name "*.db" | head -1) > $O/${T}_bench_kernel_stats_joint.csv
head -4 $O/${T}_bench_kernel_stats_joint.csv | cut -c1-150
# 3. HBM traffic counters, separate passes; then the SQ counters of the same command (MFMA-busy of the headline kernel)
#    -- joint launches: the counters of the 256-workgroup launch the roofline entry refers to
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
unset TRL_PPO_CHAINS
# 4. cfg 3 / cfg 5 tables and MFMA counters
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"
for w in sac dqn qrdqn; do
  case $w in sac) CMD="python tools/bench_sac.py --epochs 6";; dqn) CMD="python tools/bench_dqn.py --epochs 4";; qrdqn) CMD="python tools/bench_dqn.py --epochs 4 --quantiles 200";; esac
  case $w in sac) LONG="python tools/bench_sac.py --epochs 20";; dqn) LONG="python tools/bench_dqn.py --epochs 12";; qrdqn) LONG="python tools/bench_dqn.py --epochs 12 --quantiles 200";; esac
  $LONG 2>/dev/null | tail -1 > $O/${T}_${w}_bench.json
  rm -rf $O/prof_$w; rocprofv3 --kernel-trace --stats -d $O/prof_$w -- $CMD > /dev/null 2>&1
  python tools/kstats.py $(find $O/prof_$w -name "*.db" | head -1) > $O/${T}_${w}_kernel_stats.csv
  python tools/ktimeline.py $(find $O/prof_$w -name "*.db" | head -1) 120 > $O/${T}_${w}_timeline.csv
  rm -rf $O/pmc_$w; rocprofv3 --pmc $CNT --output-format csv -d $O/pmc_$w -- $CMD > /dev/null 2>&1
  python tools/summarize_pmc.py $(find $O/pmc_$w -name "*counter_collection.csv") > $O/${T}_${w}_pmc_per_kernel_mean.csv
  echo "== $w"; cat $O/${T}_${w}_bench.json | cut -c1-300; head -6 $O/${T}_${w}_kernel_stats.csv | cut -c1-150
done
rm -rf $O/prof_* $O/pmc_*
# 5. two ranks on one GPU (peer transport), then BASELINE cfg 4's layout: eight ranks x 2048 envs on one GPU (all-reduce route)
TRL_BENCH_DEVICE_MAP=0,0 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/${T}_bench_2ranks_one_gpu.json 2> $O/${T}_bench_2ranks.log
tail -1 $O/${T}_bench_2ranks_one_gpu.json | cut -c1-300
TRL_BENCH_DEVICE_MAP=0,0,0,0,0,0,0,0 timeout 400 python bench.py --gpus 8 --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > $O/${T}_bench_8ranks_one_gpu.json 2> $O/${T}_bench_8ranks.log
tail -1 $O/${T}_bench_8ranks_one_gpu.json | cut -c1-300
# 6. PPO iteration timeline (reference noise)
rm -rf $O/prof_t; rocprofv3 --kernel-trace -d $O/prof_t -- python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/ktimeline.py $(find $O/prof_t -name "*.db" | head -1) 100 > $O/${T}_ppo_iteration_timeline_reference_noise.csv
rm -rf $O/prof_t
# 7. the host side of the reference noise stream (jump-ahead pass at 8 ranks), the conv input-gradient A/B
python tools/bench_noise.py 2>/dev/null | tail -1 > $O/${T}_bench_noise.json; cut -c1-600 $O/${T}_bench_noise.json
python tools/ab_convdx.py 2>/dev/null | tail -1 > $O/${T}_ab_convdx.json; cut -c1-300 $O/${T}_ab_convdx.json
# 8. BASELINE.md section 3, literally: 3 + 20 whole CPU iterations (about two minutes of host time)
if [ "${FULL_CPU:-1}" = "1" ]; then python bench.py --cpu-baseline-full > $O/${T}_cpu_baseline_full.json 2> $O/${T}_cpu_baseline_full.log; cut -c1-700 $O/${T}_cpu_baseline_full.json; fi
