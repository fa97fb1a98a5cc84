#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
T=${T:-r05e}
timeout 120 python tools/ab_convdx.py 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items(): print(k, {a:b for a,b in v.items() if a.endswith('_us') or not b})"
timeout 600 python -m pytest tests/test_dqn_gpu.py -x -q 2>&1 | tail -3
timeout 300 python -m pytest tests/test_fullsize_offpolicy_gpu.py -x -q -k cfg5 2>&1 | tail -2
rm -rf $O/prof_dqn; timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_dqn -- python tools/bench_dqn.py --epochs 4 > /dev/null 2>&1
python tools/kstats.py $(find $O/prof_dqn -name "*.db" | head -1) > $O/${T}_dqn_kernel_stats.csv
grep -E "conv_dx" $O/${T}_dqn_kernel_stats.csv | cut -c1-160
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"
rm -rf $O/pmc_dqn; timeout 300 rocprofv3 --pmc $CNT --output-format csv -d $O/pmc_dqn -- python tools/bench_dqn.py --epochs 4 > /dev/null 2>&1
python tools/summarize_pmc.py $(find $O/pmc_dqn -name "*counter_collection.csv") > $O/${T}_dqn_pmc_per_kernel_mean.csv
grep -E "conv_dx|kernel," $O/${T}_dqn_pmc_per_kernel_mean.csv | cut -c1-200
rm -rf $O/prof_dqn $O/pmc_dqn
for i in 1 2; do
TRL_DX_CLASS_FORM=0 timeout 200 python tools/bench_dqn.py --epochs 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('image form', d['ms_per_update'])"
TRL_DX_CLASS_FORM=1 timeout 200 python tools/bench_dqn.py --epochs 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('class form', d['ms_per_update'])"
done
