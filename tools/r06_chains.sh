#!/bin/bash
# round 6: the two update chains -- parity, then A/B of the iteration against the joint sequence
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "one_network or one_launch or ppo_" 2>&1 | tail -4 )
( timeout 1200 python -m pytest tests/test_product_gpu.py tests/test_fullsize_gpu.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -6 )
B="python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-secondary"
for rep in 1 2 3; do
  for mode in joint two; do
    TRL_PPO_CHAINS=$mode timeout 300 $B 2>/dev/null | tail -1 > $O/r06h_bench_${mode}_$rep.json
    python - <<PY
import json; d=json.load(open("$O/r06h_bench_${mode}_$rep.json")); print("$mode", $rep, "ms_per_step %.4f device %.4f parity %.4f avg_launch_us %.2f" % (d["ms_per_step"], d.get("device_noise_ms_per_step",0), d.get("parity_mode_ms_per_step",0), d["roofline"]["avg_launch_us"]))
PY
  done
done
