#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_product_gpu.py tests/test_kernels_gpu.py tests/test_generic_shapes_gpu.py tests/test_fullsize_gpu.py tests/test_obsnorm_gpu.py tests/test_a2c_gpu.py tests/test_noise_prefetch_gpu.py -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d.get('device_noise_ms_per_step'))"; done
rm -rf $O/prof_b; rocprofv3 --kernel-trace --stats -d $O/prof_b -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/kstats.py $(find $O/prof_b -name "*.db" | head -1) | grep -E "value_pass|rollout_kernel|ppo_grad" | cut -c1-120
rm -rf $O/prof_b
