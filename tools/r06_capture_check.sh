#!/bin/bash
# development aid (round 6): the chain graphs are captured inside the warm-up; default bench lines before / after
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_product_gpu.py -x -q -m gpu -k "two_update_chains or graph_replay or chain_error" 2>&1 | tail -3
for i in 1 2 3; do
  python bench.py 2>/dev/null | tail -1 > gpurun_out/r06y_bench_default_$i.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r06y_bench_default_$i.json"))
print("default line $i:", d["ms_per_step"], d["device_noise_ms_per_step"], d["config"]["noise_blocks"])
PY
done
