#!/bin/bash
# Long runs of the headline loop in both launch-sequence modes: no hang, no stall, iteration times stay flat.
#   gpurun -- 'bash tools/soak.sh [steps]'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
STEPS=${1:-3000}
for chains in two joint; do
  TRL_PPO_CHAINS=$chains timeout 600 python bench.py --steps "$STEPS" --warmup 3 --no-cpu-baseline --no-secondary 2>gpurun_out/soak_$chains.log | tail -1 > gpurun_out/soak_$chains.json
  echo "exit $? ($chains)"
  python - "$chains" <<'PY'
import json, sys
d = json.load(open("gpurun_out/soak_%s.json" % sys.argv[1]))
print("%s: %d iterations, parity %.4f ms, device noise %.4f ms, noise_blocks %s" % (sys.argv[1], d["steps"], d["ms_per_step"], d["device_noise_ms_per_step"], d["config"]["noise_blocks"]), flush=True)
PY
done
