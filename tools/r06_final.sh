#!/bin/bash
# round 6, end-of-round: the whole GPU test suite (parity error log), smoke(), then the measurement set
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
rm -f $O/parity_errors.json
( time timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) 2>&1 | tail -8
cp $O/parity_errors.json $O/r06z_parity_errors.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/measure_round.sh 2>&1 | cut -c1-400
