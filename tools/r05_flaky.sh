#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for i in 1 2; do ( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 ) 2>&1 | grep -E "passed|failed|error|real"; done
