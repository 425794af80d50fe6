#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 || exit 1
SEED=41 timeout -k 10 600 python tests/stress_nlp.py 600 2>&1 | tail -1
for NS in 2 3 4; do
  NS=$NS BATCHES=4096,8192,16384,32768 MODES=auto timeout -k 10 600 python scripts/bl4_time.py 2>&1 | grep "batch" | cut -c1-70 | sed "s/^/nS $NS /"
done
BATCHES=4096,8192,65536 MODES=auto timeout -k 10 600 python scripts/bl4_time.py 2>&1 | grep "batch" | cut -c1-70 | sed "s/^/BL-4 /"
