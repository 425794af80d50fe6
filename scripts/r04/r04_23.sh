#!/bin/bash
# the N-rank code paths of bench.py on a one-GPU box: 2 and 3 ranks over gloo sharing the GPU (numbers meaningless)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_rehearsal
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_fb_sweep.py -m gpu -q -x -k "interpolant" 2>&1 | tail -2
for n in 2 3; do
  OCS_BENCH_REHEARSAL=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_n$n.json 2> $OUT/bench_n$n.err || { echo "n=$n FAILED"; tail -30 $OUT/bench_n$n.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$OUT/bench_n$n.json").read().strip().splitlines()[-1])
print("n=$n ok:", d["n_gpus"], d.get("rehearsal"), d["config"]["batch_total"], round(d["ms_per_step"],4), list(d.get("other_configs",{}).keys()))
PY
done
OCS_BENCH_REHEARSAL=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --scaling strong > $OUT/bench_n2_strong.json 2> $OUT/bench_n2_strong.err || { echo "strong FAILED"; tail -30 $OUT/bench_n2_strong.err; exit 1; }
tail -c 300 $OUT/bench_n2_strong.json
