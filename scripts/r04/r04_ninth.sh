#!/bin/bash
# round 4, ninth GPU call: multi-device tests, bench.py under torch.distributed.run with one rank and forced collectives,
# bench.py plain, SQ counters of the sweep kernels
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04l
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_multi_device.py tests/test_gpu_fb_sweep.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest.log
OCS_FORCE_COLLECTIVES=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_forced.json 2> $OUT/bench_forced.err || tail -20 $OUT/bench_forced.err
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err || tail -20 $OUT/bench.err
python - <<'PY'
import json
for n in ("bench_forced", "bench"):
    d=json.load(open(f"gpurun_out/r04l/{n}.json"))
    print(n, "headline", d["ms_per_step"], d["collectives_executed"], d["roofline"]["frac"], "fb", d["fb_sweep"]["batch_sweeps_per_s"])
    for k,v in d["other_configs"].items():
        if "rotating" in k: print("  rotating", v["ms_per_pass_pair"], v["roofline"]["frac"], v["ms_per_pass_pair_spread"]["min"], v["ms_per_pass_pair_spread"]["max"])
PY
bash scripts/profile_sq.sh r04fbs scripts/fbs_prof.py > $OUT/sq_fbs.log 2>&1; tail -2 $OUT/sq_fbs.log
rm -rf $ROOT/gpurun_out/sq_r04fbs/*/runc/*kernel_trace.csv
