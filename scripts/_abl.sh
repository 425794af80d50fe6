python -m pytest tests/test_gpu_rk4_parity.py -x -q -m gpu -k "scan or mappings or bl2" 2>&1 | tail -3
for a in 0 1 4; do echo "ABL=$a remap"; OCS_SCAN_ABL=$a BATCHES=4096,16384 python scripts/bench_passes.py 2>&1 | grep "nS=4"; done
for a in 0; do echo "ABL=$a noremap"; OCS_SCAN_NOREMAP=1 OCS_SCAN_ABL=$a BATCHES=4096,16384 python scripts/bench_passes.py 2>&1 | grep "nS=4"; done
