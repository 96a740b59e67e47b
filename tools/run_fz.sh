set -o pipefail
IVS_FUZZ_SEEDS=20000:20700 timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -30
