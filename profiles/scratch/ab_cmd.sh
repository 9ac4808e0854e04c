set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_robustness.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5
for B in 1024 2048 4096; do PYTHONPATH=. timeout -k 10 300 python3 profiles/microbench/ab.py B=$B reps=7 force_general=5 product 2>&1 | grep "ms per solve"; PYTHONPATH=. timeout -k 10 300 python3 profiles/microbench/ab.py B=$B reps=7 force_general=6 product 2>&1 | grep "ms per solve"; done
