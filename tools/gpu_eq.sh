#!/bin/bash
# GPU check of the EQ row: tests, then timing of the cascade kernel on a cfg-4 shaped batch.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_eq.py -x -q -m gpu 2>&1 | tail -25
timeout 300 python tools/eq_probe.py 2>&1 | tail -10
