#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_configs.py -q -m gpu --no-header -p no:cacheprovider --durations=6 2>&1 | tail -30 | tee gpurun_out/cfgtests.log
