#!/bin/bash
mkdir -p gpurun_out/o
timeout 1200 python -m pytest tests/test_gpu_rows_loopback.py -q -m gpu -x > gpurun_out/o/tests.log 2>&1; tail -30 gpurun_out/o/tests.log
