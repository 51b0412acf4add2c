#!/bin/bash
mkdir -p gpurun_out/b
python -m pytest tests/test_gpu_rows_loopback.py tests/test_gpu_rank_oracle.py tests/test_gpu_general.py tests/test_gpu_rows.py tests/test_gpu_rccl_world1.py -m gpu -q -s > gpurun_out/b/tests.log 2>&1
echo "tests rc $?" >> gpurun_out/b/tests.log
grep -E "max \|device|passed|failed|rc " gpurun_out/b/tests.log | tail -40
