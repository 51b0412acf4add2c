#!/bin/bash
mkdir -p gpurun_out/c
python -m pytest tests/test_gpu_rank_oracle.py tests/test_gpu_dense_pass.py -m gpu -q -x > gpurun_out/c/tests.log 2>&1; tail -2 gpurun_out/c/tests.log
for rep in 1 2; do python tools/kbench.py run base qlate pipe kc32 pipekc32; done 2>&1 | tee gpurun_out/c/kbench.txt
