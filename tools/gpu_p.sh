#!/bin/bash
python -m pytest tests/test_gpu_gemm_bf16x3.py -q -m gpu 2>&1 | tail -3
