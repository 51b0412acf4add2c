#!/bin/bash
python -m pytest tests/test_gpu_bench_multi.py -q -m gpu -k "median" 2>&1 | tail -5
