#!/bin/bash
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -q -m gpu -s > $O/suite.log 2>&1; tail -3 $O/suite.log
MKB_FUZZ_SEEDS=200 MKB_TEST_DIRTY_MEMORY=3 python -m pytest tests/test_gpu_shape_fuzz.py tests/test_gpu_train_fuzz.py tests/test_gpu_sampler.py -q -k "random or fuzz" > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log
python tools/eval_speed.py > $O/eval_speed.txt 2>&1; tail -8 $O/eval_speed.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
