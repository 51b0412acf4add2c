#!/bin/bash
# round 5: what the advance launch costs without its sampler riders (per-kernel-class times of the headline step)
R=$(pwd)
for v in "X=1" "MKB_BENCH_NO_RIDE=1" "MKB_BENCH_NO_RIDE=1 MKB_BENCH_NO_DRAW_AHEAD=1"; do echo "== $v"; env $v python tools/kbench.py _one; done
