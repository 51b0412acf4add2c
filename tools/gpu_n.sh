#!/bin/bash
O=gpurun_out/n; mkdir -p $O
for st in 4 40 120 400; do echo "== traced step ~$st"; python tools/wgtrace.py run bwd1 $st 2 2>/dev/null | head -4; done | tee $O/trace.txt
