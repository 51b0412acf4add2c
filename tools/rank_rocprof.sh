#!/bin/bash
# One rank of a torch.distributed.run launch under its OWN rocprofv3 (run with --no-python): a profiler wrapped around the
# launcher itself hangs at teardown.  Usage: torch.distributed.run --no-python ... tools/rank_rocprof.sh <outdir> bench.py args...
out=$1; shift
cd /tmp; export TMPDIR=/tmp
exec rocprofv3 --kernel-trace -d "$out/rank${RANK}" -o run -- python "$GRAFT_REPO_ROOT/$1" "${@:2}"
