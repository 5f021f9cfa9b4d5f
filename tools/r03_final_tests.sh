#!/bin/bash
# Final check, part A: __graft_entry__.smoke and the whole GPU suite.  Outputs under gpurun_out/r03final.
set -e
export TMPDIR=/tmp
O=gpurun_out/r03final
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
