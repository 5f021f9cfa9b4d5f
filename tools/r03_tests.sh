#!/bin/bash
# One gpurun call: the GPU test files named on the command line (default: all), stop at the first failure, tail of the log.
export TMPDIR=/tmp
O=gpurun_out/r03tests
mkdir -p $O
timeout -k 10 1100 python -m pytest ${@:-tests} -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -25 $O/tests.log
