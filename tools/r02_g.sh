#!/bin/bash
O=gpurun_out/r02g
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1
tail -40 $O/gpu_tests.log
