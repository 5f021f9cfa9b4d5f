#!/bin/bash
# One gpurun call: parity tests of the search kernels, then an A/B on one stream (32 cached pairs) with phase stamps.
export TMPDIR=/tmp
O=gpurun_out/r03ab
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_flat_search.py tests/test_gpu_reforder.py tests/test_gpu_parity.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
python tools/gen_cache.py --pairs 32 > $O/gen.log 2>&1
for rep in 1 2; do
  for f in 8 0 16; do
    timeout -k 10 300 python tools/prof_single.py --pairs 32 --steps 10 --phases 1 --cache /tmp/pcm_pairs.npz --flags $f > $O/f${f}_$rep.log 2>&1
    echo "flags=$f rep=$rep: $(grep 'ms per' $O/f${f}_$rep.log)"; grep "ticks\|memo" $O/f${f}_$rep.log
  done
done
