#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (the reference itself cannot be
built or imported here -- SURVEY.md §8c -- so these vectors pin the oracle against
regressions and give the GPU tests a committed target; they are NOT reference
outputs: parity stays "unpinned by the reference's own fixtures").

  python tests/make_golden.py        # rewrites tests/golden/
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
synth = importlib.import_module("pointcloud-slam_amd.synth")
from oracle import Oracle  # noqa: E402
from oracle.loader import result_T  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def p2plane_case(seed, n_scan, m_map, optimizer):
    p = synth.make_pair(seed, n_scan, m_map)
    o = Oracle("P2PLANE", optimizer, voxel_resolution=0.5, num_neighbors=27)
    o.set_input_target(p.submap)
    o.set_input_source(p.scan)
    o.enable_trace(128)
    r = o.align(p.guess)
    tr = o.trace()
    return dict(seed=seed, n_scan=n_scan, m_map=m_map, optimizer=optimizer, guess=p.guess, T_gt=p.T_gt, T=result_T(r),
                iterations=r.iterations, converged=r.converged, num_inliers=r.num_inliers, num_linearize=r.num_linearize,
                num_compute_error=r.num_compute_error, H=np.array(r.H[:]).reshape(6, 6), trace=tr,
                scan_head=p.scan[:8].copy(), submap_head=p.submap[:8].copy())


def main():
    os.makedirs(OUT, exist_ok=True)
    cases = {}
    for seed in (0, 1, 2):           # three seeds of BASELINE config 1 (10k-pt scan vs 100k-pt submap)
        for opt in ("GN", "LM"):
            cases["p2plane_s%d_%s" % (seed, opt)] = p2plane_case(seed, 10000, 100000, opt)
    flat = {}
    for name, c in cases.items():
        for k, v in c.items():
            flat[name + "/" + k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "p2plane_config1.npz"), **flat)
    # corner KAT: exact pose known analytically
    sc, sm, T = synth.corner_scene(2000, 30000, seed=5)
    np.savez_compressed(os.path.join(OUT, "corner_kat.npz"), scan=sc, submap=sm, T=T)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
