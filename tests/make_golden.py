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


MODEL_CASES = [   # (name, oracle model, optimizer, oracle kwargs, dense submap?)
    ("gicp_lm", "GICP", "LM", dict(voxel_resolution=0.5, max_corr_dist=2.0), False),
    ("vgicp_d7_gn", "VGICP", "GN", dict(voxel_resolution=1.0, num_neighbors=7, max_iterations=12), False),
    ("vgicp_mult_lm", "VGICP", "LM", dict(voxel_resolution=1.0, num_neighbors=1, voxel_mode=2, regularization="MIN_EIG", max_iterations=12), False),
    ("vgicp_cuda_lm", "VGICP_CUDA", "LM", dict(voxel_resolution=1.0, num_neighbors=1, max_iterations=12), True),
    ("ndt_d2d_lm", "NDT_D2D", "LM", dict(voxel_resolution=1.0, num_neighbors=7), True),
    ("ndt_p2d_gn", "NDT_P2D", "GN", dict(voxel_resolution=1.0, num_neighbors=1, max_iterations=12), True),
    ("ndt_omp_d7", "NDT_OMP", "LM", dict(voxel_resolution=1.0, num_neighbors=7, translation_eps=0.1, max_iterations=35), True),
    ("ndt_omp_kdtree", "NDT_OMP", "LM", dict(voxel_resolution=1.0, num_neighbors=0, translation_eps=0.01, max_iterations=35), True),
]


def model_case(name, model, optimizer, kw, dense):
    p = synth.make_pair(7, 4000, 40000, density=60.0 if dense else 8.0)
    o = Oracle(model, optimizer, **kw)
    o.set_input_target(p.submap)
    o.set_input_source(p.scan)
    r = o.align(p.guess)
    return dict(T=result_T(r), iterations=r.iterations, converged=r.converged, num_linearize=r.num_linearize, num_compute_error=r.num_compute_error,
                cost=r.cost, H=np.array(r.H[:]).reshape(6, 6))


def gicp_bfgs_case():
    """pclomp GICP-BFGS functor (oracle/orc_gicp_bfgs.c): a small correspondence set with its inputs, f and g in the three modes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gicp_bfgs import _problem
    from oracle import loader as L
    src, tgt, isrc, itgt, maha, base, x = _problem(21, n=400, m=300, stride=4, base_translation=True)
    out = dict(src=src, tgt=tgt, idx_src=isrc, idx_tgt=itgt, maha=maha, base=base, x=x, T=L.gicp_bfgs_apply_state(base, x))
    for mode in (0, 1, 2):
        f, g = L.gicp_bfgs_fdf(src, tgt, isrc, itgt, maha, base, x, mode)
        out["f%d" % mode] = f
        out["g%d" % mode] = g
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "gicp_bfgs_functor.npz"), **gicp_bfgs_case())
    cases = {}
    for seed in (0, 1, 2):           # three seeds of BASELINE config 1 (10k-pt scan vs 100k-pt submap)
        for opt in ("GN", "LM"):
            cases["p2plane_s%d_%s" % (seed, opt)] = p2plane_case(seed, 10000, 100000, opt)
    flat = {}
    for name, c in cases.items():
        for k, v in c.items():
            flat[name + "/" + k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "p2plane_config1.npz"), **flat)
    flat = {}
    for name, model, opt, kw, dense in MODEL_CASES:
        for k, v in model_case(name, model, opt, kw, dense).items():
            flat[name + "/" + k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, "models_small.npz"), **flat)
    # corner KAT: exact pose known analytically
    sc, sm, T = synth.corner_scene(2000, 30000, seed=5)
    np.savez_compressed(os.path.join(OUT, "corner_kat.npz"), scan=sc, submap=sm, T=T)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
