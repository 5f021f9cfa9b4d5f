"""How much does the ROW ORDER of the five neighbours handed to esti_plane matter?

The reference takes whatever libstdc++'s std::nth_element leaves (ivox3d.h:173-178, ivox3d_node.hpp:176-181); the HIP kernels
and the oracle's default mode sort ascending.  Both orders hold the same neighbour SET; the float ColPivHouseholderQR of a
row-permuted matrix rounds differently, which flips marginal `|n.p + d| > 0.1` verdicts and moves the pose.  This script runs
the CPU oracle under both orders on BASELINE config 1 (3 seeds, GN and LM) and, with --full, on one config-2 pair, and writes
the spread to profiles/.  CPU only (oracle = test infrastructure)."""
import argparse, json, os, sys, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import Oracle
from oracle.loader import result_T

ap = argparse.ArgumentParser()
ap.add_argument("--full", type=int, default=1, help="number of full-size (100k / 1M) pairs")
ap.add_argument("--out", default="profiles/r03_knn_order_sensitivity.json")
a = ap.parse_args()
synth = importlib.import_module("pointcloud-slam_amd.synth")
threads = len(os.sched_getaffinity(0))
cases = [("config1_seed%d" % s, s, 10000, 100000) for s in (0, 1, 2)] + [("config2_pair%d" % s, s, 100000, 1000000) for s in range(a.full)]
rows = []
for name, seed, ns, nm in cases:
    p = synth.make_pair(seed, ns, nm)
    for opt in ("GN", "LM"):
        out = {}
        for order in ("ascending", "libstdcxx"):
            o = Oracle("P2PLANE", opt, voxel_resolution=0.5, num_neighbors=27, num_threads=threads)
            o.set_knn_order(order)
            o.set_input_target(p.submap); o.set_input_source(p.scan)
            c, H, b = o.linearize(p.guess.astype(np.float64))
            pl, sel = o.get_planes(len(p.scan))
            inl0 = o.num_inliers
            t0 = time.perf_counter()
            r = o.align(p.guess)
            out[order] = dict(T=result_T(r), it=r.iterations, inl=r.num_inliers, conv=bool(r.converged), H0=H, b0=b, c0=c, inl0=inl0, pl=pl, sel=sel, s=time.perf_counter() - t0)
        A, B = out["ascending"], out["libstdcxx"]
        D = np.linalg.inv(A["T"]) @ B["T"]
        both = A["sel"] & B["sel"]
        row = dict(case=name, optimizer=opt, scan=ns, map=nm,
                   dt_m=float(np.linalg.norm(D[:3, 3])), dR=float(np.linalg.norm(D[:3, :3] - np.eye(3))),
                   iterations=[A["it"], B["it"]], converged=[A["conv"], B["conv"]], final_inliers=[A["inl"], B["inl"]],
                   first_linearize=dict(inliers=[A["inl0"], B["inl0"]], selected_differs=int(np.sum(A["sel"] != B["sel"])),
                                        planes_bitwise_different=int(np.sum(np.any(A["pl"][both] != B["pl"][both], axis=1))), planes_compared=int(both.sum()),
                                        max_plane_abs_diff=float(np.max(np.abs(A["pl"][both] - B["pl"][both]))) if both.any() else 0.0,
                                        rel_dH=float(np.linalg.norm(A["H0"] - B["H0"]) / np.linalg.norm(A["H0"])), rel_db=float(np.linalg.norm(A["b0"] - B["b0"]) / np.linalg.norm(A["b0"]))))
        rows.append(row)
        print(json.dumps(row), flush=True)
summary = dict(what="oracle P2PLANE under ORC_KNN_ORDER_ASCENDING vs ORC_KNN_ORDER_LIBSTDCXX (std::nth_element of g++ %s libstdc++)" % os.popen("g++ -dumpversion").read().strip(),
               max_dt_m=max(r["dt_m"] for r in rows), max_dR=max(r["dR"] for r in rows),
               max_inlier_delta=max(abs(r["final_inliers"][0] - r["final_inliers"][1]) for r in rows),
               iteration_changes=[(r["case"], r["optimizer"], r["iterations"]) for r in rows if r["iterations"][0] != r["iterations"][1]],
               tolerance_m=1e-4, rows=rows)
json.dump(summary, open(a.out, "w"), indent=1)
print("max dt %.3g m, max dR %.3g" % (summary["max_dt_m"], summary["max_dR"]))
