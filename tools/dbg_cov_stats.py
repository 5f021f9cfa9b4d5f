"""Diagnostic (library built with -DPCM_COV_STATS): distribution of kNN candidates / passes per query."""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
synth = importlib.import_module("pointcloud-slam_amd.synth")
p = synth.make_pair(0, 100000, 1000000)
import torch
import pointcloud_slam_amd as pcm
g = pcm.GicpRegistration(0); g.set_input_target(p.submap); g.set_input_source(p.scan)
for tgt in (False, True):
    c = g.get_covariances(tgt)
    cand, npass, ins, exact, rk = c[:, 0, 0], c[:, 0, 1], c[:, 0, 2], c[:, 1, 1], c[:, 1, 2]
    print("target" if tgt else "source", "n", len(c))
    for name, v in (("scanned", cand), ("pass", npass), ("segments", ins), ("r_k", rk)):
        print("  %s mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (name, v.mean(), *np.percentile(v, [50, 90, 99]), v.max()))
    print("  slow-path lanes frac", exact.mean(), " waves with a slow lane frac", (exact.reshape(-1)[:len(exact)//64*64].reshape(-1,64).max(axis=1)).mean())
