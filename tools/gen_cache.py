"""Generate the bench pairs once into an .npz (fork pool, no GPU touched) so that the profiling
passes of one gpurun call (each a fresh process under rocprofv3, which must not fork) load them."""
import sys, os, importlib, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multiprocessing as mp

def gen(a):
    synth = importlib.import_module("pointcloud-slam_amd.synth")
    p = synth.make_pair(*a)
    return p.scan, p.submap, p.guess, p.T_gt

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--scan", type=int, default=100000)
    ap.add_argument("--map", type=int, default=1000000)
    ap.add_argument("--out", default="/tmp/pcm_pairs.npz")
    ap.add_argument("--workers", type=int, default=16)
    a = ap.parse_args()
    with mp.get_context("fork").Pool(min(a.workers, a.pairs)) as pool:
        res = pool.map(gen, [(i, a.scan, a.map) for i in range(a.pairs)])
    d = {}
    for i, (s, m, g, t) in enumerate(res):
        d["scan%d" % i] = s; d["map%d" % i] = m; d["guess%d" % i] = g; d["gt%d" % i] = t
    np.savez(a.out, n=np.int64(a.pairs), **d)
    print("wrote", a.out)
