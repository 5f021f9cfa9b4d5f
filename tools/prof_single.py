"""Small profiling driver: a few batched aligns of N pairs on ONE stream (no torch.distributed, no fork;
safe as the program behind `rocprofv3 ... --`).  --cache loads pairs written by tools/gen_cache.py."""
import sys, os, importlib, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=4)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--scan", type=int, default=100000)
ap.add_argument("--map", type=int, default=1000000)
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--cache", default="")
ap.add_argument("--lib", default="", help="A/B: load this build of the library instead of the in-tree one")
ap.add_argument("--phases", type=int, default=1)
a = ap.parse_args()
if a.cache:
    z = np.load(a.cache)
    n = min(int(z["n"]), a.pairs)
    scans = [z["scan%d" % i] for i in range(n)]; maps = [z["map%d" % i] for i in range(n)]; guesses = [z["guess%d" % i] for i in range(n)]
else:
    synth = importlib.import_module("pointcloud-slam_amd.synth")
    ps = [synth.make_pair(i, a.scan, a.map) for i in range(a.pairs)]
    scans = [p.scan for p in ps]; maps = [p.submap for p in ps]; guesses = [p.guess for p in ps]
import pointcloud_slam_amd as pcm
if a.lib:
    pcm.capi.library_path = lambda: os.path.abspath(a.lib)
regs = []
for s, m in zip(scans, maps):
    r = pcm.P2PlaneRegistration(0, optimizer="GN", voxel_resolution=0.5, num_neighbors=27, flags=a.flags)
    r.set_input_target(m); r.set_input_source(s); regs.append(r)
g = np.stack(guesses)
res = pcm.align_batch(regs, g)
t0 = time.perf_counter()
for s in range(a.steps):
    res = pcm.align_batch(regs, g)
dt = (time.perf_counter() - t0) / max(1, a.steps)
print("linearize passes", [r.num_linearize for r in res], "sum", sum(r.num_linearize for r in res))
print("ms per batched align (one stream, scans already ordered): %.3f  -> %.0f reg/s" % (dt * 1e3, len(regs) / dt))
regs[0].reset_stats(); regs[0].set_profiling(2)
res = pcm.align_batch(regs, g)
st = regs[0].stats(); regs[0].set_profiling(0)
print('candidates/point %.2f  tiles staged %.3f' % (st['candidates'] / max(1, st['point_passes']), st['tiles_lds_points'] / max(1, st['tiles'])))
if a.phases:
    regs[0].set_profiling(4)
    res = pcm.align_batch(regs, g)
    pc = regs[0].phase_cycles()
    n = max(1, pc[7])
    names = ['load+box', 'probe', 'stage+grid', '-', 'search', 'fit', 'jobs+residual+reduce'] if (a.flags & 8) else ['load+box', 'probe', 'stage', 'cellgrid', 'search', 'fit', 'jobs+residual+reduce']
    print('tiles', pc[7]); print({k: round(v / n) for k, v in zip(names, pc[:7])}, 'ticks/tile')
