import sys, numpy as np, importlib
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import pointcloud_slam_amd as pcm
synth = importlib.import_module("pointcloud-slam_amd.synth")
from oracle import Oracle
p = synth.make_pair(0, 10000, 100000)
o = Oracle("P2PLANE","GN", voxel_resolution=0.5, num_neighbors=27); o.set_input_target(p.submap); o.set_input_source(p.scan)
g = pcm.P2PlaneRegistration(0, optimizer="GN", voxel_resolution=0.5, num_neighbors=27); g.set_input_target(p.submap); g.set_input_source(p.scan)
T = p.guess.astype(np.float64)
c0,H0,b0 = o.linearize(T); c1,H1,b1,inl = g.evaluate_cost(T)
po, so = o.get_planes(10000); pg = g.get_planes(10000); sg = ~np.isnan(pg[:,0])
print("sel equal", np.array_equal(so, sg), so.sum(), sg.sum())
both = so & sg
d = np.abs(po[both]-pg[both])
print("max plane diff", d.max(0), "n bitexact", (d.max(1)==0).sum(), "of", both.sum())
bad = np.nonzero(d.max(1)>0)[0][:5]
for i in bad: print(po[both][i], pg[both][i])
print("cost", c0, c1, "b", b0, b1)
