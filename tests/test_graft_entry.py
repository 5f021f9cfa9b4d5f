"""The driver's build check: __graft_entry__.build() must compile everything for gfx950 and load the library (no GPU needed)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_entry_point_runs_without_a_gpu():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()                      # incremental: make is a no-op when nothing changed
    import pointcloud_slam_amd as pcm
    assert os.path.exists(pcm.library_path())
