"""One LiDAR frame on device buffers (pcm_lio_frame_begin / pcm_obs_model / pcm_lio_frame_end; LaserMapping::Run,
jueying_lio/src/laser_mapping.cc:323-347, 525-583) against the same operators called one by one through host buffers: the same
scan bit for bit, the same measurement model, the same map update.  ``-m gpu``."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

pytestmark = pytest.mark.gpu


def _frame_msg(synth, scene, T, n, seed):
    """A synthetic Livox frame as the driver delivers it: CustomPoint records, time-ordered, with some noise tags."""
    sc, ex = synth.livox_scan(scene, T, n, seed)
    a = synth.custom_msg(sc, ex)
    rng = np.random.default_rng(seed)
    a["tag"] = np.where(rng.random(n) < 0.03, 0x20, 0x10)
    return a


def _poses(vel):
    K = 11
    poses = np.zeros((K, 22))
    for k in range(K):
        poses[k, 0] = 0.01 * k; poses[k, 4:7] = [0.01, -0.02, 0.05]; poses[k, 7:10] = vel; poses[k, 10:13] = vel * 0.01 * k
        poses[k, 13:22] = Rotation.from_rotvec(np.array([0.01, -0.02, 0.05]) * 0.01 * k).as_matrix().ravel()
    return poses


@pytest.mark.parametrize("ref_semantics", [False, True])
def test_frame_pipeline_equals_the_single_operators(pcm, synth, ref_semantics):
    scene = synth.scene_for_points(1234, 200000, 8.0)
    submap = synth.sample_submap(scene, 200000, 4321)
    T0 = synth.sensor_pose(scene, 77)
    flags = 4 if ref_semantics else 0
    a = pcm.P2PlaneRegistration(0, voxel_resolution=0.5, num_neighbors=27, map_capacity=1000000, flags=flags)   # frame calls
    b = pcm.P2PlaneRegistration(0, voxel_resolution=0.5, num_neighbors=27, map_capacity=1000000, flags=flags)   # operator by operator
    a.set_input_target(submap); b.set_input_target(submap)
    off_R, off_T = [0.0, 0.0, 0.0, 1.0], [0.02, -0.01, 0.03]
    vel = T0[:3, 0] * 4.0
    poses = _poses(vel)
    for f in range(3):
        T = T0.copy(); T[:3, 3] += T[:3, 0] * 0.2 * f
        msg = _frame_msg(synth, scene, T, 30000, 900 + f)
        rot = Rotation.from_matrix(T[:3, :3]).as_quat(); pos = T[:3, 3].copy()
        end = dict(rot_xyzw=rot, pos=pos, off_R_xyzw=off_R, off_T=off_T)
        n_a = a.lio_frame_begin(msg, poses, num_scans=6, point_filter_num=2, blind=0.1, leaf_size=0.5, **end)
        # the same chain through host buffers
        flt = b.livox_filter(msg, 6, 2, 0.1)
        b.undistort(flt, 9, poses, rot, pos, off_R, off_T)          # curvature = column 9 of a PointXYZINormal record
        ds = b.voxel_downsample(flt, 0.5)
        b.set_input_source(ds)
        assert n_a == len(ds) and n_a > 1000
        assert np.array_equal(a.get_source(), np.ascontiguousarray(ds[:, :3]))
        for rematch in (True, False, False):
            ra = a.obs_model(rot, pos, off_R, off_T, False, rematch)
            rb = b.obs_model(rot, pos, off_R, off_T, False, rematch)
            assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]) and ra[2] == rb[2]
        added_a = a.lio_frame_end(rot, pos, off_R, off_T, 0.5, True)
        added_b = b.map_incremental(rot, pos, off_R, off_T, 0.5, True)
        assert added_a == added_b
        ta, tb = a.get_target(), b.get_target()
        assert ta.shape == tb.shape and np.array_equal(ta, tb)


def test_frame_from_a_device_message_and_without_imu(pcm, synth):
    import torch
    scene = synth.scene_for_points(1234, 100000, 8.0)
    submap = synth.sample_submap(scene, 100000, 4321)
    T = synth.sensor_pose(scene, 77)
    msg = _frame_msg(synth, scene, T, 20000, 5)
    a = pcm.P2PlaneRegistration(0, voxel_resolution=0.5, num_neighbors=27)
    b = pcm.P2PlaneRegistration(0, voxel_resolution=0.5, num_neighbors=27)
    a.set_input_target(submap); b.set_input_target(submap)
    d_msg = torch.from_numpy(msg.view(np.uint8).copy()).cuda()
    n = a.lio_frame_begin(d_msg, None, leaf_size=0.0, point_filter_num=1)            # device message, no compensation, no down-sampling
    flt = b.livox_filter(msg, 6, 1, 0.1)
    assert n == len(flt) and np.array_equal(a.get_source(), np.ascontiguousarray(flt[:, :3]))
    with pytest.raises(pcm.PcmError):
        a.lio_frame_begin(msg[:1], None)    # the first point of a message never passes (pointcloud_preprocess.cc:57): empty scan
