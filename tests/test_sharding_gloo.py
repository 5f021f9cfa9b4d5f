"""CPU test of the N>1 path: world_size-2 gloo processes shard the pairs, pack their
result records and all-gather them exactly as bench.py does with RCCL (same code:
pointcloud-slam_amd/sharding.py), one process group per pipeline slot."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def _worker(rank, world, port, pairs_per_rank, slots, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pointcloud_slam_amd as pcm
    sh = pcm.sharding
    ids = sh.pair_ids_for_rank(pairs_per_rank, rank)
    groups = sh.split_sub_batches(pairs_per_rank, slots)
    pgs = [dist.new_group(list(range(world))) for _ in groups]
    gathered = np.zeros((world, pairs_per_rank * sh.RECORD_BYTES), np.uint8)
    for grp, pg in zip(groups, pgs):
        res = []
        for i in grp:
            T = np.eye(4); T[:3, 3] = [ids[i], rank, 0.5]
            res.append({"T64": T, "iterations": ids[i] % 7, "converged": 1, "num_inliers": 1000 + ids[i]})
        local = torch.from_numpy(sh.pack_results(res))
        got = sh.gather_records(local, world, group=pg).numpy()
        lo, hi = grp[0] * sh.RECORD_BYTES, (grp[-1] + 1) * sh.RECORD_BYTES
        gathered[:, lo:hi] = got
    out = [sh.records_to_results(gathered[r]) for r in range(world)]
    summary = [[(round(float(x["T64"][0, 3])), x["iterations"], x["num_inliers"], x["converged"]) for x in row] for row in out]
    q.put((rank, summary))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("slots", [1, 2])
def test_gather_of_sharded_results_gloo(slots):
    world, ppr = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ppr, slots, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [[(r * ppr + i, (r * ppr + i) % 7, 1000 + r * ppr + i, True) for i in range(ppr)] for r in range(world)]
    assert got[0] == want and got[1] == want      # every rank holds every pair's record, rank-major


def test_shard_helpers():
    sys.path.insert(0, ROOT)
    import pointcloud_slam_amd as pcm
    sh = pcm.sharding
    assert sh.pair_ids_for_rank(32, 3) == list(range(96, 128))
    assert sh.split_sub_batches(5, 2) == [[0, 1], [2, 3, 4]]
    assert sh.split_sub_batches(3, 8) == [[0], [1], [2]]
    recs = sh.pack_results([{"T64": np.eye(4), "iterations": 3, "converged": 1}])
    assert recs.size == sh.RECORD_BYTES and sh.records_to_results(recs)[0]["iterations"] == 3
