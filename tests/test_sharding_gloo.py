"""CPU tests of the N>1 path with gloo: the ranks shard the pairs, pack their result records and all-gather them with the
code bench.py runs with RCCL (pointcloud-slam_amd/sharding.py) -- including its control flow: one host thread per pipeline
slot for the compute, ONE communicator, and ONE thread that issues every collective in (step, slot) order."""
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
    gathered = np.zeros((world, pairs_per_rank * sh.RECORD_BYTES), np.uint8)
    for grp in groups:
        res = []
        for i in grp:
            T = np.eye(4); T[:3, 3] = [ids[i], rank, 0.5]
            res.append({"T64": T, "iterations": ids[i] % 7, "converged": 1, "num_inliers": 1000 + ids[i]})
        local = torch.from_numpy(sh.pack_results(res))
        got = sh.gather_records(local, world).numpy()
        lo, hi = grp[0] * sh.RECORD_BYTES, (grp[-1] + 1) * sh.RECORD_BYTES
        gathered[:, lo:hi] = got
    out = [sh.records_to_results(gathered[r]) for r in range(world)]
    summary = [[(round(float(x["T64"][0, 3])), x["iterations"], x["num_inliers"], x["converged"]) for x in row] for row in out]
    q.put((rank, summary))
    dist.barrier()
    dist.destroy_process_group()


def _pipeline_worker(rank, world, port, pairs_per_rank, slots, steps, q):
    """bench.py's N>1 control flow (sharding.run_pipelined_steps) with random per-rank, per-slot delays in place of the GPU work:
    ranks finish their slots in different orders, the collectives must still pair up and every block must arrive intact."""
    import random
    import time
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pointcloud_slam_amd as pcm
    sh = pcm.sharding
    ids = sh.pair_ids_for_rank(pairs_per_rank, rank)
    groups = sh.split_sub_batches(pairs_per_rank, slots)
    S = len(groups)
    rec = sh.RECORD_BYTES
    results = torch.zeros(pairs_per_rank * rec, dtype=torch.uint8)
    gathered = torch.zeros(world, pairs_per_rank * rec, dtype=torch.uint8)
    rng = random.Random(1000 * rank + 17)
    step_of = [0] * S
    log = []

    def sub_step(j, wait_prev):
        time.sleep(rng.uniform(0.0, 0.03) * (1 + (rank + j) % 3))     # uneven "compute": rank- and slot-dependent
        if wait_prev is not None:
            wait_prev()
        res = []
        for i in groups[j]:
            T = np.eye(4); T[:3, 3] = [ids[i], rank, step_of[j]]
            res.append({"T64": T, "iterations": step_of[j], "converged": 1, "num_inliers": 1000 + ids[i]})
        lo, hi = groups[j][0] * rec, (groups[j][-1] + 1) * rec
        results[lo:hi] = torch.from_numpy(sh.pack_results(res))
        step_of[j] += 1
        return step_of[j]

    def gather(j):
        lo, hi = groups[j][0] * rec, (groups[j][-1] + 1) * rec
        got = sh.gather_records(results[lo:hi].clone(), world)
        gathered[:, lo:hi] = got
        rows = [sh.records_to_results(got[r].numpy()) for r in range(world)]
        log.append((j, [row[0]["iterations"] for row in rows]))    # the step every rank's block of this collective belongs to

    last = sh.run_pipelined_steps(steps, S, sub_step, gather, stagger_s=0.01 * (rank % 2))
    assert last == [steps] * S
    out = [sh.records_to_results(gathered[r].numpy()) for r in range(world)]
    summary = [[(round(float(x["T64"][0, 3])), round(float(x["T64"][2, 3])), x["num_inliers"]) for x in row] for row in out]
    q.put((rank, summary, log))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,slots", [(2, 2), (3, 3), (4, 2)])
def test_pipelined_steps_with_uneven_ranks_gloo(world, slots):
    ppr, steps = 6, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, ppr, slots, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, summary, log = q.get(timeout=180)
        got[rank] = (summary, log)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [[(r * ppr + i, steps - 1, 1000 + r * ppr + i) for i in range(ppr)] for r in range(world)]   # the last step's blocks of every rank
    nslots = len([1 for _ in range(min(slots, ppr))])
    for rank in range(world):
        summary, log = got[rank]
        assert summary == want
        # every collective matched blocks of the SAME (step, slot) on all ranks, in (step, slot) order
        assert [j for j, _ in log] == [j for _ in range(steps) for j in range(nslots)]
        for n, (j, steps_seen) in enumerate(log):
            assert steps_seen == [n // nslots] * world


def _rotating_worker(rank, world, port, pairs_per_rank, slots, steps, q):
    """bench.py's default N>1 control flow (sharding.run_rotating_steps: pass s of the whole batch on slot s % slots) with random
    per-rank delays in place of the GPU work: the ranks finish their passes in different orders, a slot overwrites its result
    block only after that block's gather, and every collective must pair up blocks of the same pass."""
    import random
    import time
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pointcloud_slam_amd as pcm
    sh = pcm.sharding
    ids = sh.pair_ids_for_rank(pairs_per_rank, rank)
    S = min(slots, steps)
    rec = sh.RECORD_BYTES
    results = torch.zeros(S, pairs_per_rank * rec, dtype=torch.uint8)      # one result block per slot
    gathered = torch.zeros(S, world, pairs_per_rank * rec, dtype=torch.uint8)
    rng = random.Random(1000 * rank + 29)
    pass_of = [j - S for j in range(S)]      # the pass whose results slot j holds
    log = []

    def step_fn(j, wait_prev):
        time.sleep(rng.uniform(0.0, 0.03) * (1 + (rank + j) % 3))
        if wait_prev is not None:
            wait_prev()
        pass_of[j] += S
        res = []
        for i in range(pairs_per_rank):
            T = np.eye(4); T[:3, 3] = [ids[i], rank, pass_of[j]]
            res.append({"T64": T, "iterations": pass_of[j], "converged": 1, "num_inliers": 1000 + ids[i]})
        results[j] = torch.from_numpy(sh.pack_results(res))
        return pass_of[j]

    def gather(j):
        got = sh.gather_records(results[j].clone(), world)
        gathered[j] = got
        rows = [sh.records_to_results(got[r].numpy()) for r in range(world)]
        log.append((j, [row[0]["iterations"] for row in rows]))

    last = sh.run_rotating_steps(steps, slots, step_fn, gather, stagger_s=0.01 * (rank % 2))
    assert last == steps - 1
    jl = (steps - 1) % S
    out = [sh.records_to_results(gathered[jl, r].numpy()) for r in range(world)]
    summary = [[(round(float(x["T64"][0, 3])), round(float(x["T64"][2, 3])), x["num_inliers"]) for x in row] for row in out]
    q.put((rank, summary, log))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,slots,steps", [(2, 2, 7), (3, 3, 5), (4, 2, 6), (2, 4, 3)])
def test_rotating_steps_with_uneven_ranks_gloo(world, slots, steps):
    ppr = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rotating_worker, args=(r, world, port, ppr, slots, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, summary, log = q.get(timeout=180)
        got[rank] = (summary, log)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    S = min(slots, steps)
    want = [[(r * ppr + i, steps - 1, 1000 + r * ppr + i) for i in range(ppr)] for r in range(world)]   # the last pass, every rank's block
    for rank in range(world):
        summary, log = got[rank]
        assert summary == want
        assert [j for j, _ in log] == [s % S for s in range(steps)]            # collectives in pass order ...
        for s, (_, passes_seen) in enumerate(log):
            assert passes_seen == [s] * world                                  # ... each pairing blocks of the same pass on all ranks


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


def test_slot_schedulers_without_a_process_group():
    """run_rotating_steps / run_pipelined_steps on one rank (no collective): every pass runs once, on the slot the schedule names, the
    result of the last pass comes back, and an exception in a slot surfaces in the caller instead of hanging the other threads."""
    sys.path.insert(0, ROOT)
    import threading
    import pointcloud_slam_amd as pcm
    sh = pcm.sharding
    lock = threading.Lock()
    seen = []

    def step(slot, wait_prev):
        assert wait_prev is None                      # no gather: nothing to wait for
        with lock:
            seen.append(slot)
            return len(seen)

    last = sh.run_rotating_steps(7, 3, step)
    assert sorted(seen) == sorted([s % 3 for s in range(7)]) and last is not None
    seen.clear()
    assert sh.run_rotating_steps(2, 5, step) is not None and sorted(seen) == [0, 1]      # more slots than passes
    seen.clear()
    assert len(sh.run_pipelined_steps(4, 2, step)) == 2 and sorted(seen) == [0] * 4 + [1] * 4

    def bad(slot, wait_prev):
        if slot == 1:
            raise ValueError("boom")
        return 1

    with pytest.raises(ValueError):
        sh.run_rotating_steps(6, 2, bad)
    with pytest.raises(ValueError):
        sh.run_pipelined_steps(3, 2, bad)
    gathered = []
    with pytest.raises(ValueError):                   # with a gather thread: it must be released, not left waiting for the dead slot
        sh.run_rotating_steps(6, 2, bad, gather=lambda j: gathered.append(j))


def _failing_worker(rank, world, port, q):
    """One rank's step function raises in pass 2: with on_error ending that process the peer must come back (with an error on its
    closed connection), not hang in the all-gather."""
    import time
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    import pointcloud_slam_amd as pcm
    sh = pcm.sharding
    blocks = [torch.zeros(4 * sh.RECORD_BYTES, dtype=torch.uint8) for _ in range(2)]
    calls = {"n": 0}

    def step_fn(slot, wait_prev):
        calls["n"] += 1
        if rank == 1 and calls["n"] == 3:
            raise RuntimeError("injected failure on rank 1")
        time.sleep(0.01)
        if wait_prev:
            wait_prev()
        return [slot]

    def gather(slot):
        sh.gather_records(blocks[slot], world)

    def fail_fast(exc):
        q.put((rank, "on_error: %s" % exc))
        time.sleep(0.2)   # let the queue feeder thread flush
        os._exit(13)

    try:
        sh.run_rotating_steps(8, 2, step_fn, gather, 0.0, on_error=fail_fast)
        q.put((rank, "finished"))
    except Exception as e:   # the surviving rank: its collective fails when the peer is gone
        q.put((rank, "raised: %s" % type(e).__name__))


def test_a_failing_rank_does_not_hang_its_peers_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=90)
    alive = [p.is_alive() for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    assert not any(alive), "a rank hung after its peer failed"
    got = {}
    while not q.empty():
        r, msg = q.get()
        got[r] = msg
    assert got.get(1, "").startswith("on_error"), got
    assert procs[1].exitcode == 13
    assert 0 not in got or not got[0].startswith("finished"), got   # rank 0 cannot have completed 8 gathers without its peer
