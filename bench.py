#!/usr/bin/env python3
"""bench.py -- scan registrations/sec on MI355X (BASELINE.json metric).

One *step* = one pass of the hot path over one batch: every rank aligns its
shard of independent (100k-pt Livox-shaped scan, 1M-pt submap) pairs with the
point-to-plane model -- voxel-hash 5-NN, plane fit, J^T J / J^T r reduction and
the Gauss-Newton loop to convergence -- through the C ABI (pcm_align_batch),
followed by the RCCL all-gather of the solved poses when N > 1.  Inputs are
resident in HBM and the submap voxel hashes are built before the timed region
(the reference's `100times_reuse` protocol, fast_gicp/src/align.cpp:51-104);
the cold rate (hash build included) is reported next to it.

Schedule (default `--schedule rotate --pipeline 2`): two passes are in flight per GPU, pass s on slot s % 2, each slot
with registration objects, stream and result block of its own -- the few slow-converging pairs of one pass (59 GN
iterations where the mean is 12) finish under the bulk of the next.  The timed region is EXACTLY `--steps` passes.

Prints ONE JSON line (rank 0).  Launch for N>1 with torch.distributed.run.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
SLOT_BYTES = 16                # hash slot  
POINT_BYTES = 16               # float4 query / map point
# VALU issue ceiling of the chip: 256 CUs x 4 SIMDs, one wave64 vector instruction per SIMD every 2 cycles at 2.4 GHz
# (/opt/skills/guides/MI355X_MICROARCH.md, "Wave scheduling" and the cycle-constants table; measured here 2.4-2.7 cycles:
# profiles/r02_valu_issue_rate.txt), in G wave-instructions per second
VALU_PEAK_GIPS = 1024 * 2.4 / 2.0


def source_key():
    """Hash of the kernel sources: ties a committed counter profile to the binary it was taken from."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pointcloud-slam_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_profile(name, workload_key):
    """A committed counter summary under profiles/ (rocprofv3 --pmc passes cannot run inside the timed process); returns the dict
    and whether it was taken on this workload with these kernel sources."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, False
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception:
        return None, False
    return d, bool(d.get("workload_key") == workload_key and d.get("source_key") == source_key())


def _gen_pair(args):
    pair_id, n_scan, n_map = args
    import importlib
    synth = importlib.import_module("pointcloud-slam_amd.synth")
    p = synth.make_pair(pair_id, n_scan, n_map)
    return p.scan, p.submap, p.guess, p.T_gt


def _profiler_preloaded():
    """Under rocprofv3 the preloaded tool library initialises the GPU runtime before this script starts: forking then is a fork
    of a GPU-initialised process (and the tool's signal handler fires in the pool workers at teardown)."""
    env = os.environ
    return "rocprof" in env.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCP_TOOL", "ROCPROFILER_", "ROCPROF_")) for k in env)


def generate_pairs(ids, n_scan, n_map, workers):
    """Seeded synthetic pairs, generated on host cores BEFORE anything touches the GPU (in-process, no pool, when a profiler is preloaded)."""
    jobs = [(i, n_scan, n_map) for i in ids]
    if workers <= 1 or len(jobs) == 1 or _profiler_preloaded():
        return [_gen_pair(j) for j in jobs]
    with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
        return pool.map(_gen_pair, jobs)


def cpu_baseline(pairs, cfg, budget_s, threads=None):
    """Oracle (CPU restatement of the reference path) timed on this host's cores on
    a bounded sample of the same workload.  A reported baseline, not the target."""
    from oracle import Oracle
    if threads is None:
        threads = min(len(os.sched_getaffinity(0)), 16)   # the GPU box's CPU share for one GPU
    t_all = time.perf_counter()
    n_done, t_align, iters, results = 0, 0.0, [], []
    for scan, submap, guess, _ in pairs:
        o = Oracle("P2PLANE", cfg["optimizer"], voxel_resolution=cfg["voxel_resolution"], num_neighbors=cfg["num_neighbors"],
                   max_iterations=cfg["max_iterations"], num_threads=threads)
        o.set_input_target(submap)
        o.set_input_source(scan)
        o.linearize(np.asarray(guess, np.float64))       # builds the target voxel map (excluded, like the GPU number)
        t0 = time.perf_counter()
        r = o.align(guess)
        t_align += time.perf_counter() - t0
        iters.append(r.num_linearize)
        results.append(r)
        n_done += 1
        if time.perf_counter() - t_all > budget_s:
            break
    return {"value": n_done / t_align, "unit": "registrations/s", "cores": threads, "kind": "port",
            "sample": "%d of the rank-0 pairs, oracle/ OpenMP restatement, target map prebuilt, mean %.1f linearize passes" % (n_done, float(np.mean(iters)))}, results


def full_size_parity(gpu_results, oracle_results):
    """Untimed check where the number is produced: the GPU poses of the timed workload against the oracle's on the same
    full-size pairs (the ones the cpu_baseline leg aligned).  north_star tolerance: 1e-4 m / 1e-4 rad."""
    from oracle.loader import result_T
    max_dt = max_dr = 0.0
    counts_equal = True
    for rg, ro in zip(gpu_results, oracle_results):
        D = np.linalg.inv(result_T(ro)) @ rg.T64
        max_dt = max(max_dt, float(np.linalg.norm(D[:3, 3])))
        max_dr = max(max_dr, float(np.linalg.norm(D[:3, :3] - np.eye(3))))
        counts_equal &= (rg.iterations == ro.iterations and rg.num_linearize == ro.num_linearize and rg.num_inliers == ro.num_inliers
                         and bool(rg.converged) == bool(ro.converged))
    n = min(len(gpu_results), len(oracle_results))
    return {"parity_pairs": n, "parity_max_dt_m": max_dt, "parity_max_dr_rad": max_dr, "parity_counts_equal": bool(counts_equal),
            "parity_ok": bool(n > 0 and max_dt < 1e-4 and max_dr < 1e-4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs-per-gpu", type=int, default=64, help="registrations per step and GPU (one batch = one step)")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="independent batches in flight per GPU (each on its own stream / host thread), so the few slow-converging "
                         "pairs of one batch run under the bulk of the next")
    ap.add_argument("--schedule", default="rotate", choices=["rotate", "split"],
                    help="rotate: pass s of the whole batch runs on slot s %% pipeline (every slot sees the same mix of fast and slow pairs); "
                         "split: every pass is split into fixed sub-batches, one per slot (round 1; the slot holding the slowest pair is the critical path)")
    ap.add_argument("--window", type=int, default=0, help="pairs of a sub-batch iterating at a time (0 = all): finished pairs hand their slot to queued ones")
    ap.add_argument("--flags", type=int, default=0, help="pcm_config.flags (A/B switches; never change a result)")
    ap.add_argument("--scan-points", type=int, default=100000)
    ap.add_argument("--map-points", type=int, default=1000000)
    ap.add_argument("--optimizer", default="GN", choices=["GN", "LM"])
    ap.add_argument("--max-iterations", type=int, default=64)
    ap.add_argument("--sort-source", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=40.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--gen-workers", type=int, default=0)
    ap.add_argument("--pairs-cache", default="", help="load the rank-0 pairs from an .npz written by tools/gen_cache.py (profiling runs: the program behind "
                    "rocprofv3 must not fork, and generating 64 pairs in-process takes minutes)")
    ap.add_argument("--slot-priority", type=int, default=0, help="1: the pipeline slots run on streams of descending priority (slot 0 highest), so the "
                    "tail rounds of one slot are not queued behind the bulk of another")
    ap.add_argument("--stagger", type=float, default=0.5, help="start offset between the pipeline slots, in units of one warm pass")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the measured path); gloo only to rehearse the N>1 control flow with several ranks sharing one GPU")
    ap.add_argument("--slot-streams", type=int, default=1, help="1: every pipeline slot runs on a stream of its own, created back to back; 0: on the stream of its first object")
    ap.add_argument("--collectives-at-one", type=int, default=0,
                    help="1: at N=1 still create the process group and run the per-pass all-gather (world size 1): executes the RCCL path of N>1 on one GPU")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))

    use_dist = world >= 2 or bool(args.collectives_at_one)   # N>1, or the same code path at world size 1
    if use_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
    cfg = dict(optimizer=args.optimizer, voxel_resolution=0.5, num_neighbors=27, max_iterations=args.max_iterations)

    # ---- inputs (host), before the GPU is touched so that fork() is safe -------------------------
    ncpu = min(len(os.sched_getaffinity(0)), 16 * max(1, world))
    workers = args.gen_workers or max(1, min(ncpu // max(1, min(world, 8)), 16))
    ids = [rank * args.pairs_per_gpu + i for i in range(args.pairs_per_gpu)]   # = sharding.pair_ids_for_rank (kept import-free: runs before the GPU is touched)
    t0 = time.perf_counter()
    if args.pairs_cache and world == 1:
        z = np.load(args.pairs_cache)
        if int(z["n"]) < len(ids):
            raise SystemExit("--pairs-cache holds %d pairs, %d needed" % (int(z["n"]), len(ids)))
        pairs = [(z["scan%d" % i], z["map%d" % i], z["guess%d" % i], z["gt%d" % i]) for i in ids]
    else:
        pairs = generate_pairs(ids, args.scan_points, args.map_points, workers)
    t_gen = time.perf_counter() - t0

    import torch
    import torch.distributed as dist
    import pointcloud_slam_amd as pcm
    from pointcloud_slam_amd import capi
    import ctypes

    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    saved_stdout = None
    if use_dist:
        # RCCL prints a version banner on STDOUT when the first communicator is created: stdout is kept for the one JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        if args.backend == "nccl":
            dist.init_process_group("nccl")   # backend "nccl" is RCCL on ROCm; communicators are created by the warm-up collectives below
        else:
            dist.init_process_group("gloo")

    # ---- residency: scans + submaps in HBM, voxel hashes built ------------------------------------
    import threading
    from pointcloud_slam_amd import sharding
    rotate = args.schedule == "rotate"
    n_local = len(pairs)
    S = max(1, min(args.pipeline, n_local))
    groups = sharding.split_sub_batches(n_local, S)       # split schedule: the pairs of slot j
    n_sets = S if rotate else 1                            # rotate: every slot registers the whole batch with objects of its own
    d_inputs = []
    guesses = np.stack([guess for _, _, guess, _ in pairs])
    for scan, submap, _, _ in pairs:
        d_inputs.append((torch.from_numpy(scan).to(dev), torch.from_numpy(submap).to(dev)))
    reg_sets = []
    for _ in range(n_sets):
        regs = []
        for d_scan, d_map in d_inputs:
            r = pcm.P2PlaneRegistration(dev_index, optimizer=args.optimizer, voxel_resolution=cfg["voxel_resolution"],
                                        num_neighbors=cfg["num_neighbors"], max_iterations=args.max_iterations, sort_source=args.sort_source, batch_window=args.window, flags=args.flags)
            r.set_input_target(d_map)
            r.set_input_source(d_scan)
            regs.append(r)
        reg_sets.append(regs)
    regs = reg_sets[0]
    rec = ctypes.sizeof(capi.PcmResult)
    # result blocks: split -> one block of n_local records, slot j owns its slice; rotate -> one block of n_local records per slot
    d_results = torch.zeros(n_sets * n_local * rec, dtype=torch.uint8, device=dev)
    d_gather = torch.zeros(n_sets * world * n_local * rec, dtype=torch.uint8, device=dev) if use_dist else None

    # ---- S batches in flight per GPU, each driven by its own host thread on its own stream ----
    lead = [reg_sets[j][0] if rotate else regs[groups[j][0]] for j in range(S)]    # the object whose stream / statistics a slot's batches use
    slot_streams = []
    if args.slot_priority:
        lo_p, hi_p = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
        for j in range(S):
            st_j = torch.cuda.Stream(device=dev, priority=(hi_p if j == 0 else lo_p))
            slot_streams.append(st_j)
            for r in (reg_sets[j] if rotate else [regs[i] for i in groups[j]]):
                r.set_stream(st_j.cuda_stream)
    elif S > 1 and args.slot_streams:
        # streams of the slots' own, created back to back: HIP deals streams onto a handful of hardware queues in creation order, and two
        # slots whose streams share a queue do not overlap (measured on the pclomp NDT groups: 3 200 against 2 000 registrations/s)
        for j in range(S):
            st_j = torch.cuda.Stream(device=dev)
            slot_streams.append(st_j)
            for r in (reg_sets[j] if rotate else [regs[i] for i in groups[j]]):
                r.set_stream(st_j.cuda_stream)
    # ONE communicator (the default group) and ONE thread that issues every collective, in the order (step, slot) -- the same on
    # every rank by construction.  (Round 1 gave each pipeline slot its own communicator and let the slot threads issue their
    # gathers as they finished: rank A could enqueue slot 0 then 1 while rank B enqueued 1 then 0, the documented way to hang
    # concurrent RCCL communicators.)  The slots keep overlapping their COMPUTE; a slot only waits for the gather of its previous
    # step before it overwrites that step's result block.
    comm_stream = torch.cuda.Stream(device=dev) if use_dist else None
    bar_kw = {"device_ids": [dev_index]} if (use_dist and args.backend == "nccl") else {}
    if use_dist:   # create the communicator before the timed region
        dist.all_reduce(torch.zeros(1, device=dev))
        torch.cuda.synchronize()
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)

    def sub_step(j, before_results=None):
        """One pass of the hot path on slot j: over the whole batch (rotate) or over sub-batch j (split)."""
        if rotate:
            mine, lo, hi, out_ptr = reg_sets[j], 0, n_local, d_results.data_ptr() + j * n_local * rec
        else:
            idx = groups[j]
            lo, hi = idx[0], idx[-1] + 1
            mine, out_ptr = [regs[i] for i in idx], d_results.data_ptr() + lo * rec
        # a registration starts from a NEW scan: hand every object its (HBM-resident) scan
        # again, so the on-device Morton re-ordering is inside the timed step
        for k, r in enumerate(mine):
            r.set_input_source(d_inputs[lo + k][0])
        if before_results is not None:
            t_w = time.perf_counter()
            before_results()     # the gather of this slot's previous result block has been issued and has finished
            comm_times["wait_prev_s"] += time.perf_counter() - t_w
        return pcm.align_batch(mine, guesses[lo:hi], device_out=out_ptr)

    comm_times = {"gather_s": 0.0, "wait_prev_s": 0.0, "gathers": 0}   # host time inside the collective thread / a slot waiting for it

    def gather_slot(j):
        """RCCL all-gather of slot j's solved poses over xGMI (one small collective), complete on return."""
        t_g = time.perf_counter()
        with torch.cuda.stream(comm_stream):
            if rotate:
                got = sharding.gather_records(d_results[j * n_local * rec:(j + 1) * n_local * rec], world)
                d_gather.view(S, world, n_local * rec)[j].copy_(got.view(world, n_local * rec))
            else:
                lo, hi = groups[j][0], groups[j][-1] + 1
                got = sharding.gather_records(d_results[lo * rec:hi * rec], world)
                d_gather.view(world, n_local * rec)[:, lo * rec:hi * rec].copy_(got)
        comm_stream.synchronize()
        comm_times["gather_s"] += time.perf_counter() - t_g
        comm_times["gathers"] += 1

    def run_steps(k, stagger_s=0.0):
        """k passes over the batch; returns the results of the last one (sharding.run_rotating_steps / run_pipelined_steps:
        slot threads for the compute, one thread for the collectives)."""
        def fail_fast(exc):
            # this rank cannot issue its remaining collectives: end the process, so the peers fail on a closed connection instead of
            # waiting in their all-gather for the backend's timeout
            sys.stderr.write("rank %d: %r -- aborting the job\n" % (rank, exc))
            sys.stderr.flush()
            os._exit(13)
        kw = dict(on_thread_start=lambda: torch.cuda.set_device(dev_index), on_error=fail_fast if use_dist else None)
        if rotate:
            return sharding.run_rotating_steps(k, S, sub_step, gather_slot if use_dist else None, stagger_s, **kw)
        last = sharding.run_pipelined_steps(k, S, sub_step, gather_slot if use_dist else None, stagger_s, **kw)
        return [r for grp in last for r in grp]

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(**bar_kw)
            torch.cuda.synchronize()

    # cold pass: includes voxel-hash build + scan re-ordering (lazy on first align); rotate: one pass per slot, so that every
    # slot's objects have built their maps before the timed region
    per = S if rotate else 1
    fence()
    t0 = time.perf_counter()
    res = run_steps(per)
    fence()
    t_cold = (time.perf_counter() - t0) / per

    run_steps(per)   # second registration against every target: the per-voxel candidate lists are built here (untimed, not part of t_warm)
    fence()
    t0 = time.perf_counter()
    for _ in range(max(1, args.warmup - 1)):
        run_steps(per)
    fence()
    t_warm = (time.perf_counter() - t0) / (max(1, args.warmup - 1) * per)

    for j in range(S):
        lead[j].reset_stats()
        # HIP events around every 4th residual launch, on the launch stream (bit3: sampled, the phase rotating from batch to batch:
        # bracketing EVERY launch costs 16 % of the headline -- three events on the critical path of every round)
        lead[j].set_profiling(int(os.environ.get("PCM_BENCH_LAUNCH_EVENTS", "9")))
    fence()
    t0 = time.perf_counter()
    res = run_steps(args.steps, stagger_s=args.stagger * t_warm if S > 1 else 0.0)
    fence()
    elapsed = time.perf_counter() - t0
    sts = [lead[j].stats() for j in range(S)]
    st = {k: sum(x[k] for x in sts) for k in sts[0]}
    for j in range(S):
        lead[j].set_profiling(0)

    t = torch.tensor([elapsed, t_cold], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, t_cold = float(t[0]), float(t[1])

    # single-stream pass (untimed): the dominant kernel alone on the device, HIP events around EVERY launch on the launch stream
    # -- the two-slot figure above includes the time a launch shares the CUs with the other slot's grid
    lead[0].reset_stats()
    lead[0].set_profiling(1)
    n_single = max(2, min(10, args.steps))
    if rotate:
        sharding.run_rotating_steps(n_single, 1, sub_step, None, 0.0, on_thread_start=lambda: torch.cuda.set_device(dev_index))
    else:
        sharding.run_pipelined_steps(n_single, 1, sub_step, None, 0.0, on_thread_start=lambda: torch.cuda.set_device(dev_index))
    torch.cuda.synchronize()
    s1 = lead[0].stats()
    lead[0].set_profiling(0)

    # counters pass (untimed): candidates / probes per point for the algorithmic-byte model
    for j in range(S):
        lead[j].reset_stats()
        lead[j].set_profiling(3)
    run_steps(S if rotate else 1)
    scs = [lead[j].stats() for j in range(S)]
    sc = {k: sum(x[k] for x in scs) for k in scs[0]}
    for j in range(S):
        lead[j].set_profiling(0)
    kbar = sc["candidates"] / max(1, sc["point_passes"])
    probes = sc["slots_probed"] / max(1, sc["point_passes"])

    gathered_ok = None
    if use_dist:   # untimed check of the exchange: own block round-trips, every rank's poses arrived
        allrec = d_gather.view(n_sets, world, n_local * rec)[0]
        if not torch.equal(allrec[rank], d_results[:n_local * rec]):
            raise SystemExit("rank %d: gathered block differs from the local results" % rank)
        gathered_ok = sum(int(r["status"] == 0) for r in sharding.records_to_results(allrec.reshape(-1)))

    total_regs = args.steps * n_local * world
    value = total_regs / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- roofline of the dominant kernel (k_linearize) ------------------------------------------------------------
    # The kernel is a gather + reduction served from LDS: its physical HBM traffic is a few percent of the peak, and the
    # SURVEY §8(d) byte model ("no reuse assumed": query + 27 slots + K-bar candidates per point pass) prices bytes the LDS staging
    # never moves -- round 2's fraction against it exceeded 1.  What bounds it is vector-instruction issue: the roofline is
    # wave-instructions per second against 1024 SIMDs x 2.4 GHz / 2 cycles.  Instructions per wave come from a committed
    # SQ_INSTS_VALU / SQ_WAVES pass of the same sources (profiles/pmc_valu.json, tools/r03_counters.sh); time is the step's wall time.
    workload_key = "%d_%d_%d_%s" % (args.scan_points, args.map_points, n_local, args.optimizer)
    launches = max(1, st["linearize_launches"])
    timed = max(1, st["timed_launches"])
    share = st["timed_pair_slots"] / max(1, st["launched_pair_slots"])   # the timed launches' share of the point passes: by the length of their pair lists
    pp_per_step = st["point_passes"] / args.steps                         # point passes of this rank per step
    waves_per_step = pp_per_step / 64.0
    avg_launch_ms = st["linearize_ms"] / timed                            # two slots: sampled launches, other slot's grid sharing the device
    pp_per_timed_launch = st["point_passes"] * share / timed
    single_launch_ms = s1["linearize_ms"] / max(1, s1["timed_launches"])  # the kernel alone on the device
    pp_per_single_launch = s1["point_passes"] / max(1, s1["timed_launches"])
    valu, valu_ok = load_profile("pmc_valu.json", workload_key)
    traffic_p, traffic_ok = load_profile("pmc_traffic.json", workload_key)
    valu_per_wave = float(valu["valu_per_wave"]) if valu else None
    b_pi = POINT_BYTES + cfg["num_neighbors"] * SLOT_BYTES + kbar * POINT_BYTES
    # the kernel the timed passes run (pcm_amd.h PCM_FLAG_*): candidate lists by default from a target's second registration on
    search_kernel = "k_linearize_counted" if (args.flags & 8) else ("k_linearize" if (args.flags & (64 | 32 | 2 | 1)) else "k_linearize_lists")
    roof = {"bound": "valu_issue", "unit": "G wave-instructions/s", "peak": VALU_PEAK_GIPS, "kernel": search_kernel,
            "achieved": None, "frac": None, "traffic": None}
    if valu_per_wave:
        ach = valu_per_wave * waves_per_step / (ms_per_step * 1e-3) / 1e9
        alone = valu_per_wave * (pp_per_single_launch / 64.0) / (single_launch_ms * 1e-3) / 1e9 if single_launch_ms > 0 else None
        roof.update({"achieved": ach, "frac": ach / VALU_PEAK_GIPS,
                     "valu_instructions_per_wave": valu_per_wave, "counters_match_binary": valu_ok, "counters_file": "profiles/pmc_valu.json",
                     "wave_cycles_waiting_frac": valu.get("wait_any_frac"), "wave_cycles_issue_stalled_frac": valu.get("wait_inst_any_frac"),
                     "kernel_alone_achieved": alone, "kernel_alone_frac": (alone / VALU_PEAK_GIPS) if alone else None})
    if traffic_p:
        hb = float(traffic_p["hbm_bytes_per_launch"])   # FETCH_SIZE x 2 + WRITE_SIZE of one launch of the profiled run (its launches carry the same mean pair count)
        roof.update({"traffic": hb, "traffic_match_binary": traffic_ok,
                     "hbm_GBps_physical": hb / (single_launch_ms * 1e-3) / 1e9 if single_launch_ms > 0 else None,
                     "hbm_frac_physical": hb / (single_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if single_launch_ms > 0 else None})
    roof.update({"avg_launch_ms_two_slots": avg_launch_ms, "point_passes_per_launch_two_slots": pp_per_timed_launch,
                 "avg_launch_ms_single_stream": single_launch_ms, "point_passes_per_launch_single_stream": pp_per_single_launch,
                 "launches_per_step": launches / args.steps,
                 # reconciliation: sum of the dominant kernel's launch times per step over the step's wall time; > 1 = the two slots' grids overlap
                 "sum_launch_ms_over_step_ms": (launches / args.steps) * avg_launch_ms / ms_per_step,
                 "algorithmic_bytes_per_point_pass": b_pi, "algorithmic_GBps_informational": pp_per_step * b_pi / (ms_per_step * 1e-3) / 1e9,
                 "candidates_per_point": kbar, "slots_probed_per_point": probes,
                 "tiles_on_lds_grid": sc["tiles_lds_grid"] / max(1, sc["tiles"]), "tiles_staged_through_lds": sc["tiles_lds_points"] / max(1, sc["tiles"])})

    out = None
    if rank == 0:
        iters = [r.num_linearize for r in res]
        out = {
            "metric": "scan registrations/sec (100k-pt scan vs 1M-pt submap)",
            "value": value, "unit": "registrations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 geometry / f64 accumulate", "data": "synthetic",
            "config": {"workload": "configs[1]: %d-pt Livox-shaped scan vs %d-pt submap, point-to-plane ICP (5-NN voxel hash, %s to convergence), %d independent pairs per GPU and step"
                                   " (configs[2] = the same pairs at 32 per GPU over 8 GPUs: --gpus 8 --pairs-per-gpu 32)"
                                   % (args.scan_points, args.map_points, args.optimizer, n_local),
                       "pairs_per_gpu": n_local, "batches_in_flight": S, "schedule": args.schedule, "voxel_m": cfg["voxel_resolution"], "neighbors": cfg["num_neighbors"],
                       "flags": args.flags, "search_kernel": search_kernel,
                       "target_reuse": True, "collectives_executed": bool(use_dist),
                       "gather_ms_per_pass": (1e3 * comm_times["gather_s"] / max(1, comm_times["gathers"])) if use_dist else None,
                       "slot_wait_for_gather_ms_per_pass": (1e3 * comm_times["wait_prev_s"] / max(1, comm_times["gathers"])) if use_dist else None,
                       "parallelism": "independent pairs sharded over %d GPU(s), %s all_gather of poses (one communicator, collectives issued in (step, slot) order by one thread)" % (world, "RCCL" if args.backend == "nccl" else "gloo (rehearsal, ranks share a GPU)"),
                       "mean_linearize_passes": float(np.mean(iters)), "converged": int(sum(r.converged for r in res)),
                       "cold_registrations_per_s": n_local * world / t_cold, "gen_s": t_gen, "gathered_ok": gathered_ok},
            "roofline": roof,
        }
        # the CPU leg is timed on rank 0 of the single-GPU run only (the other ranks would idle behind it)
        out["cpu_baseline"] = None
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"], oracle_results = cpu_baseline(pairs, cfg, args.cpu_seconds)
            out["config"].update(full_size_parity(res, oracle_results))   # untimed: the timed GPU poses against the oracle's, full size
            one, _ = cpu_baseline(pairs[:3], cfg, min(15.0, args.cpu_seconds), threads=1)
            out["cpu_baseline"]["one_thread"] = {"value": one["value"], "unit": one["unit"], "cores": 1, "sample": one["sample"]}
    if use_dist:
        dist.barrier(**bar_kw)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
