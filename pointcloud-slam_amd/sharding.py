"""Sharding of independent (scan, submap) pairs over the GPUs of one node and the
gather of the solved poses.

The path shards with no data-path collective: every pair is a closed problem
(SURVEY.md §8e), rank r registers its own contiguous block of pairs, and the only
exchange is ONE small all-gather of the packed ``pcm_result`` records per batch
(RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).  The record
is latency-bound (a few hundred bytes per pair), so it is issued once per batch.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import capi

RECORD_BYTES = ctypes.sizeof(capi.PcmResult)


def pair_ids_for_rank(pairs_per_rank: int, rank: int) -> list:
    """Weak scaling: every rank owns ``pairs_per_rank`` pairs; global ids are contiguous per rank."""
    return [rank * pairs_per_rank + i for i in range(pairs_per_rank)]


def run_pipelined_steps(k: int, n_slots: int, sub_step, gather=None, stagger_s: float = 0.0, on_thread_start=None, on_error=None):
    """k passes over every pipeline slot of this rank; returns the last result of every slot.

    ``sub_step(j, wait_prev)`` runs one pass of slot j on the caller's stream of that slot and returns when the slot's result
    block is complete; it must call ``wait_prev()`` (when not None) before it overwrites the block of its previous pass.
    ``gather(j)`` (None on a single rank) performs the collective for slot j's block and returns when it has completed.

    The slots are driven by one host thread each, so their compute overlaps.  Every collective is issued by ONE further thread in
    the order (step, slot) -- identical on every rank by construction, on ONE communicator: ranks may finish their slots in any
    order without the collectives ever being enqueued in different orders (the hazard of one communicator per slot thread).
    """
    import threading
    import time
    last = [None] * n_slots
    errs = []
    done = [[threading.Event() for _ in range(n_slots)] for _ in range(k)] if gather else None       # (step, slot) finished its pass
    gathered = [[threading.Event() for _ in range(n_slots)] for _ in range(k)] if gather else None   # ... and its block has been gathered

    def worker(j):
        try:
            if on_thread_start:
                on_thread_start()
            if stagger_s > 0 and j > 0:
                time.sleep(j * stagger_s)
            for step in range(k):
                wait_prev = (lambda st=step: gathered[st - 1][j].wait()) if (gather and step > 0) else None
                last[j] = sub_step(j, wait_prev)
                if gather:
                    done[step][j].set()
        except Exception as e:   # surfaced in the calling thread
            errs.append(e)
            if gather:
                for row in done:
                    row[j].set()

    def comm_worker():
        try:
            if on_thread_start:
                on_thread_start()
            for step in range(k):
                for j in range(n_slots):
                    done[step][j].wait()
                    if errs:
                        if on_error:
                            on_error(errs[0])   # see run_rotating_steps: the peers are inside this collective
                        return
                    gather(j)
                    gathered[step][j].set()
        except Exception as e:
            errs.append(e)
            if on_error:
                on_error(e)
        finally:
            for row in gathered:
                for ev in row:
                    ev.set()

    ths = [threading.Thread(target=worker, args=(j,)) for j in range(n_slots)] if (n_slots > 1 or gather) else []
    if gather:
        ths.append(threading.Thread(target=comm_worker))
    if not ths:
        worker(0)
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    if errs:
        raise errs[0]
    return last


def run_rotating_steps(k: int, n_slots: int, step_fn, gather=None, stagger_s: float = 0.0, on_thread_start=None, on_error=None):
    """k passes over the rank's WHOLE batch, pass s on pipeline slot ``s % n_slots``; returns the result of the last pass.

    ``step_fn(slot, wait_prev)`` runs one pass on the slot's own registration objects / stream and returns when the slot's result
    block is complete; it must call ``wait_prev()`` (when not None) before it overwrites the block of the slot's previous pass.
    ``gather(slot)`` (None on a single rank) performs the collective for that block and returns when it has completed.

    Unlike :func:`run_pipelined_steps` (a fixed subset of the pairs per slot) every slot sees the same mix of fast and slow
    converging pairs, so no slot is the permanent critical path; the slow tail of pass s runs under the bulk of pass s + 1.
    Collectives: ONE thread issues them in pass order -- identical on every rank by construction, on ONE communicator.

    A local failure (an exception in ``step_fn`` or ``gather``) leaves this rank's remaining collectives unissued while every other
    rank is inside its all-gather: ``on_error(exc)`` is called once, from the thread that saw it, BEFORE the error is raised in
    the caller -- a multi-rank job passes a function that ends the process (or aborts the process group), so the peers fail
    fast on a closed connection instead of hanging until the backend's timeout (round-2 advisor finding).
    """
    import threading
    import time
    n_slots = max(1, min(n_slots, k))
    results = [None] * k
    errs = []
    done = [threading.Event() for _ in range(k)] if gather else None        # pass s finished its compute
    gathered = [threading.Event() for _ in range(k)] if gather else None    # ... and its block has been gathered

    def worker(j):
        try:
            if on_thread_start:
                on_thread_start()
            if stagger_s > 0 and j > 0:
                time.sleep(j * stagger_s)
            for s in range(j, k, n_slots):
                wait_prev = (lambda prev=s - n_slots: gathered[prev].wait()) if (gather and s >= n_slots) else None
                results[s] = step_fn(j, wait_prev)
                if s >= n_slots:
                    results[s - n_slots] = None
                if gather:
                    done[s].set()
        except Exception as e:   # surfaced in the calling thread
            errs.append(e)
            if gather:
                for ev in done:
                    ev.set()

    def comm_worker():
        try:
            if on_thread_start:
                on_thread_start()
            for s in range(k):
                done[s].wait()
                if errs:
                    if on_error:
                        on_error(errs[0])
                    return
                gather(s % n_slots)
                gathered[s].set()
        except Exception as e:
            errs.append(e)
            if on_error:
                on_error(e)
        finally:
            for ev in gathered:
                ev.set()

    ths = [threading.Thread(target=worker, args=(j,)) for j in range(n_slots)] if (n_slots > 1 or gather) else []
    if gather:
        ths.append(threading.Thread(target=comm_worker))
    if not ths:
        worker(0)
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    if errs:
        raise errs[0]
    return results[k - 1]


def split_sub_batches(n_local: int, n_slots: int) -> list:
    """Contiguous sub-batches of a rank's pairs (one per pipeline slot / stream)."""
    n_slots = max(1, min(n_slots, n_local))
    bounds = [(n_local * j) // n_slots for j in range(n_slots + 1)]
    return [list(range(bounds[j], bounds[j + 1])) for j in range(n_slots)]


def gather_records(local_records, world: int, group=None):
    """All-gather the packed result records of one (sub-)batch.

    ``local_records``: uint8 tensor of n_local * RECORD_BYTES bytes (device tensor
    with the nccl/RCCL backend, CPU tensor with gloo).  Returns a (world, n_local *
    RECORD_BYTES) uint8 tensor on the same device, rank-major.
    """
    import torch
    import torch.distributed as dist
    if world == 1 and not dist.is_initialized():   # a process group of one rank still runs the collective (bench.py --collectives-at-one)
        return local_records.view(1, -1)
    if local_records.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal of the N>1 control flow on a box with fewer GPUs than ranks: gloo moves host bytes
        host = local_records.cpu()
        out = torch.empty(world * host.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(out, host, group=group)
        return out.view(world, -1).to(local_records.device)
    out = torch.empty(world * local_records.numel(), dtype=torch.uint8, device=local_records.device)
    dist.all_gather_into_tensor(out, local_records.contiguous(), group=group)
    return out.view(world, -1)


def records_to_results(raw) -> list:
    """Decode packed ``pcm_result`` bytes (tensor / bytes / ndarray) into dicts."""
    if hasattr(raw, "cpu"):
        raw = raw.cpu().numpy()
    buf = np.ascontiguousarray(np.frombuffer(bytes(raw), dtype=np.uint8))
    n = buf.size // RECORD_BYTES
    arr = (capi.PcmResult * n).from_buffer_copy(buf.tobytes())
    out = []
    for r in arr:
        out.append({"T": np.array(r.T[:], np.float32).reshape(4, 4), "T64": np.array(r.T64[:]).reshape(4, 4),
                    "iterations": r.iterations, "converged": bool(r.converged), "num_inliers": r.num_inliers,
                    "num_linearize": r.num_linearize, "status": r.status, "cost": r.cost})
    return out


def pack_results(results) -> np.ndarray:
    """Inverse of records_to_results for tests: list of dicts -> packed uint8 array."""
    arr = (capi.PcmResult * len(results))()
    for r, d in zip(arr, results):
        T64 = np.asarray(d["T64"], np.float64).reshape(16)
        for k in range(16):
            r.T64[k] = T64[k]
            r.T[k] = np.float32(T64[k])
        r.iterations = int(d.get("iterations", 0))
        r.converged = int(d.get("converged", 0))
        r.num_inliers = int(d.get("num_inliers", 0))
        r.num_linearize = int(d.get("num_linearize", 0))
        r.status = int(d.get("status", 0))
        r.cost = float(d.get("cost", 0.0))
    return np.frombuffer(bytes(arr), dtype=np.uint8).copy()
