"""A host model of the step kernel's in-launch schedule (snk_device.hpp: Sched, sched_pop, sched_push): the same
policy -- ticket FIFO filled longest-first, slices of `quantum` substeps, carry on unless a queued env-step has more
substeps left -- replayed on the substep counts a real launch had (tests/golden/bench_substeps.npy: four launches of
bench.py's workload on an MI355X, written by tools/balance_dump.py).  Pins the two claims DESIGN.md 4 makes about
it: whole env-steps cannot be levelled over 2048 waves (makespan 45 substeps for a mean load of 35), slices of a
few substeps can (within 5 % of the mean load); and the queue never strands an env-step, whatever the quantum."""
import collections
import heapq
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def run_schedule(work, quantum, waves=2048):
    """Event-driven replay.  Returns (makespan in substeps, slices per wave, substeps executed per env)."""
    n = len(work)
    rem = work.astype(int).copy()
    done = np.zeros(n, int)
    order = np.argsort(-rem, kind="stable")              # plan_sched_kernel: counting sort, longest first
    queue = collections.deque(order.tolist())
    waiting = np.zeros(64, int)
    for e in order:
        waiting[rem[e]] += 1
    events, free = [], list(range(waves))                # (time, wave, env, substeps in this slice)
    t, finished, makespan, slices = 0.0, 0, 0.0, 0
    while finished < n:
        while free and queue:                            # sched_pop: one ticket each, FIFO
            e = queue.popleft()
            waiting[rem[e]] -= 1
            w = free.pop()
            slices += 1
            run = min(quantum, rem[e])
            heapq.heappush(events, (t + run, w, e, run))
        assert events, "queue empty, nobody running, env-steps left: an env-step was stranded"
        t, w, e, run = heapq.heappop(events)
        rem[e] -= run
        done[e] += run
        if rem[e] <= 0:
            finished += 1
            makespan = max(makespan, t)
            free.append(w)
        else:
            nz = np.nonzero(waiting > 0)[0]
            if len(nz) and nz[-1] > rem[e]:              # somebody queued has more left: hand the env-step back
                waiting[rem[e]] += 1
                queue.append(e)
                free.append(w)
            else:                                        # carry on with the same env-step
                run = min(quantum, rem[e])
                heapq.heappush(events, (t + run, w, e, run))
    return makespan, slices / waves, done


@pytest.fixture(scope="module")
def launches():
    return np.load(os.path.join(HERE, "golden", "bench_substeps.npy")).astype(int)


def test_whole_env_steps_cannot_be_levelled_slices_can(launches):
    for work in launches:
        ideal = work.sum() / 2048.0
        whole, _, _ = run_schedule(work, quantum=64)
        assert whole > 1.2 * ideal                       # 45 substeps against a mean load of 35
        for q in (1, 2, 3):
            sliced, per_wave, done = run_schedule(work, quantum=q)
            assert sliced < 1.05 * ideal + q, (q, sliced, ideal)
            assert np.array_equal(done, work)            # every env-step ran exactly its substeps, once


@pytest.mark.parametrize("quantum", [1, 3, 64])
@pytest.mark.parametrize("case", ["fewer envs than waves", "all zero", "one long among short", "random"])
def test_queue_never_strands_an_env_step(quantum, case):
    rng = np.random.default_rng(1)
    work = {"fewer envs than waves": rng.integers(0, 42, 100), "all zero": np.zeros(5000, int),
            "one long among short": np.r_[41, np.ones(4999, int)], "random": rng.integers(0, 42, 5000)}[case]
    makespan, _, done = run_schedule(work, quantum, waves=256)
    assert np.array_equal(done, work)
    assert makespan >= work.max() and makespan <= work.sum() / 256.0 + work.max() + quantum
