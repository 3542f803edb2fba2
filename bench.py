#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched snake stepper on N MI355X (one process per GPU).

    python bench.py --gpus N --steps K --warmup W            (any N: for N > 1 and no WORLD_SIZE in the environment
                                                               the ranks are launched from here, as a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one batched SnakeGymEnv.step over 4096 envs per GPU (BASELINE.json configs[1] at
N=1, configs[2] at N>1): the fused env-step kernel (0..41 physics substeps per env, reward,
termination, auto-reset), for N>1 the RCCL actions scatter and the packed obs/reward/done
gather to rank 0, and on rank 0 the asynchronous D2H copy of the [obs | reward | done] block
into pinned host memory (the trainers are host-side).  Actions ("serpenoid gait") are
precomputed into ONE pinned host buffer and uploaded step by step INSIDE the timed region, on
the launch stream, as SURVEY §8(d) defines the metric ("including action upload and
obs/reward/done availability at the trainer rank"; ppo/train.py:122 pays that copy every
step).  The rate with the actions already resident in HBM is measured in a second timed
region of the same length and reported beside it (`actions_resident`).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      dominant kernel (env_step_sched_kernel) against the HBM roofline, algorithmic
                bytes / HIP-event launch time
  cpu_baseline  the CPU oracle (oracle/, kind "port") timed on this box's host cores on a
                bounded sample of the same workload (N=1 only)
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
N_LINKS = 16
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
# algorithmic bytes, N=16.  One physics substep of one env: read state
# (45 f32) + motor targets (16 f32), write state = 424 B; per env-step add action 32 B,
# obs 224 B, reward 4 B, done 1 B, substep count 4 B.
BYTES_PER_SUBSTEP = 424
BYTES_PER_ENVSTEP = 32 + 224 + 4 + 1 + 4


# The synthetic inputs of SURVEY 8(d) -- per-env gait phases, the serpenoid action stream, configs[4]'s per-env plane
# friction -- live in bullet-envs_amd/synthetic.py (pure numpy), loaded by path so that nothing of the package is imported
# before the CPU baseline runs; tests take them from the same file.
def _load_synthetic():
    import importlib.util
    spec = importlib.util.spec_from_file_location("snk_synthetic", os.path.join(ROOT, "bullet-envs_amd", "synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_syn = _load_synthetic()
splitmix64, env_phases, gait_actions, env_friction = _syn.splitmix64, _syn.env_phases, _syn.gait_actions, _syn.env_friction


# --------------------------------------------------------------------------------------
# CPU baseline: the oracle (restatement of the reference's CPU path) on the host cores
# --------------------------------------------------------------------------------------
def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(steps=2000, warmup=20, n_links=16, friction_seed=None, hull_sides=32, contact_model=1, warm_start=0,
                 self_collision=1, one_thread_only=False, contact_order=0):
    """BASELINE.md row B3 / SURVEY 8(d): the float64 C++ oracle on the configs[0] action stream, timed from C++
    (oracle/snake_oracle.cpp: orc_bench_gait -- no Python in the timed loop), 1 thread and all cores, `steps`
    env-steps after `warmup` warm-up steps each.  PyBullet itself is probed at run time and reported, never
    substituted silently.  one_thread_only: the short form the variants use (no all-cores leg)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    import pybullet_live
    orc.build()
    have_pb = pybullet_live.available()
    nproc = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # "all cores" = the cores this process may actually use: the affinity mask, cut down to the cgroup's CPU quota
    # where there is one (a 1-GPU box of the pool exposes all 256 hardware threads of the host but grants a share)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        pass
    cores = max(1, min(nproc, quota) if quota else nproc)
    if quota is None and nproc > 64:
        cores = 64          # no quota visible: one thread per physical core of a 64-core socket at most
    # the same model switches as the GPU run (self_collision: inert for 16 links under the gait; the device evaluates
    # the flag for both chains)
    kw = dict(n_modules=n_links, hull_sides=hull_sides, contact_model=contact_model, self_collision=self_collision,
              warm_start=warm_start, contact_order=contact_order)

    def mu_of(ids):
        return None if friction_seed is None else env_friction(ids, friction_seed)
    # one thread: env 0 (the canonical gait, phi_0 = 0) for `steps` env-steps
    sec1, sub1 = orc.bench_gait(1, env_phases([0]), warmup, steps, 1, mu_plane=mu_of([0]), **kw)
    one = {"value": steps / sec1, "unit": "env-steps/s", "cores": 1, "substeps_per_s": sub1 / sec1,
           "mean_substeps": sub1 / float(steps), "env_steps": steps, "seconds": sec1}
    pb_note = ("a PyBullet IS importable on this box: run oracle/pybullet_live.py to pin the oracle against it"
               if have_pb else "`import pybullet` fails on this box, probed at run time")
    fr_note = "" if friction_seed is None else ", plane friction U[0.5,1.5) seed %d" % friction_seed
    if one_thread_only:
        one.update({"kind": "port", "cpu_model": _cpu_model(), "n_links": n_links,
                    "sample": "float64 C++ oracle (oracle/; %s), gait stream of BASELINE configs[0]%s, one thread: env 0 x %d "
                              "env-steps after %d warm-up steps (%.1f s)" % (pb_note, fr_note, steps, warmup, sec1)})
        return one
    # all cores: one env per core, the same number of env-steps in total (at least 20 batched steps)
    per = max(20, -(-steps // cores))
    ids = np.arange(cores)
    secN, subN = orc.bench_gait(cores, env_phases(ids), warmup, per, cores, mu_plane=mu_of(ids), **kw)
    return {
        "value": cores * per / secN, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": "float64 C++ oracle (oracle/, restates PyBullet's pipeline; %s), gait stream of BASELINE configs[0]%s, "
                  "timed inside C++ after %d warm-up steps: all cores = %d envs x %d env-steps on %d threads (%.1f s); "
                  "one thread = env 0 x %d env-steps (%.1f s)" % (pb_note, fr_note, warmup, cores, per, cores, secN, steps, sec1),
        "substeps_per_s": subN / secN, "mean_substeps": subN / float(cores * per),
        "one_thread": one,
        "cpu_model": _cpu_model(), "nproc": nproc, "cgroup_cpu_quota": quota, "n_links": n_links,
    }


# --------------------------------------------------------------------------------------
# roofline block: algorithmic bytes over the HIP-event launch time, + what the committed rocprofv3 summaries of this
# configuration say (PMC counters need the profiler around the process: replayed, and labelled so)
# --------------------------------------------------------------------------------------
# tools/ubench_solve.hip (round 4, profiles/r04_ubench_solve.txt): the row steps of the register-resident solve -- the
# instruction mix that is 82 % of this kernel's VALU instructions -- sustain this rate at the kernel's own two waves per
# SIMD (0.20 for one wave alone, 0.253 at three, 0.255 at four), although a plain v_fma stream issues every 2 clocks
SOLVE_MIX_VALU_PER_CLK_PER_SIMD = 0.248
N_SIMDS = 1024


def profile_replay(cfg_key):
    import glob
    traffic = traffic_src = valu = None
    try:
        # (round 3 changed the default contact model: summaries of earlier rounds describe other kernels)
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_%s_pmc_summary.json" % cfg_key)))
        if cands:
            with open(cands[-1]) as f:
                pm = json.load(f)
            traffic = pm.get("hbm_bytes_per_launch")
            traffic_src = os.path.relpath(cands[-1], ROOT)
            sq = pm.get("sq_per_launch", {})
            if sq.get("SQ_INSTS_VALU") and sq.get("GRBM_GUI_ACTIVE"):
                # the path is VALU-bound.  GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = the launch's
                # clocks; a SIMD-32 takes a wave64 VALU instruction every 2 clocks at best (MI355X_MICROARCH.md).
                cycles = sq["GRBM_GUI_ACTIVE"] / 8.0
                per_clk = sq["SQ_INSTS_VALU"] / (cycles * N_SIMDS)
                valu = {"wave_insts_per_launch": sq["SQ_INSTS_VALU"], "launch_clocks": cycles,
                        "valu_per_clk_per_simd": per_clk,
                        "valu_pipe_frac": 2.0 * per_clk,
                        "valu_pipe_frac_note": "SQ_INSTS_VALU x 2 clocks / (launch clocks x 1024 SIMDs): the fraction of the "
                                               "2-clock issue slots a plain v_fma stream could fill",
                        "solve_mix_ceiling_valu_per_clk_per_simd": SOLVE_MIX_VALU_PER_CLK_PER_SIMD,
                        "frac_of_solve_mix_ceiling": per_clk / SOLVE_MIX_VALU_PER_CLK_PER_SIMD,
                        "valu_active_per_wave": pm.get("derived", {}).get("valu_active_fraction_of_wave_cycles"),
                        "profiled_kernel_ms": pm.get("kernel_trace_average_ms")}
    except Exception:  # noqa: BLE001
        pass
    return traffic, traffic_src, valu


def roofline_block(NL, E, K, local_sub, kernel_ms, kcount, cfg_key):
    per_sub = BYTES_PER_SUBSTEP if NL == 16 else 744
    per_env = BYTES_PER_ENVSTEP if NL == 16 else 64 + 416 + 4 + 1 + 4
    alg_bytes_launch = (local_sub * per_sub + E * K * per_env) / K
    achieved = alg_bytes_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    traffic, traffic_src, valu = profile_replay(cfg_key)
    return {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
        "traffic_note": "HBM bytes per launch from separate rocprofv3 --pmc passes of this configuration, "
                        "replayed from the committed summary (not collected in this run)",
        "kernel": "env_step_sched_kernel<%d> (+ its plan kernel)" % NL, "kernel_ms": kernel_ms, "launches": kcount,
        "algorithmic_bytes_per_launch": alg_bytes_launch,
        "valu_replayed_from_profile": valu,
        # measured bytes over the profiled launch's duration: what the memory system actually carried
        "traffic_rate_GBps_profiled": (traffic / (valu["profiled_kernel_ms"] * 1e-3) / 1e9
                                       if (traffic and valu and valu.get("profiled_kernel_ms")) else None),
        "note": ("recurrence-bound path: ~1e3 flop per algorithmic byte; the HBM fraction is reported as the contract asks, "
                 "it is not the limiter: the VALU issue rate of the solve's instruction mix is "
                 "(valu_replayed_from_profile, DESIGN.md 5)") if NL == 16 else
                ("38 velocity components do not fit the register-resident solve: the constraint rows stream from memory "
                 "50 times per substep, and THAT stream bounds this kernel (traffic_rate_GBps_profiled against the "
                 "8 TB/s peak); the algorithmic-byte fraction above is reported as the contract asks (DESIGN.md 8)"),
    }


HIST_STEPS = 10      # env-steps of the (untimed) pass that counts contact points per substep


def histogram_pass(env, torch, acts):
    """Contact points per physics substep (snk_contact_histogram) over len(acts) more env-steps, AFTER and outside the
    timed region: the counting costs one atomic per substep (0.7 % of the rate), so it is off while anything is timed."""
    st = env.stepper
    st.contact_histogram_enable(True)
    st.contact_histogram(reset=True)
    for a in acts:
        env.step(a)
    torch.cuda.synchronize()
    out = histogram_summary(st)
    st.contact_histogram_enable(False)
    return out


def histogram_summary(stepper):
    h = stepper.contact_histogram(reset=True).astype(np.int64)
    n = int(h.sum())
    if n == 0:
        return None
    c = np.cumsum(h)
    out = {"substeps": n, "mean": float((h * np.arange(len(h))).sum() / n)}
    for p in (50, 90, 99):
        out["p%d" % p] = int(np.searchsorted(c, p / 100.0 * n))
    out["max"] = int(np.nonzero(h)[0].max())
    return out


def measure_variant(pkg, torch, dev, device_index, E, NL, K, W, friction_seed=None, cpu_steps=0, cfg_key=None, **params):
    """One more single-GPU configuration, after and outside the headline's timed region: K batched env-steps after W
    warm-up steps on its own handle, HIP-event kernel time, roofline block, and (cpu_steps > 0) the one-thread leg of
    the CPU baseline on the same configuration."""
    A = NL // 2
    env = pkg.DeviceVecEnv(E, device_index=device_index, n_modules=NL, **params)
    if friction_seed is not None:
        env.set_ground_friction(env_friction(np.arange(E), friction_seed).astype(np.float32))
    gids = np.arange(E)
    # as in the headline: the action blocks wait in ONE pinned host buffer and are uploaded step by step inside the
    # timed region, results go to pinned host memory
    acts = torch.empty((W + K + HIST_STEPS, E, A), dtype=torch.float32).pin_memory()
    for j in range(W + K + HIST_STEPS):
        acts[j] = torch.from_numpy(gait_actions(gids, j, A).astype(np.float32))
    a_dev = torch.empty((E, A), dtype=torch.float32, device=dev)
    O = env.obs_dim
    h_all = torch.empty((E, O + 2), dtype=torch.float32).pin_memory()        # rows [obs | reward | done], one D2H copy per step
    pack_dev = torch.zeros((E, O + 2), dtype=torch.float32, device=dev)
    sub_rows = torch.zeros((K, E), dtype=torch.int32, device=dev)            # substep counts, reduced after the timed region
    env.reset()

    def step(j, row=None):
        a_dev.copy_(acts[j], non_blocking=True)
        env.step_packed(a_dev, pack_dev, substeps=None if row is None else sub_rows[row])
        h_all.copy_(pack_dev, non_blocking=True)
    for j in range(W):
        step(j)
    torch.cuda.synchronize()
    env.stepper.timing_enable(K)
    t0 = time.perf_counter()
    for j in range(W, W + K):
        step(j, j - W)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    kernel_ms, kcount = env.stepper.timing_read()
    nsub = float(sub_rows.sum().item())
    ov = env.stepper.contact_overflow()
    hist = histogram_pass(env, torch, [acts[j].to(dev) for j in range(W + K, W + K + HIST_STEPS)])
    out = {"value": E * K / el, "unit": "env-steps/s", "steps": K, "warmup": W, "ms_per_step": 1e3 * el / K,
           "n_links": NL, "friction_seed": friction_seed, "substeps_per_s": nsub / el,
           "mean_substeps_per_env_step": nsub / (E * K),
           "contact_overflow": {"substeps_on_streamed_rows": ov[0], "points_without_rows": ov[1],
                                "link_link_or_obstacle_without_rows": ov[2]},
           "contacts_per_substep": hist,
           "roofline": roofline_block(NL, E, K, nsub, kernel_ms, kcount, cfg_key) if cfg_key else None}
    out.update({k: v for k, v in params.items()})
    env.close()
    del acts
    if cpu_steps > 0:
        out["cpu_baseline"] = cpu_baseline(steps=cpu_steps, warmup=5, n_links=NL, friction_seed=friction_seed,
                                           hull_sides=params.get("hull_sides", 32), contact_model=params.get("contact_model", 1),
                                           self_collision=params.get("self_collision", 1), one_thread_only=True)
    return out


def self_launch(n, argv):
    """`python bench.py --gpus N` from a plain shell, N > 1: the reference's SubprocVecEnv fans out by itself when it is
    constructed (ppo/multiprocessing_env.py:97-117); so does this.  One rank per GPU is started through
    torch.distributed.run as a CHILD process (never an exec: this parent has not touched the GPU and never will), on a
    port taken from a bound socket (repeated on another port if exactly that port was taken in between); rank 0's JSON
    line -- the last one, should there be several -- is relayed as this process's only stdout line, everything else the
    ranks print (stdout and stderr) goes to stderr, and the exit code is the child's."""
    import socket
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SNK_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this pool's hosts
    env.setdefault("OMP_NUM_THREADS", "1")
    rc, lines = 1, []
    for attempt in range(3):
        # A port from a bound socket is free when it is taken and may be gone when torch.distributed.run binds it (another
        # process can grab it in between: ADVICE r5).  The window cannot be closed from here, so a launch that dies of
        # exactly that is repeated on another port -- and of nothing else.
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, text=True, cwd=ROOT)
        lines, in_use = [], False
        for ln in proc.stdout:
            if ln.startswith("{") and '"metric"' in ln:
                lines.append(ln.rstrip("\n"))
            else:
                in_use = in_use or "EADDRINUSE" in ln or "ddress already in use" in ln
                sys.stderr.write(ln)
        rc = proc.wait()
        if rc == 0 or lines or not in_use:
            break
        sys.stderr.write("bench.py: port %d was taken before the launcher could bind it; once more on another port\n" % port)
    if len(lines) > 1:
        sys.stderr.write("bench.py: the ranks printed %d JSON lines, expected one (rank 0's); relaying the last\n" % len(lines))
    if lines:
        print(lines[-1], flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited 0 but rank 0 printed no JSON line\n")
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--links", type=int, default=N_LINKS, choices=(16, 32),
                    help="32 = BASELINE configs[3] (a parity-test case, not the headline metric)")
    ap.add_argument("--friction-seed", type=int, default=None,
                    help="BASELINE configs[4]: per-env ground friction mu_e ~ U[0.5, 1.5), counter-based generator with "
                         "this seed (the config says seed 1), keyed by the global env index")
    ap.add_argument("--hull-sides", type=int, default=32,
                    help="32 (default) = cylinders as the 32-gon hulls PyBullet imports; 0 = implicit cylinders, the "
                         "round-1 model (DESIGN.md 3)")
    ap.add_argument("--contact-model", type=int, default=1, choices=(0, 1),
                    help="1 (default) = Bullet's persistent <= 4-point contact manifold; 0 = the stateless two-point "
                         "manifold of round 1 (DESIGN.md 3)")
    ap.add_argument("--warm-start", type=int, default=0, choices=(0, 1),
                    help="1 = warm-started normal rows (off in Bullet's multibody solver, hence off by default)")
    ap.add_argument("--self-collision", type=int, default=1, choices=(0, 1),
                    help="link-link contacts (the reference's URDF_USE_SELF_COLLISION load flag, snake.py:93): evaluated by "
                         "the 32-link kernels; 0 switches them off (round-1 state of configs[3])")
    ap.add_argument("--contact-order", type=int, default=0,
                    help="snk_params::contact_order (DESIGN.md 3): the order in which the solver sweeps the ground manifolds; "
                         "0 (default) link order, 1 reversed, 2 link order after Bullet's quickSort on equal island ids, "
                         "k >= 3 fixed permutations -- an error-bar switch, not a BASELINE config")
    ap.add_argument("--cpu-steps", type=int, default=2000,
                    help="env-steps of the CPU baseline (BASELINE.md B3: 2000 after 20 warm-up steps)")
    ap.add_argument("--streamed-rows", action="store_true",
                    help="diagnostic: 16 links on the streamed-row solve (SNK_FORCE_STREAMED=1)")
    ap.add_argument("--obstacle", type=float, default=None, metavar="X",
                    help="not a BASELINE config: the block of snake_gait_test.py:51 (static) with its centre at x = X")
    ap.add_argument("--obstacle-free", action="store_true",
                    help="with --obstacle: the block as the reference loads it, a free 200-kg body (obstacle = 2)")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the extra measurement of the round-1 contact model (1 GPU, default configuration only)")
    ap.add_argument("--profile", action="store_true",
                    help="the form rocprofv3 is put around (tools/prof_passes.sh): W + K launches of the step kernel and "
                         "nothing else -- no CPU baseline, no variants, no second (actions-resident) timed region, no "
                         "histogram pass -- so that a trace's launches W .. W + K - 1 ARE the timed region")
    ap.add_argument("--policy", action="store_true",
                    help="not the BASELINE metric: actions sampled from an on-device 2x256 actor-critic "
                         "(bullet-envs_amd/rollout.py, SURVEY 8(f)-1) instead of the precomputed gait; 1 GPU")
    args = ap.parse_args()
    if args.profile:
        args.no_cpu_baseline = args.no_variants = True

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain shell: launch the ranks ourselves, BEFORE torch or any HIP call in this process
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # C++ threads inside this process (no fork): safe before or after GPU initialisation, profiler or not
        if args.obstacle is None and not args.policy:
            cpu = cpu_baseline(steps=args.cpu_steps, n_links=args.links, friction_seed=args.friction_seed,
                               hull_sides=args.hull_sides, contact_model=args.contact_model, warm_start=args.warm_start,
                               self_collision=args.self_collision, contact_order=args.contact_order)
        # (with --obstacle / --policy the C++ gait driver would time another workload than the GPU: no baseline then)

    import importlib
    import torch
    pkg = importlib.import_module("bullet-envs_amd")

    E, K, W = args.envs_per_gpu, args.steps, args.warmup
    NL = args.links
    A = NL // 2
    n_dev = torch.cuda.device_count()                 # (counting devices does not initialise the GPU)
    if os.environ.get("SNK_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % max(1, n_dev)       # rehearsal: the ranks share the GPUs there are
    elif world > n_dev:
        # one process per GPU: refuse rather than put two RCCL ranks on one device
        raise SystemExit("bench.py: --gpus %d but this node shows %d GPU(s) (torch.cuda.device_count()); "
                         "SNK_BENCH_BACKEND=gloo rehearses N ranks on fewer GPUs" % (world, n_dev))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SNK_BENCH_BACKEND", "nccl")     # "gloo": functional rehearsal of N ranks on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        # the record must prove what ran: RCCL (or the rehearsal backend) saw exactly --gpus ranks, one per device
        assert dist.get_world_size() == args.gpus == world, (dist.get_world_size(), args.gpus, world)
        assert dist.get_rank() == rank, (dist.get_rank(), rank)

    rank_info = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(),
                 "name": torch.cuda.get_device_name(dev), "pid": os.getpid()}
    try:
        rank_info["pci_bus_id"] = torch.cuda.get_device_properties(dev).pci_bus_id
    except Exception:  # noqa: BLE001  (older torch: no such field)
        pass
    ranks = [rank_info]
    if dist is not None:
        ranks = [None] * world
        dist.all_gather_object(ranks, rank_info)
        if backend == "nccl":
            # one GPU per rank: no two ranks on the same device of this node
            assert len({(r["device"], r.get("pci_bus_id")) for r in ranks}) == world, ranks
    extra = {}
    if args.streamed_rows:
        os.environ["SNK_FORCE_STREAMED"] = "1"           # read by snk_create
    if args.warm_start:
        extra["warm_start"] = 1
    if args.contact_order:
        extra["contact_order"] = args.contact_order
    if args.obstacle is not None:
        extra.update(obstacle=2 if args.obstacle_free else 1, obstacle_pos=[args.obstacle, 0.0, 0.1])
    local = pkg.DeviceVecEnv(E, device_index=local_rank, n_modules=NL, hull_sides=args.hull_sides,
                             contact_model=args.contact_model, self_collision=args.self_collision, **extra)
    if args.friction_seed is not None:      # configs[4]: this rank's shard of the per-env plane friction
        local.set_ground_friction(env_friction(np.arange(rank * E, (rank + 1) * E), args.friction_seed).astype(np.float32))
    env = pkg.ShardedVecEnv(local, root=0, device=dev) if world > 1 else None

    # actions for every step of both timed regions: ONE pinned host buffer on the trainer rank; step j's block is
    # uploaded inside the timed region, on the launch stream, in front of the step
    second = 0 if (args.profile or args.policy) else K        # the actions-resident region (same length, gait carried on)
    total = W + K + second
    if rank == 0:
        gids = np.arange(world * E)
        h_acts = torch.empty((total, world * E, A), dtype=torch.float32).pin_memory()
        for j in range(total):
            h_acts[j] = torch.from_numpy(gait_actions(gids, j, A).astype(np.float32))
        acts_dev = torch.empty((world * E, A), dtype=torch.float32, device=dev)      # N = 1: the step's action block
        acts_res = h_acts[W + K:].to(dev) if second else None                       # second region: resident in HBM
        O = local.obs_dim
        if world > 1:
            # the gathered [world * E, O + 2] block comes to the host as it is, in ONE contiguous copy; obs / reward /
            # done are host-side views of the pinned buffer (no strided-copy kernel on the root)
            h_all = torch.empty((world * E, O + 2), dtype=torch.float32).pin_memory()
            h_obs, h_rew, h_done = h_all[:, :O], h_all[:, O], h_all.view(torch.int32)[:, O + 1]
        else:
            # N = 1: the step kernel writes rows [obs | reward | done] of one block (snk_step_packed, what the sharded path
            # gathers) and the block goes to the host in ONE copy (rounds 1-5: three tensors, three copies)
            h_all = torch.empty((E, O + 2), dtype=torch.float32).pin_memory()
            h_obs, h_rew, h_done = h_all[:, :O], h_all[:, O], h_all.view(torch.int32)[:, O + 1]
            pack_dev = torch.zeros((E, O + 2), dtype=torch.float32, device=dev)
    # Snake.counter of every env, one row per step of a timed region: the mean substep count is a statistic of the run,
    # reduced AFTER the region (rounds 1-5 reduced it after every step: two small kernels per step on the launch stream)
    sub_rows = torch.zeros((max(K, 1), E), dtype=torch.int32, device=dev)
    sub_total = torch.zeros((), dtype=torch.int64, device=dev)

    if world > 1:
        env.reset()
    else:
        local.reset()

    net = None
    if args.policy:
        assert world == 1, "--policy measures one GPU"
        torch.manual_seed(0)
        net = pkg.rollout.ActorCritic(local.obs_dim, A, [256, 256]).to(dev)
        pol_state = {"obs": local.obs, "h_obs": torch.empty((E, local.obs_dim), dtype=torch.float32).pin_memory(),
                     "h_rew": torch.empty((E,), dtype=torch.float32).pin_memory(),
                     "h_done": torch.empty((E,), dtype=torch.uint8).pin_memory()}

    def one_step(j, resident=False, row=None):
        # row: the row of sub_rows this step's substep counts go to (None: warm-up, into the env's own buffer)
        if net is not None:
            with torch.no_grad():
                mu, sigma, _value = net.heads(pol_state["obs"])
                act = torch.normal(mu, sigma)
            obs, rew, done = local.step(act)
            sub_total.add_(local.substeps.sum())
            pol_state["h_obs"].copy_(obs, non_blocking=True)
            pol_state["h_rew"].copy_(rew, non_blocking=True)
            pol_state["h_done"].copy_(done, non_blocking=True)
            return
        # resident: the second timed region -- the action block is in HBM already (rounds 1-5's headline)
        if world > 1:
            # the root hands step_block() its pinned host block (H2D inside, non-blocking, then the scatter) or, in the
            # second region, the block resident on its device
            blk = env.step_block((acts_res[j - W - K] if resident else h_acts[j]) if rank == 0 else None,
                                 substeps=None if row is None else sub_rows[row])
            if rank == 0:     # trainer side: the whole block to pinned host memory, one contiguous D2H copy
                h_all.copy_(blk, non_blocking=True)
            return
        if resident:
            a_j = acts_res[j - W - K]
        else:
            acts_dev.copy_(h_acts[j], non_blocking=True)      # 128 KiB H2D on the launch stream, in front of the step
            a_j = acts_dev
        local.step_packed(a_j, pack_dev, substeps=None if row is None else sub_rows[row])
        # trainer side: obs / reward / done to pinned host memory, one contiguous D2H copy of the block
        h_all.copy_(pack_dev, non_blocking=True)

    def timed_region(j0, resident):
        """K steps j0 .. j0 + K - 1 bracketed by barrier + synchronize on both sides; returns (this rank's seconds,
        the maximum over ranks, substeps of all ranks, this rank's substeps, HIP-event kernel ms, launches)."""
        sub_total.zero_()
        sub_rows.zero_()
        local.stepper.timing_enable(K)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(j0, j0 + K):
            one_step(j, resident, j - j0)
        torch.cuda.synchronize()
        t_own = time.perf_counter() - t0            # this rank's own K steps (a straggler shows in config.ranks)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        k_ms, k_n = local.stepper.timing_read()
        if net is None:
            sub_total.add_(sub_rows.sum())           # (the policy path keeps its per-step reduction)
        t_el = torch.tensor([el], dtype=torch.float64, device=dev)
        subs = sub_total.to(torch.float64).reshape(1)
        loc = float(subs.item())
        if dist is not None:
            dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
            dist.all_reduce(subs, op=dist.ReduceOp.SUM)
        return t_own, float(t_el.item()), float(subs.item()), loc, k_ms, k_n

    for j in range(W):
        one_step(j)
    torch.cuda.synchronize()
    own_s, elapsed, substeps, local_sub, kernel_ms, kcount = timed_region(W, False)
    resident = None
    if second:
        r_own, r_el, r_sub, _r_loc, r_kms, _ = timed_region(W + K, True)
        resident = {"value": world * E * K / r_el, "unit": "env-steps/s", "ms_per_step": 1e3 * r_el / K, "kernel_ms": r_kms,
                    "mean_substeps_per_env_step": r_sub / (world * E * K),
                    "note": "a second timed region of the same K steps (the gait carried on), the action blocks resident in "
                            "HBM before it starts: what rounds 1-5 reported as `value`"}
    # every rank's own time over the headline's K steps reaches the record
    rank_info["ms_per_step"] = 1e3 * own_s / K
    rank_info["kernel_ms"] = kernel_ms
    if dist is not None:
        ranks = [None] * world
        dist.all_gather_object(ranks, rank_info)
    else:
        ranks = [rank_info]
    # the record must show every rank's own time, not only the maximum (a straggler is then visible)
    assert len(ranks) == world and all(r is not None and "ms_per_step" in r and "kernel_ms" in r for r in ranks), ranks

    # [0] substeps that went through the streamed-row solve because their contacts outgrew the register-resident one's
    # slots (16 links), [1] [2] contacts left without rows: must be zeros
    overflow = local.stepper.contact_overflow()
    # Not the headline: further single-GPU measurements, after and outside its timed region (default run only).
    #   configs3_c32 / configs4_c16_fric: BASELINE configs[3] and [4], so that the driver's record carries a timed line
    #     for every single-GPU configuration (own K, ms_per_step, mean substeps, roofline block, one-thread CPU leg);
    #   round1_contact_model: the stateless two-point manifold on implicit cylinders that rounds 1 and 2 measured
    #     (self_collision 0: those rounds built no link-link rows for 16 links), keeps the record comparable.
    variants = None
    headline_hist = None
    if rank == 0 and world == 1 and not args.policy and not args.profile:
        # (an untimed pass of its own, the gait stream carried on)
        more = [torch.from_numpy(gait_actions(np.arange(E), total + j, A).astype(np.float32)).to(dev) for j in range(HIST_STEPS)]
        headline_hist = histogram_pass(local, torch, more)
    if (world == 1 and not args.no_variants and not args.policy and NL == 16 and args.hull_sides == 32
            and args.contact_model == 1 and not args.warm_start and args.friction_seed is None
            and not args.streamed_rows and args.obstacle is None and args.self_collision == 1 and not args.contact_order):
        local.close()
        variants = {}
        Kv = min(K, 50)
        variants["configs4_c16_fric"] = measure_variant(pkg, torch, dev, local_rank, E, 16, Kv, min(W, 10), friction_seed=1,
                                                        cpu_steps=min(args.cpu_steps, 400), cfg_key="c16_fric")
        variants["configs4_c16_fric"]["workload"] = "BASELINE configs[4]: 4096 envs x 16-link snake, ground friction mu_e ~ U[0.5,1.5) seed 1"
        variants["configs3_c32"] = measure_variant(pkg, torch, dev, local_rank, E, 32, min(K, 20), min(W, 6),
                                                   cpu_steps=min(args.cpu_steps, 120), cfg_key="c32")
        variants["configs3_c32"]["workload"] = "BASELINE configs[3]: 4096 envs x 32-link snake (obs 104, action 16), flat ground"
        v = measure_variant(pkg, torch, dev, local_rank, E, 16, Kv, min(W, 10), hull_sides=0, contact_model=0,
                            relative_breaking_threshold=0, self_collision=0)
        v["note"] = ("same action stream; the stateless two-point manifold on implicit cylinders that rounds 1 and 2 "
                     "measured, instead of PyBullet's 32-gon hull import + Bullet's persistent <= 4-point manifold at "
                     "the dispatcher's relative breaking threshold (the default since round 3, DESIGN.md 3)")
        variants["round1_contact_model"] = v

    if rank == 0:
        n_env_steps = world * E * K
        # rocprofv3 summaries of THIS configuration (profiles/README.md) are replayed into the roofline block: they are
        # measured in separate --pmc runs of the same command, not in this run; the key names say so.
        cfg_key = ("c%d" % NL) + ("_fric" if args.friction_seed is not None else "") + ("_policy" if args.policy else "") + (
            "_hull%d_cm%d" % (args.hull_sides, args.contact_model) if (args.hull_sides != 32 or args.contact_model != 1) else "") + (
            "_warm" if args.warm_start else "") + ("_order%d" % args.contact_order if args.contact_order else "") + (
            "_nosc" if (NL == 32 and not args.self_collision) else "") + ("_streamed" if args.streamed_rows else "") + (
            ("_obstacle2" if args.obstacle_free else "_obstacle") if args.obstacle is not None else "")
        cfg_index = 3 if NL == 32 else (4 if args.friction_seed is not None else (1 if world == 1 else 2))
        out = {
            "metric": "env-steps/sec (whole node), %d-link snake, 4096 envs/GPU" % NL + (
                " -- with on-device policy inference (not the BASELINE metric)" if args.policy else ""),
            "value": n_env_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": ("%d envs/GPU x %d-link snake, %s, %s "
                             "(BASELINE configs[%d])" % (E, NL,
                                                         "flat ground plane" if args.friction_seed is None else
                                                         "ground friction mu_e ~ U[0.5,1.5) seed %d" % args.friction_seed,
                                                         "actions sampled from a random-init 2x256 actor-critic on the GPU"
                                                         if args.policy else "serpenoid gait actions", cfg_index)),
                "envs_per_gpu": E, "n_links": NL, "friction_seed": args.friction_seed,
                "hull_sides": args.hull_sides, "contact_model": args.contact_model,
                "relative_breaking_threshold": int(local.params.relative_breaking_threshold),
                "warm_start": int(local.params.warm_start),
                "contact_order": int(local.params.contact_order),
                "contact_overflow": {"substeps_on_streamed_rows": overflow[0], "points_without_rows": overflow[1],
                                     "link_link_or_obstacle_without_rows": overflow[2]},
                # contact points per physics substep (snk_contact_histogram) over HIST_STEPS further env-steps, counted after
                # the timed region: what sizes the register-resident solve's 64 slots
                "contacts_per_substep": headline_hist,
                "self_collision": args.self_collision if NL == 32 else
                ("flag on: a substep with a pair of links within the breaking threshold goes through the streamed-row solve "
                 "(none under the gait: contact_overflow.substeps_on_streamed_rows)" if args.self_collision else 0),
                "world_size": (dist.get_world_size() if dist is not None else 1),
                "backend": (dist.get_backend() if dist is not None else None),
                "ranks": ranks,
                # the entry point inside the timed loop: N = 1 DeviceVecEnv.step_packed (tensors in, one [obs | reward |
                # done] block out: snk_step_packed, the same C entry point the sharded path uses),
                # N > 1 ShardedVecEnv.step_block (the gathered [obs | reward | done] block, one D2H copy on the root)
                "api": "ShardedVecEnv.step_block" if world > 1 else "DeviceVecEnv.step_packed",
                "timed_region": "per step: H2D of the step's action block from pinned host memory (%d B), %sthe fused "
                                "env-step kernel, %sD2H of obs / reward / done to pinned host memory" % (
                                    world * E * A * 4, "RCCL scatter, " if world > 1 else "",
                                    "RCCL gather to rank 0, " if world > 1 else ""),
                "self_launched": bool(os.environ.get("SNK_BENCH_SELF_LAUNCHED")),
                "parallelism": "envs sharded over %d GPU(s), no data-path collective; "
                               "RCCL actions scatter + obs/reward/done gather to rank 0" % world
                               if world > 1 else "1 GPU, one wavefront per env",
            },
            "substeps_per_s": substeps / elapsed,
            "mean_substeps_per_env_step": substeps / n_env_steps,
            "roofline": roofline_block(NL, E, K, local_sub, kernel_ms, kcount, cfg_key),
            "actions_resident": resident,
            "cpu_baseline": cpu,
            "variants": variants,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
