"""Turns a gpurun_out/profN directory (rocprofv3 kernel trace + separate PMC passes of
bench.py) into the summaries committed under profiles/.  Usage:
    python tools/summarize_prof.py gpurun_out/prof_c32 r02_c32 --links 32 --round 2 \
        --workload "4096 envs x 32-link snake, serpenoid gait" --command "python3 bench.py --links 32 ..."
The tag's middle part is the configuration key bench.py looks for (profiles/r*_<key>_pmc_summary.json):
c16, c32, c16_fric.  Expects SRC/trace, SRC/pmc_fetch, SRC/pmc_write and (optionally) SRC/pmc_sq.

Round 6: every figure is a mean over the TIMED launches only.  The passes run `bench.py --profile`
(W warm-up + K timed launches of the step kernel and nothing else: no histogram pass, no second timed region), W and K
are read from the bench line the traced run printed (SRC/trace.log), launches W .. W + K - 1 of the per-dispatch trace
(kernel_trace.csv, not rocprofv3's --stats average over all launches) are averaged, and the roofline fraction is
recomputed from the trace beside the bench line's own."""
import argparse, collections, csv, glob, json, os, shutil

ap = argparse.ArgumentParser()
ap.add_argument("src")
ap.add_argument("tag")
ap.add_argument("--links", type=int, default=16)
ap.add_argument("--round", type=int, default=2)
ap.add_argument("--kernel", default=None, help="default env_step_sched_kernel<LINKS>")
ap.add_argument("--workload", default="4096 envs x 16-link snake, serpenoid gait")
ap.add_argument("--command", default="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline")
ap.add_argument("--skip", type=int, default=None, help="leading launches left out of the per-launch means "
                "(default: the warm-up count of the bench line in SRC/trace.log)")
ap.add_argument("--hbm-peak", type=float, default=8.0e12)
a = ap.parse_args()
src, tag, N = a.src, a.tag, a.links
K = a.kernel or 'env_step_sched_kernel<%d,' % N     # <N, rows-in-registers?>
RK = 'reset_kernel<%d,' % N


def one(pattern, required=True):
    g = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)       # gpurun_out/ keeps earlier runs' files:
    assert g or not required, pattern                                               # the newest one is this run's
    return g[-1] if g else None


def agg(path, kern):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if kern in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
    return d


def bench_line(log):
    """the JSON line bench.py printed inside the traced run"""
    if not os.path.exists(log):
        return None
    for ln in open(log, errors="replace"):
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            try:
                return json.loads(ln)
            except ValueError:
                pass
    return None


line = bench_line(os.path.join(src, "trace.log"))
W = a.skip if a.skip is not None else (line["warmup"] if line else 2)
Kt = line["steps"] if line else None
stats = one('trace/*/*kernel_stats.csv')
shutil.copy(stats, 'profiles/%s_kernel_stats.csv' % tag)
all_avg = None
for r in csv.DictReader(open(stats)):
    if K in r['Name']:
        all_avg = float(r['AverageNs']) * 1e-6
# the per-dispatch trace: launches in dispatch order
dur, plan = [], []
for r in sorted(csv.DictReader(open(one('trace/*/*kernel_trace.csv'))), key=lambda r: int(r['Start_Timestamp'])):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
    if K in r['Kernel_Name']:
        dur.append(d)
    elif 'plan_sched_kernel<%d>' % N in r['Kernel_Name']:
        plan.append(d)
timed = dur[W:W + Kt] if Kt else dur[W:]
timed_plan = plan[W:W + Kt] if Kt else plan[W:]
assert timed, (len(dur), W, Kt)
if Kt is not None:
    assert len(dur) == W + Kt, "the traced run launched %d step kernels, its bench line says %d + %d: not a --profile run?" % (len(dur), W, Kt)
kavg = sum(timed) / len(timed)
pavg = sum(timed_plan) / len(timed_plan) if timed_plan else 0.0
a.skip = W
fpath, wpath = one('pmc_fetch/*/*counter_collection.csv'), one('pmc_write/*/*counter_collection.csv')
f, w = agg(fpath, K), agg(wpath, K)
fr, wr = agg(fpath, RK), agg(wpath, RK)
fs, ws = f['FETCH_SIZE'][a.skip:], w['WRITE_SIZE'][a.skip:]
assert len(f['FETCH_SIZE']) == len(dur) == len(w['WRITE_SIZE']), (len(f['FETCH_SIZE']), len(w['WRITE_SIZE']), len(dur))
fetch_kb, write_kb = sum(fs) / len(fs), sum(ws) / len(ws)
rec_kb = 4096 * (256 if N == 16 else 512) / 1024.0
out = {
    "round": a.round, "tag": tag, "kernel": "snk::" + K.rstrip(",") + (", ...>" if K.endswith(",") else ""),
    "workload": a.workload,
    # (--command may or may not carry the profiler's prefix already: it is printed once)
    "command": "%s%s%s" % ("" if a.command.lstrip().startswith("rocprofv3") else "rocprofv3 --pmc <counters> --output-format csv -- ",
                           a.command, "" if "separate passes" in a.command else
                           "  (FETCH_SIZE, WRITE_SIZE and the SQ set in separate passes; kernel trace in its own pass)"),
    "kernel_trace_average_ms": kavg,
    "kernel_trace_note": "mean over the TIMED launches of the traced run only (launches %d .. %d of %d; bench.py --profile: "
                         "no histogram pass, no second timed region); rocprofv3's own --stats average over all %d launches "
                         "is kept as kernel_stats_average_ms_all_launches" % (W, W + len(timed) - 1, len(dur), len(dur)),
    "n_launches_averaged": len(timed), "kernel_trace_min_ms": min(timed), "kernel_trace_max_ms": max(timed),
    "plan_kernel_average_ms": pavg,
    "kernel_stats_average_ms_all_launches": all_avg,
    "calibration": {"kernel": "snk::" + RK.rstrip(",") + ", ...>", "known_read_KB": rec_kb, "FETCH_SIZE_KB": fr['FETCH_SIZE'],
                    "known_write_KB_record_only": rec_kb, "WRITE_SIZE_KB": wr['WRITE_SIZE'],
                    "conclusion": "FETCH_SIZE reads 1/2 of the bytes of this 4-B-per-lane coalesced record load "
                                  "(gfx950 rule of MI355X_MICROARCH.md); WRITE_SIZE exact"},
    "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
    "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
}
qpath = one('pmc_sq/*/*counter_collection.csv', required=False)
if qpath:
    q = agg(qpath, K)
    sq = {k: sum(v[a.skip:]) / len(v[a.skip:]) for k, v in q.items()}
    out["sq_per_launch"] = sq
    out["derived"] = {"valu_insts_per_wave": sq['SQ_INSTS_VALU'] / sq['SQ_WAVES'],
                      "valu_active_fraction_of_wave_cycles": sq['SQ_ACTIVE_INST_VALU'] / sq['SQ_WAVE_CYCLES'],
                      "mean_wave_residency_fraction_of_kernel": (sq['SQ_WAVE_CYCLES'] * 4 / sq['SQ_WAVES']) / (sq['GRBM_GUI_ACTIVE'] / 8),
                      "clock_GHz_from_GRBM": sq['GRBM_GUI_ACTIVE'] / 8 / (kavg * 1e-3) / 1e9 if kavg else None}
if line:
    rf = line.get("roofline") or {}
    alg = rf.get("algorithmic_bytes_per_launch")
    out["bench_line_of_the_traced_run"] = {"value": line.get("value"), "ms_per_step": line.get("ms_per_step"),
                                           "steps": line.get("steps"), "warmup": line.get("warmup"),
                                           "kernel_ms_hip_events": rf.get("kernel_ms"), "roofline_frac": rf.get("frac"),
                                           "algorithmic_bytes_per_launch": alg,
                                           "mean_substeps_per_env_step": line.get("mean_substeps_per_env_step")}
    if alg:
        # the same quotient as bench.py's roofline.frac, from the tracer's durations (step kernel + its plan kernel: what the
        # HIP events bracket)
        out["roofline_frac_from_trace"] = alg / ((kavg + pavg) * 1e-3) / a.hbm_peak
        out["roofline_frac_from_trace_step_kernel_alone"] = alg / (kavg * 1e-3) / a.hbm_peak
json.dump(out, open('profiles/%s_pmc_summary.json' % tag, 'w'), indent=1)
print(json.dumps({k: out.get(k) for k in ("kernel_trace_average_ms", "n_launches_averaged", "kernel_trace_min_ms", "kernel_trace_max_ms",
                                          "plan_kernel_average_ms", "hbm_bytes_per_launch", "roofline_frac_from_trace",
                                          "bench_line_of_the_traced_run", "derived")}, indent=1))
