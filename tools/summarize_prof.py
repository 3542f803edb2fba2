"""Turns a gpurun_out/profN directory (rocprofv3 kernel trace + separate PMC passes of
bench.py) into the summaries committed under profiles/.  Usage:
    python tools/summarize_prof.py gpurun_out/prof4 r01_v3 [kernel name, default env_step_sched_kernel<16>]"""
import collections, csv, glob, json, os, shutil, sys
src, tag = sys.argv[1], sys.argv[2]
def one(pattern):
    g = glob.glob(os.path.join(src, pattern))
    assert g, pattern
    return g[0]
def agg(path, kern):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if kern in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
    return d
K = sys.argv[3] if len(sys.argv) > 3 else 'env_step_sched_kernel<16>'   # sets up to r01_v6: env_step_kernel<16>
stats = one('trace/*/*kernel_stats.csv')
shutil.copy(stats, 'profiles/%s_kernel_stats.csv' % tag)
kavg = None
for r in csv.DictReader(open(stats)):
    if K in r['Name']:
        kavg = float(r['AverageNs']) * 1e-6
f = agg(one('pmc_fetch/*/*counter_collection.csv'), K); w = agg(one('pmc_write/*/*counter_collection.csv'), K)
q = agg(one('pmc_sq/*/*counter_collection.csv'), K)
fr = agg(one('pmc_fetch/*/*counter_collection.csv'), 'reset_kernel<16>'); wr = agg(one('pmc_write/*/*counter_collection.csv'), 'reset_kernel<16>')
fs, ws = f['FETCH_SIZE'][2:], w['WRITE_SIZE'][2:]
fetch_kb, write_kb = sum(fs) / len(fs), sum(ws) / len(ws)
sq = {k: sum(v[2:]) / len(v[2:]) for k, v in q.items()}
out = {
    "round": 1, "tag": tag, "kernel": "snk::" + K,
    "workload": "4096 envs x 16-link snake, serpenoid gait (bench.py --steps 10 --warmup 2 --no-cpu-baseline)",
    "command": "rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py ...  (FETCH_SIZE, WRITE_SIZE and the SQ set in separate passes; kernel trace in its own pass)",
    "kernel_trace_average_ms": kavg,
    "calibration": {"kernel": "snk::reset_kernel<16>", "known_read_KB": 1024.0, "FETCH_SIZE_KB": fr['FETCH_SIZE'],
                    "known_write_KB": [1024.0, 1920.0], "WRITE_SIZE_KB": wr['WRITE_SIZE'],
                    "conclusion": "FETCH_SIZE reads 1/2 of the bytes of this 4-B-per-lane coalesced record load (gfx950 rule of MI355X_MICROARCH.md); WRITE_SIZE exact"},
    "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
    "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
    "sq_per_launch": sq,
    "derived": {"valu_insts_per_wave": sq['SQ_INSTS_VALU'] / sq['SQ_WAVES'],
                "valu_active_fraction_of_wave_cycles": sq['SQ_ACTIVE_INST_VALU'] / sq['SQ_WAVE_CYCLES'],
                "mean_wave_residency_fraction_of_kernel": (sq['SQ_WAVE_CYCLES'] * 4 / sq['SQ_WAVES']) / (sq['GRBM_GUI_ACTIVE'] / 8),
                "clock_GHz_from_GRBM": sq['GRBM_GUI_ACTIVE'] / 8 / (kavg * 1e-3) / 1e9 if kavg else None},
}
json.dump(out, open('profiles/%s_pmc_summary.json' % tag, 'w'), indent=1)
print(json.dumps({k: out[k] for k in ("kernel_trace_average_ms", "hbm_bytes_per_launch", "derived")}, indent=1))
