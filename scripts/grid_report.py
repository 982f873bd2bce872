#!/usr/bin/env python3
"""The report on synthetic grids (north_star; the reference's benchmark_mmsbm.py:60-73
prints its numbers and keeps nothing): runs bench.py -- the same program the driver runs -- once per row ON THE GPU BOX
and writes profiles/<tag>_grid.json (every row's full bench line) + profiles/<tag>_grid.md (the table README links).

    python scripts/grid_report.py r6            # all rows (about 4 minutes on one MI355X)
    python scripts/grid_report.py r6 --rows c3,c3x8

Rows: BASELINE's configs C1, C2, C3, C5; C3 with 8 restarts advancing as slots of one context; K = L = 16 / 20 / 24 / 32
at C3's size; C5 with its pair stage on the vector ALUs beside the matrix cores (the stated deviation from north_star's
"no MFMA": both numbers from ONE run of this script).  Columns: it/s (the driver's definition: K steps between fences)
and the steady-state rate, algorithmic GB/s (B_read x it/s), % of the 8.0 TB/s HBM peak and of the 6.29 TB/s
measured copy ceiling, the dominant kernel's counter traffic in GB/s (profiles/pmc_summary.json, when taken with these
sources), fp64-VALU %, the CPU restatement's it/s (cores) with its calibration against the real reference, build id.
Every row carries the build id of the library that ran: the report is valid when all equal HEAD's source id.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ROWS = [
    # key, label, bench.py arguments
    ("c1", "C1: 100 ratings, K=2, L=4", ["--config", "c1", "--steps", "1000", "--warmup", "50"]),
    ("c2", "C2: 100k ratings, K=L=10", ["--config", "c2", "--steps", "1000", "--warmup", "50", "--cpu-iters", "20"]),
    ("c3", "C3: 1M ratings, K=L=20 (headline; the driver's 20 steps)", ["--config", "c3", "--steps", "20", "--warmup", "5"]),
    ("c3long", "C3, 1,000 steps", ["--config", "c3", "--steps", "1000", "--warmup", "50", "--no-cpu-baseline"]),
    ("c3x8", "C3, 8 restarts as slots of one context", ["--config", "c3", "--steps", "200", "--warmup", "20",
                                                        "--batched-restarts", "8", "--no-cpu-baseline"]),
    ("c3k16", "C3 size, K=L=16", ["--config", "c3", "--groups", "16", "--steps", "1000", "--warmup", "50", "--no-cpu-baseline"]),
    ("c3k20", "C3 size, K=L=20", ["--config", "c3", "--groups", "20", "--steps", "1000", "--warmup", "50", "--no-cpu-baseline"]),
    ("c3k24", "C3 size, K=L=24", ["--config", "c3", "--groups", "24", "--steps", "1000", "--warmup", "50", "--no-cpu-baseline"]),
    ("c3k32", "C3 size, K=L=32", ["--config", "c3", "--groups", "32", "--steps", "1000", "--warmup", "50", "--no-cpu-baseline"]),
    ("c5", "C5: 10M ratings, K=L=50, pair stage on the matrix cores", ["--config", "c5", "--steps", "100", "--warmup", "10",
                                                                       "--steady-steps", "200"]),
    ("c5valu", "C5, pair stage on the vector ALUs (mfma = 0)", ["--config", "c5", "--steps", "100", "--warmup", "10",
                                                                 "--steady-steps", "200", "--mfma", "0", "--no-cpu-baseline"]),
]


def run_row(args_list, timeout):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args_list
    t0 = time.time()
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    if res.returncode != 0 or not lines:
        raise RuntimeError(f"{' '.join(cmd)} failed (rc {res.returncode}):\n{res.stderr[-2000:]}")
    out = json.loads(lines[-1])
    out["_command"] = "python bench.py --gpus 1 " + " ".join(args_list)
    out["_wall_s"] = round(time.time() - t0, 1)
    return out


def fmt(x, nd=1):
    return "n/a" if x is None else f"{x:,.{nd}f}"


def table(rows, head):
    md = ["| row | GPUs | it/s (timed region) | it/s (median of 5 × ≥ 50 iterations) | it/s steady | µs / iteration | algorithmic GB/s | % of 8.0 TB/s | % of 6.29 TB/s | "
          "dominant kernel: µs, counter traffic GB/s | fp64 VALU % | CPU it/s (cores; port / reference) | speed-up | pair stage | build id |",
          "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for key, label, line in rows:
        it = line["iteration"]
        st = line.get("steady_state") or {}
        rf = line["roofline"]
        cpu = line.get("cpu_baseline")
        traffic = None if rf.get("traffic") is None else rf["traffic"] / (rf["avg_launch_us"] * 1e-6) / 1e9
        rate = st.get("value") or line["value"]
        gbps = it["algorithmic_read_bytes"] * rate / 1e9
        por = cpu.get("port_over_reference") if cpu else None
        cpu_txt = "n/a" if not cpu else f"{cpu['value']:.3g} ({cpu['cores']}; {'n/a' if por is None else format(por, '.3f')})"
        md.append(f"| {label} | {line['n_gpus']} | {fmt(line['value'], 0)} | {fmt((line.get('median_of_5') or {}).get('value'), 0)} | {fmt(st.get('value'), 0)} | {fmt(1e6 / rate, 2)} | {fmt(gbps, 0)} | "
                  f"{100 * gbps / 8000:.1f} | {100 * gbps / 6290:.1f} | {rf['kernel']}: {fmt(rf['avg_launch_us'], 1)}, {fmt(traffic, 0)} | "
                  f"{100 * it['fp64_valu']['frac_of_peak']:.1f} | {cpu_txt} | {fmt(line.get('gpu_over_cpu'), 0)} | {line['config']['pair_stage'].split(' (')[0]} | "
                  f"`{line['library']['build_id']}` |")
        if line.get("batched_restarts"):
            b = line["batched_restarts"]
            gb = it["algorithmic_read_bytes"] * b["value"] / 1e9
            md.append(f"| &nbsp;&nbsp;↳ {b['slots']} restarts per launch (restart-iterations/s, HIP events) | 1 | | | {fmt(b['value'], 0)} | "
                      f"{fmt(b['us_per_restart_iteration'], 2)} | {fmt(gb, 0)} | {100 * gb / 8000:.1f} | {100 * gb / 6290:.1f} | | | | | | |")
    return head + "\n\n" + "\n".join(md) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag", nargs="?", default="r6")
    ap.add_argument("--rows", default="", help="comma-separated row keys (default: all)")
    ap.add_argument("--timeout", type=int, default=600)
    args = ap.parse_args()
    from mmsbm_amd.build import ensure_library, source_id
    ensure_library()
    want = [k for k in args.rows.split(",") if k] or [r[0] for r in ROWS]
    done = []
    for key, label, cmd in ROWS:
        if key not in want:
            continue
        print(f"[grid] {key}: {label}", flush=True)     # (a progress line per row: the GPU box kills silent commands)
        done.append((key, label, run_row(cmd, args.timeout)))
    sid = source_id()
    path = os.path.join(ROOT, "profiles", f"{args.tag}_grid.json")
    if args.rows and os.path.exists(path):      # some rows again: the others are kept if they are of the same sources
        with open(path) as fh:
            old = json.load(fh)
        if old.get("source_id") == sid:
            fresh = {k: (k, lb, ln) for k, lb, ln in done}
            kept = {r["key"]: (r["key"], r["label"], r["line"]) for r in old["rows"]}
            kept.update(fresh)
            done = [kept[r[0]] for r in ROWS if r[0] in kept]
    ids = sorted({line["library"]["build_id"] for _, _, line in done})
    head = (f"# Synthetic-grid report `{args.tag}`\n\nGenerated by `python scripts/grid_report.py {args.tag}` on "
            f"{done[0][2]['ranks'][0]['device_name']} (one GPU; every row is one run of `bench.py`, the commands are in "
            f"`{args.tag}_grid.json`).  Sources `{sid}`; libraries that ran: {', '.join('`' + i + '`' for i in ids)}"
            f"{'' if ids == [sid] else ' -- NOT the sources in the tree: rerun'}.  float64 throughout; the bound is HBM "
            "bandwidth (SURVEY 8(d): B_read = N(12+8K+8L) + 8KLR per iteration); `median of 5` = SURVEY 8(d)'s protocol (>= 50 "
            "iterations per repeat after warm-up, HIP events, median of 5 repeats); `it/s steady` = 200-1,000 iterations "
            "timed with HIP events after the timed region (the columns to its right use it when present); CPU = the numpy "
            "restatement of the reference's dense dataflow on this host (C5: its first 60,000 rows scaled by rows -- the "
            "reference's own dataflow is infeasible there, omega = 200 GB); speed-up = timed-region it/s / CPU it/s.")
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(path, "w") as fh:
        json.dump({"tag": args.tag, "source_id": sid, "rows": [{"key": k, "label": lb, "line": ln} for k, lb, ln in done]},
                  fh, indent=1)
    with open(os.path.join(ROOT, "profiles", f"{args.tag}_grid.md"), "w") as fh:
        fh.write(table(done, head))
    print(table(done, head))
    if ids != [sid]:
        sys.exit(1)


if __name__ == "__main__":
    main()
