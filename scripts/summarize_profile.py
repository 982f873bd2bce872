#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs of scripts/profile_round.sh (under gpurun_out/) into the small,
committed summaries under profiles/:  <tag>_kernel_stats.csv (the --stats table, our kernels),
<tag>_pmc.csv (mean counter values per kernel) and pmc_summary.json (HBM bytes per launch,
corrected as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so the
read side is doubled; WRITE_SIZE is exact)."""
import csv, glob, hashlib, json, os, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
config = sys.argv[2] if len(sys.argv) > 2 else "c3"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

def short(name):
    for key, nm in (("seg_pass_kernel", "seg_pass_kernel"), ("pair_block_kernel<false", "pair_block_kernel(T+S)"),
                    ("pair_block_kernel<true", "pair_block_kernel(A)"), ("pair_quad_a_kernel", "pair_block_kernel(A)"),
                    # (K x L > 1024: the same two launches run pair_mfma_kernel; full_name keeps the kernel's own name)
                    ("pair_mfma_kernel<false", "pair_block_kernel(T+S)"), ("pair_mfma_kernel<true", "pair_block_kernel(A)"),
                    ("eta_p_kernel", "eta_p_kernel"), ("eta_p_w4_kernel", "eta_p_kernel"),   # (many rounds: the 256-thread form of the same stage)
                    ("pairs_fused_kernel", "pairs_fused_kernel"), ("tail_fused_kernel", "tail_fused_kernel"),
                    ("lik_wave_kernel", "lik_wave_kernel"), ("lik_lane_kernel", "lik_lane_kernel"),
                    ("seg_combine_both_kernel", "seg_combine_both_kernel"), ("theta_log_pairs_kernel", "theta_log_pairs_kernel"),
                    ("seg_combine_small_kernel", "seg_combine_small_kernel"),
                    ("seg_combine_kernel", "seg_combine_kernel"), ("likelihood_fast_kernel", "likelihood_fast_kernel"), ("log_table_kernel", "log_table_kernel"),
                    ("init_rows_kernel", "init_rows_kernel"), ("likelihood_units_kernel", "likelihood_units_kernel"),
                    ("likelihood_kernel", "likelihood_kernel"), ("prod_dist_kernel", "prod_dist_kernel")):
        if key in name:
            return nm
    return None

newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1:]
stats = newest(os.path.join(src, f"prof_{tag}_trace", "*", "*_kernel_stats.csv"))
rows = []
if stats:
    for r in csv.DictReader(open(stats[0])):
        nm = short(r["Name"])
        if nm:
            rows.append({"kernel": nm, "full_name": r["Name"], "calls": r["Calls"],
                         "avg_us": f'{float(r["AverageNs"]) / 1e3:.2f}', "min_us": f'{float(r["MinNs"]) / 1e3:.2f}',
                         "max_us": f'{float(r["MaxNs"]) / 1e3:.2f}', "total_ms": f'{float(r["TotalDurationNs"]) / 1e6:.3f}',
                         "percent": r["Percentage"]})
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for part in ("fetch", "write", "l2", "sq", "lds_a", "lds_b", "mfma"):
    for f in newest(os.path.join(src, f"prof_{tag}_{part}", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            nm = short(r["Kernel_Name"])
            if nm:
                agg[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(dst, f"{tag}_pmc.csv"), "w", newline="") as fh:
    w = csv.writer(fh); w.writerow(["kernel", "counter", "mean_per_launch", "launches_sampled"])
    for nm in sorted(agg):
        for cn in sorted(agg[nm]):
            v = agg[nm][cn]; w.writerow([nm, cn, f"{sum(v) / len(v):.1f}", len(v)])

def kernel_source_sha16():  # the same identity bench.py computes: a profile is only valid for these sources
    sys.path.insert(0, root)
    from mmsbm_amd.build import source_id
    return source_id()

summary_path = os.path.join(dst, "pmc_summary.json")
summary = json.load(open(summary_path)) if os.path.exists(summary_path) else {}
summary[config] = {}
summary.setdefault("_meta", {})[config] = {"tag": tag, "kernel_source_sha16": kernel_source_sha16(),
                                            "stats_file": f"profiles/{tag}_kernel_stats.csv",
                                            "pmc_file": f"profiles/{tag}_pmc.csv"}
avg_us = {r["kernel"]: float(r["avg_us"]) for r in rows}
for nm, cs in agg.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        fetch = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024.0
        write = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024.0
        ent = {"fetch_size_bytes_raw": fetch, "write_size_bytes": write,
               "hbm_bytes_per_launch": 2.0 * fetch + write,
               "source": f"profiles/{tag}_pmc.csv: 2 x FETCH_SIZE (gfx950 tallies 128-B requests at 64 B) + WRITE_SIZE, KiB -> bytes"}
        if "TCC_HIT_sum" in cs:
            h = sum(cs["TCC_HIT_sum"]) / len(cs["TCC_HIT_sum"]); m = sum(cs["TCC_MISS_sum"]) / len(cs["TCC_MISS_sum"])
            ent["l2_hit_rate"] = h / (h + m) if h + m else None
        ent["avg_us"] = avg_us.get(nm)
        for key in ("SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS",
                    "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU",
                    "SQ_ACTIVE_INST_ANY", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64"):
            if key in cs:
                ent[key] = sum(cs[key]) / len(cs[key])
        summary[config][nm] = ent
json.dump(summary, open(summary_path, "w"), indent=1, sort_keys=True)
for r in rows:
    print(r["kernel"], r["calls"], r["avg_us"])
print(json.dumps(summary[config], indent=1))
