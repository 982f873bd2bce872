#!/usr/bin/env python3
"""Condense the raw rocprofv3 CSVs of scripts/pmc_mem.sh (gpurun_out/pmcmem_<tag>_<pass>/) into the committed
profiles/<tag>_mem.csv: mean per-launch value of every memory-path counter for the iteration's kernels, plus the
derived figures DESIGN.md quotes (lines per gathered row, L2 read hit rate, mean L1->L2 read latency, lines in
flight per CU).  usage: scripts/summarize_mem.py <tag>"""
import collections, csv, glob, os, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = (("seg_pass_kernel", "seg_pass_kernel"), ("pair_block_kernel<false", "pair_stage(T+S)"), ("pair_block_kernel<true", "pair_stage(A)"),
           ("pair_mfma_kernel<false", "pair_stage(T+S)"), ("pair_mfma_kernel<true", "pair_stage(A)"), ("eta_p_kernel", "eta_p_kernel"),
           ("pairs_fused_kernel", "pairs_fused_kernel"), ("tail_fused_kernel", "tail_fused_kernel"), ("lik_wave_kernel", "lik_wave_kernel"))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "gpurun_out", f"pmcmem_{tag}_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        for key, nm in KERNELS:
            if key in r["Kernel_Name"]:
                agg[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = os.path.join(root, "profiles", f"{tag}_mem.csv")
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "counter", "mean_per_launch", "launches_sampled"])
    for nm in sorted(agg):
        c = {k: sum(v) / len(v) for k, v in agg[nm].items()}
        for k in sorted(c):
            w.writerow([nm, k, f"{c[k]:.1f}", len(agg[nm][k])])
        d = {}
        if c.get("TCP_TCC_READ_REQ_sum"):
            d["l2_read_hit_rate (1 - TCC_EA0_RDREQ / TCP_TCC_READ_REQ)"] = 1 - c.get("TCC_EA0_RDREQ_sum", 0) / c["TCP_TCC_READ_REQ_sum"]
            d["mean_l1_to_l2_read_latency_cycles"] = c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / c["TCP_TCC_READ_REQ_sum"]
        if c.get("TCC_EA0_RDREQ_sum") is not None:
            d["fabric_read_bytes (TCC_EA0_RDREQ x 128 B; 32-B requests: TCC_EA0_RDREQ_32B)"] = c["TCC_EA0_RDREQ_sum"] * 128 - c.get("TCC_EA0_RDREQ_32B_sum", 0) * 96
        if c.get("TCC_CYCLE_sum") and c.get("TCC_EA0_RDREQ_LEVEL_sum"):
            d["fabric_reads_in_flight_per_l2_channel (RDREQ_LEVEL / TCC_CYCLE x 16 channels x 8 XCDs)"] = c["TCC_EA0_RDREQ_LEVEL_sum"] / c["TCC_CYCLE_sum"]
        if c.get("TCC_CYCLE_sum") and c.get("TCC_BUSY_sum") is not None:
            d["l2_busy_fraction (TCC_BUSY / TCC_CYCLE)"] = c["TCC_BUSY_sum"] / c["TCC_CYCLE_sum"]
        if c.get("TCC_REQ_sum") and c.get("TCC_CYCLE_sum"):
            d["l2_requests_per_channel_cycle (TCC_REQ / TCC_CYCLE)"] = c["TCC_REQ_sum"] / c["TCC_CYCLE_sum"]
        if c.get("TCP_UTCL1_REQUEST_sum"):
            d["utcl1_miss_rate"] = c.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0) / c["TCP_UTCL1_REQUEST_sum"]
        for k, v in d.items():
            w.writerow([nm, "derived: " + k, f"{v:.6g}", ""])
print(open(out).read()) if "-q" not in sys.argv else print(out)
