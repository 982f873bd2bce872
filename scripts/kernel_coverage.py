"""Which compiled kernel instantiations did a run launch?  

    python scripts/kernel_coverage.py [--log gpurun_out/launch_log.tsv] [--lib mmsbm_amd/libmmsbm_hip.so]
                                      [--out profiles/r6_kernel_coverage.csv] [--fail-on-missing]

compiled set = the kernels' host stubs in the library's symbol table (`nm`: one `__device_stub__<kernel>` per
instantiation hipcc emitted; rocPRIM's own kernels are left out: library code of the one-off layout sorts);
launched set = the launch log the library appends when MMSBM_HIP_LAUNCH_LOG is set (tests/conftest.py sets it): one line
per (process, kernel) with the launch count and the test id at the kernel's first launch.  Writes a CSV: kernel, launches,
processes, first test -- and lists what was compiled but never launched.
"""
import argparse
import collections
import csv
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXXFILT = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"


def demangle(names):
    tool = CXXFILT if os.path.exists(CXXFILT) else "c++filt"
    res = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True)
    return res.stdout.split("\n")[:len(names)]


def canon(name: str) -> str:
    """A demangled kernel name without its argument list, host-stub marker and namespaces of ours."""
    name = name.strip()
    name = re.sub(r"^void ", "", name)
    name = name.replace("__device_stub__", "")
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):     # cut the argument list: the first '(' outside template brackets ...
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0 and not name.startswith("(anonymous namespace)", i):
            cut = i
            break
    name = name[:cut]
    for ns in ("(anonymous namespace)::", "mmsbm::gpu_layout::", "mmsbm_hip_impl::"):
        name = name.replace(ns, "")
    return name.replace(" ", "")


def compiled_kernels(lib):
    out = subprocess.run(["nm", lib], capture_output=True, text=True, check=True).stdout
    syms = sorted({ln.split()[-1] for ln in out.splitlines() if "__device_stub__" in ln})
    names = [canon(d) for d in demangle(syms)]
    return sorted({n for n in names if "rocprim" not in n})


def launched_kernels(log):
    rows = []
    with open(log) as fh:
        for ln in fh:
            parts = ln.rstrip("\n").split("\t")
            if len(parts) >= 4:
                rows.append((parts[0], int(parts[1]), parts[2], parts[3]))
    names = demangle([r[2] for r in rows]) if rows else []
    agg = collections.OrderedDict()
    for (pid, cnt, _, tag), nm in zip(rows, names):
        k = canon(nm)
        a = agg.setdefault(k, {"launches": 0, "pids": set(), "first": tag})
        a["launches"] += cnt
        a["pids"].add(pid)
        if a["first"] in ("-", "") and tag not in ("-", ""):
            a["first"] = tag
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log", default=os.path.join(ROOT, "gpurun_out", "launch_log.tsv"))
    ap.add_argument("--lib", default=os.path.join(ROOT, "mmsbm_amd", "libmmsbm_hip.so"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r6_kernel_coverage.csv"))
    ap.add_argument("--fail-on-missing", action="store_true")
    args = ap.parse_args()
    compiled = compiled_kernels(args.lib)
    launched = launched_kernels(args.log)
    sys.path.insert(0, ROOT)
    from mmsbm_amd.build import built_id
    missing = [k for k in compiled if k not in launched]
    foreign = [k for k in launched if k not in compiled and "rocprim" not in k]
    with open(args.out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["# build_id", built_id(), "compiled", len(compiled), "launched", len(compiled) - len(missing),
                    "never_launched", len(missing)])
        w.writerow(["kernel", "launches", "processes", "first_test"])
        for k in compiled:
            a = launched.get(k)
            w.writerow([k, a["launches"] if a else 0, len(a["pids"]) if a else 0, a["first"] if a else "NEVER LAUNCHED"])
    print(f"compiled own kernels: {len(compiled)}; launched: {len(compiled) - len(missing)}; never launched: {len(missing)}")
    for k in missing:
        print("  never launched:", k)
    for k in foreign:
        print("  launched but not in the library's symbol table:", k)
    print("written:", args.out)
    if args.fail_on_missing and missing:
        sys.exit(1)


if __name__ == "__main__":
    main()
