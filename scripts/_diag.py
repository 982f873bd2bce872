"""Diagnostic builds for the measurement scripts: the product library carries neither phase stamps nor ablation
switches, so a script that needs them builds csrc/unity.hip with the switch into /tmp and points the binding at it
BEFORE mmsbm_amd is imported (MMSBM_HIP_LIBRARY)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def use_diagnostic_build(*defines):
    from mmsbm_amd.build import build_diagnostic, source_id
    out = f"/tmp/libmmsbm_diag_{'_'.join(d.lower() for d in defines)}_{source_id()}.so"
    if not os.path.exists(out):
        print(f"[diag] building {out} (-D{' -D'.join(defines)}; about a minute)", file=sys.stderr, flush=True)
        build_diagnostic(out, defines)
    os.environ["MMSBM_HIP_LIBRARY"] = out
    return out
