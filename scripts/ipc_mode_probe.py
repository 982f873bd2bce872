"""Does HSA_ENABLE_IPC_MODE_LEGACY=0 matter on this pool?  ("What's weak" 6b.)

RCCL's intra-node transport and torch's CUDA-tensor sharing both pass device memory between processes through
hipIpcGetMemHandle / hipIpcOpenMemHandle.  One GPU is enough to see whether that works: the parent allocates a
device buffer, a spawned child opens it and reads it back.  Run once per setting, each in a fresh interpreter
(the variable is read when the HSA runtime starts):

    python scripts/ipc_mode_probe.py            # both settings, one child interpreter each
    python scripts/ipc_mode_probe.py --one      # the current environment only

Prints one JSON line per setting: {"HSA_ENABLE_IPC_MODE_LEGACY": "0" | null, "ok": bool, "error": "..."}.
"""
import json
import os
import subprocess
import sys


def child(q_in, q_out):
    try:
        t = q_in.get(timeout=60)
        q_out.put(("ok", float(t.sum().item())))
    except Exception as e:  # noqa: BLE001 - report whatever the runtime says
        q_out.put(("error", f"{type(e).__name__}: {e}"))


def one():
    import torch
    import torch.multiprocessing as mp
    setting = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    rec = {"HSA_ENABLE_IPC_MODE_LEGACY": setting, "ok": False, "error": None}
    try:
        ctx = mp.get_context("spawn")
        q_in, q_out = ctx.Queue(), ctx.Queue()
        p = ctx.Process(target=child, args=(q_in, q_out))
        p.start()
        t = torch.arange(1024, dtype=torch.float64, device="cuda:0")
        q_in.put(t)      # -> hipIpcGetMemHandle in the parent, hipIpcOpenMemHandle in the child
        kind, val = q_out.get(timeout=120)
        p.join(30)
        if kind == "ok":
            rec["ok"] = val == float(1023 * 1024 / 2)
            if not rec["ok"]:
                rec["error"] = f"wrong sum {val}"
        else:
            rec["error"] = val
    except Exception as e:  # noqa: BLE001
        rec["error"] = f"{type(e).__name__}: {e}"
    print(json.dumps(rec), flush=True)


def main():
    if "--one" in sys.argv:
        one()
        return
    for setting in ("0", None):
        env = dict(os.environ)
        env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
        if setting is not None:
            env["HSA_ENABLE_IPC_MODE_LEGACY"] = setting
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True,
                           text=True, timeout=300)
        out = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        print(out[-1] if out else json.dumps({"HSA_ENABLE_IPC_MODE_LEGACY": setting, "ok": False,
                                               "error": f"rc {r.returncode}: {r.stderr[-400:]}"}), flush=True)


if __name__ == "__main__":
    main()
