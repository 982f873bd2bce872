"""Where the workgroups of a pair-stage launch spend their time (diagnostic build with -DMMSBM_STAMPS:
thread 0 of every workgroup records the 100 MHz wall clock at its phase borders).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DMMSBM_STAMPS -o /tmp/libstamps.so mmsbm_amd/csrc/unity.hip
    MMSBM_HIP_LIBRARY=/tmp/libstamps.so python scripts/phase_stamps.py c3 1     # stage 1 = T+S, 3 = A
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import MMSBM, _lib
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
n, u, i, r, k, l = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
stage = int(sys.argv[2]) if len(sys.argv) > 2 else 1
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=1, seed=0); mm._prepare_objects(train)
ctx = mm._ctx(0); ctx.init_params(mm.child_states[0]); ctx.iterate(3)
lib = _lib.load()
lib.mmsbm_hip_debug_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int]
for rep in range(3):
    ctx.time_stage(stage, 1)   # 3 warm launches + 1
    buf = np.zeros(8192 * 16, dtype=np.uint64)
    assert lib.mmsbm_hip_debug_stamps(buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size) == 0
    st = buf.reshape(8192, 16).astype(np.int64)
    used = st[:, 0] > 0
    st = st[used][:, :9]
    t0 = st[:, 0].min()
    rel = (st - t0) / 100.0   # microseconds
    names = ["start", "desc+barrier", "rowid issued/ready", "C rows -> LDS + barrier", "eta rows -> LDS + barrier",
             "S", "mat-vec", "out copy issued", "slab + drain"]
    print(f"-- stage {stage}, {used.sum()} workgroups; launch span: first start 0, last start {rel[:, 0].max():.2f}, "
          f"first end {rel[:, 8].min():.2f}, last end {rel[:, 8].max():.2f} us")
    d = np.diff(rel, axis=1)
    for j in range(8):
        print(f"   {names[j + 1]:32s} mean {d[:, j].mean():6.2f}  p10 {np.percentile(d[:, j], 10):6.2f}  p90 {np.percentile(d[:, j], 90):6.2f} us")
    print(f"   workgroup lifetime mean {(rel[:, 8] - rel[:, 0]).mean():.2f} us")
