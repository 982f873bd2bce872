"""Phase borders of pair_mfma_kernel's workgroups (diagnostic build, see scripts/phase_stamps.py):
the per-unit stamps are those of the workgroup's LAST unit.
    MMSBM_HIP_LIBRARY=.../libmmsbm_hip_stamps.so python scripts/phase_stamps_mfma.py c5 1   # 1 = T+S, 3 = A"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import MMSBM, _lib
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
n, u, i, r, k, l = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c5"]
stage = int(sys.argv[2]) if len(sys.argv) > 2 else 1
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=1, seed=0); mm._prepare_objects(train)
ctx = mm._ctx(0); ctx.init_params(mm.child_states[0]); ctx.iterate(3)
assert ctx.get_option("mfma") == 1.0
lib = _lib.load()
lib.mmsbm_hip_debug_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int]
for rep in range(2):
    ctx.time_stage(stage, 1)
    buf = np.zeros(8192 * 16, dtype=np.uint64)
    assert lib.mmsbm_hip_debug_stamps(buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size) == 0
    st = buf.reshape(8192, 16).astype(np.int64)
    st = st[st[:, 0] > 0][:, :9]
    rel = (st - st[:, 0].min()) / 100.0
    life = rel[:, 8] - rel[:, 0]
    print(f"-- stage {stage}: {len(st)} workgroups; last start {rel[:, 0].max():.1f}, first end {rel[:, 8].min():.1f}, "
          f"last end {rel[:, 8].max():.1f} us; lifetime mean {life.mean():.2f} (p10 {np.percentile(life, 10):.2f}, p90 {np.percentile(life, 90):.2f})")
    names = ["(last unit) top barrier wait + LDS stores", "second barrier", "fetch issue", "S products", "T products",
             "T stores issued .. loop end", "slab + drain"]
    d = np.diff(rel[:, 1:], axis=1)
    for j, nm in enumerate(names):
        print(f"   {nm:44s} mean {d[:, j].mean():6.2f}  p10 {np.percentile(d[:, j], 10):6.2f}  p90 {np.percentile(d[:, j], 90):6.2f} us")
    print(f"   start .. last unit's top                    mean {(rel[:, 1] - rel[:, 0]).mean():6.2f} us")
