"""Where the workgroups of pairs_fused_kernel spend their time (diagnostic build with -DMMSBM_STAMPS, see
scripts/phase_stamps.py).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DMMSBM_STAMPS -o /tmp/libstamps.so mmsbm_amd/csrc/unity.hip
    MMSBM_HIP_LIBRARY=/tmp/libstamps.so python scripts/phase_stamps_fused.py c2
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mmsbm_amd import MMSBM, _lib
from mmsbm_amd.synthetic import CONFIGS, synthetic_triples
n, u, i, r, k, l = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
train = synthetic_triples(n, u, i, r, 0)
mm = MMSBM(k, l, iterations=1, seed=0); mm._prepare_objects(train)
ctx = mm._ctx(0); ctx.init_params(mm.child_states[0]); ctx.set_option("fused", 1); ctx.iterate(3)
lib = _lib.load()
lib.mmsbm_hip_debug_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int]
names = ["ids, offsets, eta rows, theta prefetch -> LDS", "barrier", "A mat-vec + barrier", "A rows out + pair segments + barrier",
         "S + barrier", "T mat-vec + barrier", "T rows out + slab hand-over + slab store", "drain"]
for rep in range(3):
    ctx.time_stage(4, 1)   # stage 4 = pairs_fused_kernel: 3 warm launches + 1
    buf = np.zeros(8192 * 16, dtype=np.uint64)
    assert lib.mmsbm_hip_debug_stamps(buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size) == 0
    st = buf.reshape(8192, 16).astype(np.int64)
    live = st[:, 0] > 0
    where = buf.reshape(8192, 16)[live][:, 9]
    st = st[live][:, :9]
    rel = (st - st[:, 0].min()) / 100.0   # microseconds
    # placement: XCC id (high word) + the SE / SH / CU bits of HW_ID -> how many workgroups share a CU
    cu_key = ((where >> np.uint64(32)) << np.uint64(8)) | ((where >> np.uint64(8)) & np.uint64(0xff))
    keys, inv, cnt = np.unique(cu_key, return_inverse=True, return_counts=True)
    life = rel[:, 8] - rel[:, 0]
    print(f"-- placement: {len(keys)} CUs used; workgroups per CU: " +
          ", ".join(f"{c}: {int((cnt == c).sum())} CUs (lifetime mean {life[cnt[inv] == c].mean():.2f}, last end {rel[cnt[inv] == c, 8].max():.2f} us)"
                    for c in sorted(set(cnt))))
    per_xcc = np.bincount((where >> np.uint64(32)).astype(np.int64) & 15)
    print(f"   per XCC: {per_xcc.tolist()}")
    print(f"-- {len(st)} workgroups; last start {rel[:, 0].max():.2f}, first end {rel[:, 8].min():.2f}, last end {rel[:, 8].max():.2f} us")
    d = np.diff(rel, axis=1)
    for j in range(8):
        print(f"   {names[j]:52s} mean {d[:, j].mean():6.2f}  p10 {np.percentile(d[:, j], 10):6.2f}  p90 {np.percentile(d[:, j], 90):6.2f} us")
    print(f"   workgroup lifetime mean {(rel[:, 8] - rel[:, 0]).mean():.2f} us")

# the three roles of tail_fused_kernel (stage 5): lifetimes per role (blocks [0, nb_p) p_update, then the user segments, then item_sum)
per_u = 256 // (4 if k <= 16 else 8)
bu = -(-(ctx.n_users if not ctx.swapped else ctx.n_items) // per_u)
nb_p = -(-(-(-k // 4) * 4 * (-(-l // 4) * 4)) // 16)
for rep in range(2):
    ctx.time_stage(5, 1)
    buf = np.zeros(8192 * 16, dtype=np.uint64)
    assert lib.mmsbm_hip_debug_stamps(buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size) == 0
    st = buf.reshape(8192, 16).astype(np.int64)
    t0 = st[:nb_p + bu, 0].min()
    life = (st[:, 8] - st[:, 0]) / 100.0
    end = (st[:, 8] - t0) / 100.0
    per_i = 256 // (4 if l <= 16 else 8)
    n_blocks = bu + nb_p + -(-(ctx.n_items if not ctx.swapped else ctx.n_users) // per_i)   # (stale stamps of launch 1 lie beyond)
    for nm, a, b in (("p_update", 0, nb_p), ("user segments", nb_p, nb_p + bu), ("item_sum", bu + nb_p, n_blocks)):
        if b > a:
            print(f"   tail role {nm:14s} {b - a:4d} blocks: lifetime mean {life[a:b].mean():5.2f} max {life[a:b].max():5.2f} us, "
                  f"first start {((st[a:b, 0] - t0) / 100.0).min():5.2f}, last end {end[a:b].max():5.2f} us")
        cuts = {"p_update": ([0, 1, 2, 3, 4, 5, 8], ["slab ranges", "slab loads + adds -> LDS", "barrier", "tree (2 barriers) + numerators", "barrier", "normalise + stores + drain"]),
                "user segments": ([0, 1, 2, 3, 8], ["offsets (+ own row asked for)", "indices", "rows + weights + sums", "store + drain"])}.get(nm)
        if cuts and b > a:
            sel = st[a:b][:, cuts[0]]
            sel = sel[(sel > 0).all(axis=1)]
            d = np.diff(sel, axis=1) / 100.0
            print("      " + "; ".join(f"{n} {d[:, j].mean():.2f}" for j, n in enumerate(cuts[1])))
