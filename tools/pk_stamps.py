"""Developer tool: phase time stamps of the packed density sweep (experiment build: tools/build_stamps.sh, then
GMMVI_HIP_LIB=gmmvi_amd/libgmmvi_hip_stamps.so python tools/pk_stamps.py [dual|post]) at the north-star shape: for every
workgroup its start / end on the 100 MHz wall clock, for every wave the stamps 0 start, 1 x tile staged, 2.. after each
component pass, 8 loop left, 9 maxima exchanged, 10 tree merge done, 11 results stored."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmmvi_amd.device import get_context
from gmmvi_amd import hip_ops

which = sys.argv[1] if len(sys.argv) > 1 else "dual"
K, D, N = int(os.environ.get("TK", 100)), int(os.environ.get("TD", 20)), int(os.environ.get("TN", 10000))
rng = np.random.default_rng(0)
ctx = get_context()
means_h = rng.normal(size=(K, D)) * 30
covs = np.stack([(lambda a: a @ a.T / D + 3.0 * np.eye(D))(rng.normal(size=(D, D)) * 4) for _ in range(K)])
means = ctx.asarray(means_h)
chols, _ = hip_ops.cholesky(ctx, ctx.asarray(covs))
packed, _ = hip_ops.pack_components(ctx, means, chols)
logw = ctx.asarray(np.full(K, -np.log(K)))
logw2 = ctx.asarray(np.log(rng.dirichlet(np.ones(K))))
x = ctx.asarray(rng.normal(size=(N, D)) * 30)
fn = (lambda: hip_ops.mixture_eval_dual(ctx, packed, logw, logw2, x, D)) if which == "dual" else \
     (lambda: hip_ops.mixture_eval(ctx, packed, logw, x, D, want_ld=True, want_lp=True))
for _ in range(20):
    fn()
ctx.sync()
tiles = (N + 127) // 128
ky = int(os.environ.get("GMMVI_ME_PK_KY", 0)) or max(1, min((K + 3) // 4, (2 * 256 + tiles // 2) // tiles))
kchunk = -(-K // ky); ky = -(-K // kchunk)
nwg = tiles * ky
wb = (ctypes.c_longlong * (2 * nwg))()
ctx.lib.gmmvi_debug_wg_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert ctx.lib.gmmvi_debug_wg_times(wb, 2 * nwg) == 0
t = np.array(wb[:], dtype=np.int64).reshape(nwg, 2)
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01
print(f"{which}: {nwg} workgroups ({tiles} tiles x {ky} chunks of {kchunk}); starts median {np.median(st):.2f} p90 {np.percentile(st, 90):.2f} "
      f"max {st.max():.2f} us; ends min {en.min():.2f} median {np.median(en):.2f} max {en.max():.2f}; life median {np.median(en - st):.2f} max {(en - st).max():.2f}")
nph = min(nwg, 1024)
pb = (ctypes.c_longlong * (nph * 16 * 16))()
ctx.lib.gmmvi_debug_wg_phases.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert ctx.lib.gmmvi_debug_wg_phases(pb, nph * 16 * 16) == 0
ph = np.array(pb[:], dtype=np.int64).reshape(nph, 16, 16)
rel = (ph - t[:nph, 0][:, None, None]) * 0.01
names = {0: "start", 1: "x staged", 2: "pass 1", 3: "pass 2", 4: "pass 3", 5: "pass 4", 8: "loop left", 9: "maxima", 10: "tree done", 11: "stored"}
for w in range(int(os.environ.get('GMMVI_ME_PK_NW', 8))):
    row = []
    for i in (0, 1, 2, 3, 4, 8, 9, 10, 11):
        v = rel[:, w, i]
        v = v[(v > -1) & (v < 1000)]
        row.append(f"{names[i]} {np.median(v):.2f}" if v.size else f"{names[i]} -")
    print(f"  wave {w}: " + "  ".join(row))
hb = (ctypes.c_ulonglong * nwg)()
ctx.lib.gmmvi_debug_wg_hw.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert ctx.lib.gmmvi_debug_wg_hw(hb, nwg) == 0
h = np.array(hb[:], dtype=np.uint64)
hw, xcc = (h & np.uint64(0xffffffff)).astype(np.int64), (h >> np.uint64(32)).astype(np.int64) & 0xf
place = xcc * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 12) & 0x1) * 50 + ((hw >> 8) & 0xf)
import collections
cnt = collections.Counter(place.tolist())
per_cu = np.array([cnt[q] for q in place.tolist()])
for c in sorted(set(per_cu.tolist())):
    sel = per_cu == c
    print(f"  workgroups on a CU shared by {c}: {sel.sum()}, life median {np.median((en - st)[sel]):.2f} max {(en - st)[sel].max():.2f} us, end median {np.median(en[sel]):.2f} max {en[sel].max():.2f}")
print("  distinct CUs used:", len(cnt))
