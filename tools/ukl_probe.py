"""Developer tool: phase time stamps of update_kl_fast (component 0) from an experiment build of the library:
  hipcc ... -DGMMVI_UKL_STAMPS -c gmmvi_amd/csrc/update_kl.hip ; link as gmmvi_amd/libgmmvi_hip_stamps.so ;
  GMMVI_HIP_LIB=gmmvi_amd/libgmmvi_hip_stamps.so python tools/ukl_probe.py [D] [K]
wall_clock64 ticks are 100 MHz (10 ns)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gmmvi_amd import hip_ops
from gmmvi_amd.device import get_context
from test_hip_kernels import _update_inputs
d = int(sys.argv[1]) if len(sys.argv) > 1 else 50
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = get_context()
rng = np.random.default_rng(0)
m, hs, gs = _update_inputs(rng, k, d)
acc = np.zeros(7)
reps = 5
for _ in range(reps):
    means, chols = ctx.asarray(m.means), ctx.asarray(m.chol_cov)
    out = hip_ops.update_components_kl(ctx, means, chols, ctx.asarray(hs), ctx.asarray(gs), ctx.asarray(np.full(k, 0.1)), 1.0, 1e-12,
                                       ctx.asarray(np.full(k, -1.0)), ctx.asarray(np.full(k, 1e-12)), ctx.asarray(np.zeros(k)),
                                       want_info=True, want_packed=True)
    acc += out[1].numpy()[:7]
names = ["-", "load+products", "householder", "search", "factor+backsubst", "pack", "inverse+fragments+end"]
prev = 0.0
for i in range(1, 7):
    t = acc[i] / reps * 10e-3
    print(f"{names[i]:24s} {t - prev:8.1f} us   (cumulative {t:8.1f})")
    prev = t
