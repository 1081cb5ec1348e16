"""Developer tool: cost of growing a mapped range chunk by chunk (gmmvi_vmm_grow), GPU idle and GPU busy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gmmvi_amd.device import get_context
from gmmvi_amd.optimization.sample_db import _MappedRange
ctx = get_context()
for busy in (False, True):
    r = _MappedRange(ctx, 64 << 30)
    big = ctx.zeros((64 << 20,)) if busy else None
    ts = []
    for c in range(1, 161):
        if busy:                                    # ~1 ms of queued fills in front of every call
            for _ in range(8):
                ctx.check(ctx.lib.gmmvi_fill_f32(ctx.handle, big.ptr, 1.0, big.size))
        t0 = time.perf_counter()
        r.ensure(c * r.chunk)
        ts.append((time.perf_counter() - t0) * 1e6)
    ctx.sync()
    ts = np.array(ts)
    print(f"GPU {'busy' if busy else 'idle'}: chunk {r.chunk >> 20} MiB; call 1-10 {ts[:10].mean():.0f} us, 71-80 {ts[70:80].mean():.0f} us, "
          f"151-160 {ts[150:].mean():.0f} us, max {ts.max():.0f} us")
    del r
