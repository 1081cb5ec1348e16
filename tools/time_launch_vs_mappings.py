"""Developer tool: does the host cost of a launch / a device copy depend on how many VMM chunks are mapped?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gmmvi_amd.device import get_context
from gmmvi_amd.optimization.sample_db import _MappedRange
ctx = get_context()
small, small2 = ctx.zeros((1024,)), ctx.zeros((1024,))
r = _MappedRange(ctx, 200 << 30)
for n_chunks in (0, 50, 200, 400):
    r.ensure(n_chunks * r.chunk)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(2000):
        ctx.check(ctx.lib.gmmvi_fill_f32(ctx.handle, small.ptr, 1.0, small.size))
    t1 = time.perf_counter()
    for _ in range(2000):
        ctx.check(ctx.lib.gmmvi_copy(ctx.handle, small2.ptr, small.ptr, 4096))
    t2 = time.perf_counter()
    for _ in range(2000):
        ctx.check(ctx.lib.gmmvi_copy(ctx.handle, r.base if n_chunks else small2.ptr, small.ptr, 4096))
    t3 = time.perf_counter()
    ctx.sync()
    print(f"{n_chunks:4d} chunks mapped ({n_chunks * r.chunk >> 30} GiB): launch {(t1 - t0) / 2000 * 1e6:6.1f} us, copy {(t2 - t1) / 2000 * 1e6:6.1f} us, "
          f"copy into the mapped range {(t3 - t2) / 2000 * 1e6:6.1f} us (host issue time per call)")
