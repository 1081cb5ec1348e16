"""Philox4x32-10 counter-based RNG (Salmon et al., SC'11, "Parallel random numbers: as easy as 1, 2, 3").

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference draws its normals from
TensorFlow's stateful generator (models/full_cov_gmm.py:36-39 ``tf.random.normal((D, n))``;
models/gmm.py:134-137 ``tf.random.uniform``), which cannot be reproduced without TensorFlow.
The build therefore defines its own stream -- Philox4x32-10 keyed by the run seed and
indexed by (global sample number, dimension block, stream id) -- and the oracle restates it
here so that GPU and CPU draw *identical* bits.  Pinned by the Random123 known-answer
vectors in tests/test_oracle_philox.py.

Stream layout (shared with gmmvi_amd/csrc/philox.h):
    key     = (seed & 0xffffffff, seed >> 32)
    counter = (index & 0xffffffff, index >> 32, block, stream)
    words w0..w3 -> uniforms u_i = ((w_i >> 8) + 0.5) * 2**-24           (exact in fp32)
    normals  (n0, n1) = BoxMuller(u0, u1), (n2, n3) = BoxMuller(u2, u3)
    BoxMuller(a, b) = sqrt(-2 ln a) * (cos(2 pi b), sin(2 pi b))
    eps[index, 4*block + j] = n_j
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)

STREAM_COMPONENT_NORMALS = 0   # x = mu_k + L_k eps      (models/full_cov_gmm.py:36-39)
STREAM_CATEGORICAL = 1         # component choice        (models/gmm.py:134-137)
STREAM_MIXTURE_NORMALS = 2     # eps for GMM.sample()    (models/gmm.py:139-163)


def philox4x32_10(counter, key):
    """counter: uint32 array [..., 4]; key: uint32 array [..., 2] (broadcastable). Returns uint32 [..., 4]."""
    c = np.asarray(counter, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    c0, c1, c2, c3 = (c[..., i].astype(np.uint64) for i in range(4))
    k0 = np.broadcast_to(k[..., 0], c0.shape).astype(np.uint32)
    k1 = np.broadcast_to(k[..., 1], c0.shape).astype(np.uint32)
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        n0 = hi1 ^ c1 ^ k0.astype(np.uint64)
        n2 = hi0 ^ c3 ^ k1.astype(np.uint64)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        with np.errstate(over="ignore"):
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def _words(seed, index, block, stream):
    index = np.asarray(index, dtype=np.uint64)
    block = np.asarray(block, dtype=np.uint64)
    index, block = np.broadcast_arrays(index, block)
    ctr = np.stack([index & MASK32, index >> np.uint64(32), block,
                    np.full(index.shape, stream, dtype=np.uint64)], axis=-1).astype(np.uint32)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    key = np.array([seed & 0xFFFFFFFF, seed >> 32], dtype=np.uint32)
    return philox4x32_10(ctr, key)


def uniforms(seed, index, block, stream, dtype=np.float64):
    """4 uniforms in (0,1) per (index, block): shape [..., 4]."""
    w = _words(seed, index, block, stream)
    return (((w >> np.uint32(8)).astype(np.float64) + 0.5) * (1.0 / 16777216.0)).astype(dtype)


def normals(seed, first_index, n, dim, stream=STREAM_COMPONENT_NORMALS, dtype=np.float64):
    """Standard normals eps[n, dim] for global sample numbers first_index .. first_index+n-1."""
    nblk = (dim + 3) // 4
    idx = (np.uint64(first_index) + np.arange(n, dtype=np.uint64))[:, None]
    blk = np.arange(nblk, dtype=np.uint64)[None, :]
    u = uniforms(seed, idx, blk, stream, dtype=np.float64)          # [n, nblk, 4]
    out = np.empty((n, nblk, 4), dtype=np.float64)
    for a, b in ((0, 1), (2, 3)):
        r = np.sqrt(-2.0 * np.log(u[..., a]))
        t = 2.0 * np.pi * u[..., b]
        out[..., a] = r * np.cos(t)
        out[..., b] = r * np.sin(t)
    return out.reshape(n, nblk * 4)[:, :dim].astype(dtype)


def uniform01(seed, first_index, n, stream=STREAM_CATEGORICAL, dtype=np.float64):
    """One uniform per index (word 0 of block 0)."""
    idx = np.uint64(first_index) + np.arange(n, dtype=np.uint64)
    return uniforms(seed, idx, np.zeros(n, dtype=np.uint64), stream, dtype=dtype)[..., 0]
