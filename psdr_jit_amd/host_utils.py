"""Host-side utility classes of the reference's Python surface that do not touch the renderer's hot path: Sampler
(src/core/sampler.cpp), DiscreteDistribution (src/core/pmf.cpp) and Bitmap::eval (src/core/bitmap.cpp:47-128), in numpy.
The kernels have their own implementations (csrc/hip/sampler.h, shade.h, csrc/common/envmath.h); these exist so that scripts written
against psdr_jit's bindings (psdr.cpp:181-219) keep running.  Arrays in, arrays out (the reference takes / returns drjit arrays)."""
import numpy as np

_U64 = np.uint64
_MULT = _U64(0x5851f42d4c957f2d)


def _tea64(v0, v1, rounds=4):
    """sample_tea_64 as the reference instantiates it: 64-bit lanes, 32-bit running sum (sampler.cpp:6-17,27)"""
    v0, v1 = v0.astype(_U64).copy(), v1.astype(_U64).copy()
    s = 0
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            s = (s + 0x9e3779b9) & 0xffffffff
            v0 = v0 + (((v1 << _U64(4)) + _U64(0xa341316c)) ^ (v1 + _U64(s)) ^ ((v1 >> _U64(5)) + _U64(0xc8013ea4)))
            v1 = v1 + (((v0 << _U64(4)) + _U64(0xad90777d)) ^ (v0 + _U64(s)) ^ ((v0 >> _U64(5)) + _U64(0x7e95761e)))
        return v0 + (v1 << _U64(32))


class Sampler:
    """psdr.Sampler: seed(seed_value[n]), next_1d(), next_2d() - one independent PCG32 stream per lane"""

    def __init__(self):
        self.m_ready = False
        self.m_sample_count = 0

    def _next_u32(self):
        with np.errstate(over="ignore"):
            old = self._state
            self._state = old * _MULT + self._inc
            xs = (((old >> _U64(18)) ^ old) >> _U64(27)).astype(np.uint32)
            rot = (old >> _U64(59)).astype(np.uint32)
            return (xs >> rot) | (xs << ((np.uint32(0) - rot) & np.uint32(31)))

    def seed(self, seed_value, lane_index=None):
        """seed_value[n]: one seed per lane (sampler.cpp:19-30).  lane_index (default arange(n)) lets a caller reproduce single lanes
        of a larger array"""
        sv = np.asarray(seed_value, dtype=_U64).reshape(-1)
        n = sv.size
        idx = np.arange(n, dtype=_U64) if lane_index is None else np.asarray(lane_index, dtype=_U64).reshape(-1)
        with np.errstate(over="ignore"):
            s = sv + _U64(0x853c49e6748fea9b)                   # PCG32_DEFAULT_STATE, sampler.cpp:24
            initstate, initseq = _tea64(s, idx), _tea64(idx, s)
            self._state = np.zeros(n, dtype=_U64)
            self._inc = (initseq << _U64(1)) | _U64(1)
            self._next_u32()
            self._state = self._state + initstate
            self._next_u32()
        self.m_sample_count, self.m_ready = n, True

    def next_1d(self):
        if not self.m_ready:
            raise RuntimeError("Sampler::next_1d: sampler is not seeded")
        bits = (self._next_u32() >> np.uint32(9)) | np.uint32(0x3f800000)
        return bits.view(np.float32) - np.float32(1.0)

    def next_2d(self):
        a = self.next_1d()
        return np.stack([a, self.next_1d()], axis=1)


class DiscreteDistribution:
    """psdr.DiscreteDistribution: init(pmf), sample(samples) -> (index, pmf[index] / sum), .sum, .pmf()"""

    def __init__(self):
        self.m_size, self.m_sum = 0, 0.0

    def init(self, pmf):
        p = np.asarray(pmf, dtype=np.float32).reshape(-1)
        if p.size == 0:
            raise RuntimeError("DiscreteDistribution: empty distribution!")
        self.m_size = p.size
        self.m_pmf = p
        self.m_sum = float(np.add.reduce(p.astype(np.float32), dtype=np.float32))
        self.m_cmf = np.cumsum(p.astype(np.float64)).astype(np.float32)          # host double accumulation, pmf.h:21-25
        self.m_pmf_normalized = p / np.float32(self.m_sum)

    @property
    def sum(self):
        return self.m_sum

    def pmf(self):
        return self.m_pmf_normalized

    def sample(self, samples):
        s = np.asarray(samples, dtype=np.float32).reshape(-1)
        if self.m_size == 1:
            return np.zeros(s.size, np.int32), np.ones(s.size, np.float32)
        t = s * np.float32(self.m_sum)
        idx = np.searchsorted(self.m_cmf[:self.m_size - 1], t, side="left").astype(np.int32)       # first i with !(cmf[i] < t)
        return idx, self.m_pmf[idx] / np.float32(self.m_sum)


def bitmap_eval(data, uv, flip_v=True, envmap_mode=False, uv_xf=(0.0, 1.0, 0.0, 0.0)):
    """Bitmap<C>::eval(uv, flip_v, envmap_mode) (bitmap.cpp:47-128); uv_xf = (rotate, scale, translate.x, translate.y).
    data: [H, W, C] or [H, W]; uv: [N, 2]; returns [N, C]."""
    a = np.asarray(data, dtype=np.float32)
    if a.ndim == 2:
        a = a[..., None]
    H, W, C = a.shape
    uv = np.asarray(uv, dtype=np.float32).reshape(-1, 2)
    if W == 1 and H == 1:
        return np.tile(a.reshape(1, C), (uv.shape[0], 1))
    if W < 2 or H < 2:
        raise RuntimeError("Bitmap: invalid resolution!")
    f = np.float32
    rot, scl, tx, ty = (f(q) for q in uv_xf)
    sr, cr = f(np.sin(rot)), f(np.cos(rot))
    u0, v0 = uv[:, 0] - f(0.5), uv[:, 1] - f(0.5)
    x, y = u0 * cr + v0 * sr + f(0.5), -u0 * sr + v0 * cr + f(0.5)
    if flip_v:
        y = -y
    off = f(-0.5) + scl * f(0.5)
    x, y = x * scl - off + tx, y * scl + off + ty
    if envmap_mode:
        x = x - np.float32(0.5 / W)
    x, y = x - np.floor(x), y - np.floor(y)
    x = x * np.float32(W if envmap_mode else W - 1)
    y = y * np.float32(H - 1)
    px, py = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    w1x, w1y = x - px.astype(np.float32), y - py.astype(np.float32)
    w0x, w0y = np.float32(1) - w1x, np.float32(1) - w1y
    flat = a.reshape(H * W, C)
    last = H * W - 1
    if envmap_mode:
        yw = np.minimum(py, H - 2) * W
        xp1 = (px + 1) % W
        i00, i10, i01, i11 = yw + px, yw + xp1, yw + px + W, yw + xp1 + W
    else:
        px, py = np.clip(px, 0, W - 2), np.clip(py, 0, H - 2)
        i00 = py * W + px
        i10, i01, i11 = i00 + 1, i00 + W, i00 + W + 1
    i00, i10, i01, i11 = (np.clip(i, 0, last) for i in (i00, i10, i01, i11))
    v0 = w0x[:, None] * flat[i00] + w1x[:, None] * flat[i10]
    v1 = w0x[:, None] * flat[i01] + w1x[:, None] * flat[i11]
    return w0y[:, None] * v0 + w1y[:, None] * v1
