"""Deterministic synthetic BGR24 inputs for the diff/threshold/pack path.

All generators are counter-based (a 32-bit integer hash of the byte index, the frame number and
the seed), so any frame can be produced independently, on the host with numpy or directly in HBM
with torch (``device="cuda"``) -- the two back ends give identical bytes.

  S0 refrand : both frames i.i.d. uniform 0..254, like tests/algorithms_benchmarks.cu:4-10
               (`rand() % 255`); ~84.6 % of the bytes are flagged (dense worst case for pack).
  S1 webcam  : smooth background + static texture, uniform [-8, 8] sensor noise on every byte
               (sub-threshold, exercises the negative feedback), a moving rectangle and 1.2 % salt
               bytes; P is ~1-6 % of N like the reference's webcam input
               (REPORT/report.tex:2250, :2594).
  S2 static  : cur == prev (P = 0).      S3 flip : cur = prev ^ 0x80 (P = N).
  S4 edge    : every (prev, cur) byte pair, 256 x 256 (all df including +-20/+-21 and wrap).
"""
import numpy as np

_M = 0xFFFFFFFF


class _NP:
    int64 = np.int64

    @staticmethod
    def arange(n, device=None):
        return np.arange(n, dtype=np.int64)

    @staticmethod
    def where(c, a, b):
        return np.where(c, a, b)

    @staticmethod
    def clip(x, lo, hi):
        return np.clip(x, lo, hi)

    @staticmethod
    def u8(x):
        return x.astype(np.uint8)


class _TH:
    def __init__(self):
        import torch
        self.t = torch
        self.int64 = torch.int64

    def arange(self, n, device=None):
        return self.t.arange(n, dtype=self.t.int64, device=device)

    def where(self, c, a, b):
        t = self.t
        if not t.is_tensor(a):
            a = t.full_like(b, a) if t.is_tensor(b) else t.tensor(a)
        if not t.is_tensor(b):
            b = t.full_like(a, b)
        return t.where(c, a, b)

    def clip(self, x, lo, hi):
        return self.t.clamp(x, lo, hi)

    def u8(self, x):
        return x.to(self.t.uint8)


def _xp(device):
    return _NP if device is None else _TH()


def hash32(x):
    """lowbias32 on an int64 array holding values < 2**32 (numpy or torch)."""
    x = x & _M
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M
    x = x ^ (x >> 16)
    return x


def _key(idx, t, seed):
    return hash32(hash32(idx) + ((t & 0xFFFF) * 0x9E3779B9 + seed * 0x85EBCA6B) % (1 << 32))


def refrand_frame(n, seed, device=None):
    """S0: n bytes uniform on 0..254."""
    xp = _xp(device)
    idx = xp.arange(n, device)
    return xp.u8(_key(idx, 0, seed) % 255)


def _scene(xp, idx, t, width, height):
    pix = idx // 3
    c = idx - pix * 3
    y = pix // width
    x = pix - y * width
    tex = (hash32(idx + 0x5BD1E995) >> 8) % 7
    bg = 40 + (x * 150) // width + (y * 40) // height + 5 * c + tex
    if t < 0:
        return bg
    rw, rh = width // 4, height // 4
    speed = max(1, width // 240)
    x0 = (t * speed) % (width - rw)
    y0 = height // 3
    inside = (x >= x0) & (x < x0 + rw) & (y >= y0) & (y < y0 + rh)
    rect = 120 + 10 * c + (x - x0) // 16
    return xp.where(inside, rect, bg)


def webcam_frame(t, width, height, seed=21, salt=0.012, noise=8, device=None):
    """S1: frame t (t = -1 gives the base frame: background only, no rectangle, no salt)."""
    xp = _xp(device)
    n = 3 * width * height
    idx = xp.arange(n, device)
    key = _key(idx, t + 1, seed)
    nz = (key & 0xFFFF) % (2 * noise + 1) - noise
    val = _scene(xp, idx, t, width, height) + nz
    if t >= 0 and salt > 0:
        is_salt = ((key >> 16) & 0xFFFF) < int(salt * 65536)
        sval = hash32(key ^ 0xABCDEF) & 0xFF
        val = xp.where(is_salt, sval, val)
    return xp.u8(xp.clip(val, 0, 255))


def webcam_stream(nframes, width, height, seed=21, start=0, device=None, **kw):
    """(base, frames[nframes, 3wh]) of the S1 stream starting at frame `start`."""
    base = webcam_frame(-1, width, height, seed, device=device, **kw)
    fr = [webcam_frame(start + t, width, height, seed, device=device, **kw) for t in range(nframes)]
    if device is None:
        return base, np.stack(fr)
    import torch
    return base, torch.stack(fr)


def static_pair(n, seed=3):
    a = refrand_frame(n, seed)
    return a.copy(), a


def flip_pair(n, seed=4):
    prev = refrand_frame(n, seed)
    return (prev ^ 0x80).astype(np.uint8), prev


def edge_strip(reps=1):
    """S4: (cur, prev) covering every byte pair; 65536 * reps bytes."""
    i = np.arange(65536 * reps, dtype=np.int64) % 65536
    prev = (i // 256).astype(np.uint8)
    cur = (i % 256).astype(np.uint8)
    return cur, prev
