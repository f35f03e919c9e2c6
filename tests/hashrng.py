"""Host replica of the dropout counter hash in hri-emo_amd/csrc/common.h (fmix32 / site_key / keep16),
so tests can build the exact keep-masks the kernels use and check dropout paths exactly."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _u(x):
    return np.asarray(x, dtype=np.uint64) & M32


def fmix32(x):
    x = _u(x)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & M32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & M32
    x ^= x >> np.uint64(16)
    return x


def site_key(seed, site, c):
    seed = int(seed)
    k = fmix32(np.uint64((seed & 0xFFFFFFFF) ^ 0x9E3779B9))
    k = fmix32((k + np.uint64(((seed >> 32) * 0x85EBCA77) & 0xFFFFFFFF) + np.uint64((site * 0x27D4EB2F) & 0xFFFFFFFF)) & M32)
    return fmix32((k + ((_u(c) * np.uint64(0x165667B1)) & M32)) & M32)


def thr16(p):
    t = int(p * 65536.0 + 0.5)
    return max(0, min(65535, t))


def mix24(x):
    """the kernels' per-element mixer (24-bit multiplies), see csrc/common.h"""
    x = _u(x)
    m24 = np.uint64(0xFFFFFF)
    x ^= x >> np.uint64(16)
    x = ((x & m24) * np.uint64(0xB2AE35)) & M32
    x ^= x >> np.uint64(13)
    x = ((x & m24) * np.uint64(0xEBCA6B)) & M32
    x ^= x >> np.uint64(15)
    return x


def keep(key32, a, b, p):
    """boolean keep-mask for integer arrays a, b (broadcast): one hash per PAIR of adjacent b, low/high 16 bits"""
    b = _u(b)
    x = mix24((_u(key32) + ((_u(a) * np.uint64(0x9E3779B1)) & M32) + (((b >> np.uint64(1)) * np.uint64(0x85EBCA77)) & M32)) & M32)
    u16 = np.where((b & np.uint64(1)) == 1, x >> np.uint64(16), x & np.uint64(0xFFFF))
    return u16 >= np.uint64(thr16(p))


def inv_keep(p):
    return 1.0 / (1.0 - thr16(p) / 65536.0)


def rows_mask(seed, site, M, N, p, row_offset=0):
    """keep-mask [M,N] of add_ln / dropout kernels"""
    k = site_key(seed, site, 0)
    r = np.arange(M, dtype=np.uint64)[:, None] + np.uint64(row_offset)
    c = np.arange(N, dtype=np.uint64)[None, :]
    return keep(k, r, c, p)


def attn_mask(seed, site, B, H, Lq, Lk, p, b_offset=0):
    """keep-mask [B,H,Lq,Lk] of the attention kernels"""
    out = np.empty((B, H, Lq, Lk), dtype=bool)
    q = np.arange(Lq, dtype=np.uint64)[:, None]
    kk = np.arange(Lk, dtype=np.uint64)[None, :]
    for b in range(B):
        for h in range(H):
            k32 = site_key(seed, site, (b_offset + b) * H + h)
            out[b, h] = keep(k32, q, kk, p)
    return out
