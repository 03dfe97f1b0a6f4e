"""Deterministic synthetic clips: frames, injected mesh motion and global homographies.

Everything is derived from a 32-bit integer hash of the element index, so the same inputs come out
on any machine, NumPy version or device (NumPy `Generator` streams are not stable across versions).
The shapes mirror what the reference's front-end hands to the hot path
(`/root/reference/meshflowstabilizer.py`, `mfs.py:N`):

  * frames: uint8 BGR (F, H, W, 3)                                   -- mfs.py:172-213
  * vertex_unstabilized_displacements: float64 (F, R+1, C+1, 2), [0] = 0; each step is a float32
    velocity added in float64                                         -- mfs.py:268-282, 354-361
  * homographies: float64 (F, 3, 3), last one identity                -- mfs.py:273-274
"""
import numpy as np

_M32 = 0xFFFFFFFF


def hash32(idx, seed=0):
    """murmur3-style finaliser on int64 arrays holding values < 2^32 (low 32 bits are kept)."""
    z = (np.asarray(idx, dtype=np.int64) + np.int64((int(seed) * 0x9E3779B1) & _M32)) & _M32
    z = ((z ^ (z >> 16)) * 0x85EBCA6B) & _M32
    z = ((z ^ (z >> 13)) * 0xC2B2AE35) & _M32
    z = z ^ (z >> 16)
    return z


def uniform01(idx, seed=0):
    """Uniform in (0, 1), 32-bit resolution."""
    return (hash32(idx, seed).astype(np.float64) + 0.5) / 4294967296.0


def normal(idx, seed=0):
    """Standard normal (Box-Muller on two hashed uniforms)."""
    idx = np.asarray(idx, dtype=np.int64)
    u1 = uniform01(idx * 2, seed)
    u2 = uniform01(idx * 2 + 1, seed)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def _tri(v, p):
    return np.abs((v % (2 * p)) - p)


def frames_numpy(num_frames, height, width, seed=0, kind='pattern', first_frame=0):
    """uint8 (F, H, W, 3).  kind='pattern': two moving triangle waves + 4-bit hash noise;
    kind='noise': every byte an independent hash (worst case for interpolation parity).
    Integer arithmetic only, so `frames_torch` reproduces it bit for bit on a device."""
    f = (np.arange(num_frames, dtype=np.int64) + first_frame)[:, None, None, None]
    y = np.arange(height, dtype=np.int64)[None, :, None, None]
    x = np.arange(width, dtype=np.int64)[None, None, :, None]
    c = np.arange(3, dtype=np.int64)[None, None, None, :]
    idx = (((f * height + y) * width + x) * 3 + c) & _M32
    h = hash32(idx, seed)
    if kind == 'noise':
        return (h & 255).astype(np.uint8)
    if kind != 'pattern':
        raise ValueError(kind)
    base = _tri(3 * x + 2 * y + 7 * f + 40 * c, 128) + _tri(5 * y - x + 3 * f + 4096, 96)
    return np.minimum(base + (h & 15), 255).astype(np.uint8)


def frames_torch(num_frames, height, width, device, seed=0, kind='pattern', first_frame=0):
    """Same bytes as `frames_numpy`, generated on `device` with torch int64 ops (one frame at a time)."""
    import torch
    out = torch.empty((num_frames, height, width, 3), dtype=torch.uint8, device=device)
    y = torch.arange(height, dtype=torch.int64, device=device)[:, None, None]
    x = torch.arange(width, dtype=torch.int64, device=device)[None, :, None]
    c = torch.arange(3, dtype=torch.int64, device=device)[None, None, :]
    sd = (int(seed) * 0x9E3779B1) & _M32
    yx = (y * width + x) * 3 + c
    if kind == 'pattern':
        b0 = 3 * x + 2 * y + 40 * c
        b1 = 5 * y - x + 4096
    for i in range(num_frames):
        f = i + first_frame
        idx = (f * height * width * 3 + yx) & _M32
        z = (idx + sd) & _M32
        z = ((z ^ (z >> 16)) * 0x85EBCA6B) & _M32
        z = ((z ^ (z >> 13)) * 0xC2B2AE35) & _M32
        z = z ^ (z >> 16)
        if kind == 'noise':
            out[i] = (z & 255).to(torch.uint8)
        elif kind == 'pattern':
            base = (((b0 + 7 * f) % 256) - 128).abs() + (((b1 + 3 * f) % 192) - 96).abs()
            out[i] = torch.clamp(base + (z & 15), max=255).to(torch.uint8)
        else:
            raise ValueError(kind)
    return out


def motion(num_frames, mesh_rows, mesh_cols, seed=0, translation_sigma=3.0, field_sigma=1.0,
           jitter_sigma=0.0):
    """(vertex_unstabilized_displacements float64 (F,R+1,C+1,2), homographies float64 (F,3,3)).

    Per-frame vertex velocity = global translation N(0, translation_sigma) + a spatially smooth field
    (three low-frequency cosines, rms ~ field_sigma) + optional iid jitter; velocities are rounded to
    float32 and accumulated in float64 with displacement[0] = 0 (mfs.py:271, 281, 359-361).
    Homographies = identity + that frame's translation + N(0, 0.01) on the 2x2 block; the last one is
    the identity (mfs.py:274)."""
    F, R1, C1 = num_frames, mesh_rows + 1, mesh_cols + 1
    t = np.arange(F - 1, dtype=np.int64)
    trans = translation_sigma * np.stack([normal(t, seed * 16 + 1), normal(t, seed * 16 + 2)], axis=-1)   # (F-1, 2)
    rr = (np.arange(R1, dtype=np.float64) / max(mesh_rows, 1))[None, :, None, None]
    cc = (np.arange(C1, dtype=np.float64) / max(mesh_cols, 1))[None, None, :, None]
    field = np.zeros((F - 1, R1, C1, 2))
    for k in range(3):
        comp = np.arange(2, dtype=np.int64)[None, :]
        base = (t[:, None] * 2 + comp) * 8 + k                                                   # (F-1, 2)
        amp = field_sigma * 0.8 * normal(base, seed * 16 + 3)[:, None, None, :]
        ph = 2 * np.pi * uniform01(base, seed * 16 + 4)[:, None, None, :]
        fr = 0.5 + 1.5 * uniform01(base, seed * 16 + 5)[:, None, None, :]
        fc = 0.5 + 1.5 * uniform01(base, seed * 16 + 6)[:, None, None, :]
        field += amp * np.cos(2 * np.pi * (fr * rr + fc * cc) + ph)
    vel = trans[:, None, None, :] + field
    if jitter_sigma > 0:
        n = np.arange(vel.size, dtype=np.int64).reshape(vel.shape)
        vel = vel + jitter_sigma * normal(n, seed * 16 + 7)
    vel = vel.astype(np.float32)
    disp = np.zeros((F, R1, C1, 2), dtype=np.float64)
    for i in range(F - 1):
        disp[i + 1] = disp[i] + vel[i]                                                            # mfs.py:281
    hom = np.tile(np.identity(3), (F, 1, 1))
    if F > 1:
        n4 = np.arange((F - 1) * 4, dtype=np.int64).reshape(F - 1, 2, 2)
        hom[:-1, :2, :2] += 0.01 * normal(n4, seed * 16 + 8)
        hom[:-1, 0, 2] = trans[:, 0]
        hom[:-1, 1, 2] = trans[:, 1]
    return disp, hom


def features(num_frames, height, width, homographies, seed=0, per_pair=(600, 900), residual_sigma=1.5,
             noise_sigma=0.4):
    """Matched features per frame pair, in the shape and dtype the reference's tracker hands on (mfs.py:518-528,
    578: float64 (K, 1, 2) arrays whose values are float32 positions plus integer sub-frame offsets):
    a list of F-1 `(early, late)` tuples.  `early` is uniform over the frame; `late` = the pair's homography
    applied to it + a smooth residual field (rms ~ residual_sigma px) + iid noise, rounded to float32."""
    F = num_frames
    lo, hi = per_pair
    out = []
    for t in range(F - 1):
        k = lo + int(hash32(np.array([t]), seed * 16 + 9)[0]) % max(hi - lo + 1, 1)
        i = np.arange(k, dtype=np.int64) + t * 65536
        ex = (uniform01(i * 2, seed * 16 + 10) * (width - 1)).astype(np.float32).astype(np.float64)
        ey = (uniform01(i * 2 + 1, seed * 16 + 10) * (height - 1)).astype(np.float32).astype(np.float64)
        m = np.asarray(homographies[t], dtype=np.float64).reshape(9)
        w = ex * m[6] + ey * m[7] + m[8]
        gx = (ex * m[0] + ey * m[1] + m[2]) / w
        gy = (ex * m[3] + ey * m[4] + m[5]) / w
        ph = 2 * np.pi * uniform01(t * 4 + np.arange(4, dtype=np.int64), seed * 16 + 11)
        rx = residual_sigma * 1.4 * np.cos(2 * np.pi * (0.9 * ex / width + 0.6 * ey / height) + ph[0]) * np.cos(ph[1])
        ry = residual_sigma * 1.4 * np.cos(2 * np.pi * (0.5 * ex / width + 1.1 * ey / height) + ph[2]) * np.cos(ph[3])
        lx = (gx + rx + noise_sigma * normal(i * 2, seed * 16 + 12)).astype(np.float32).astype(np.float64)
        ly = (gy + ry + noise_sigma * normal(i * 2 + 1, seed * 16 + 12)).astype(np.float32).astype(np.float64)
        out.append((np.stack([ex, ey], axis=-1)[:, None, :], np.stack([lx, ly], axis=-1)[:, None, :]))
    return out


def clip(num_frames, height, width, mesh_rows=16, mesh_cols=16, seed=0, kind='pattern', **motion_kw):
    """Convenience: (frames uint8 (F,H,W,3), displacements, homographies)."""
    disp, hom = motion(num_frames, mesh_rows, mesh_cols, seed=seed, **motion_kw)
    return frames_numpy(num_frames, height, width, seed=seed, kind=kind), disp, hom
