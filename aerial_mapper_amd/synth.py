"""Seeded synthetic inputs for the DSM + backward-grid orthomosaic hot path.

The reference ships no dataset (its demo data is an external download,
install/test_aerial_mapper:28), so every test and benchmark in this repo runs
on the generators below.  numpy versions are used by the
tests (small sizes, host memory); the torch versions build the large bench
workloads directly in HBM.

Conventions
  points   (N,3) float64, AoS x,y,z  -- the memory layout of
           std::vector<Eigen::Vector3d, Eigen::aligned_allocator<...>>
  poses    (F,7) float64: tx,ty,tz,qw,qx,qy,qz (T_G_B, Hamilton quaternion),
           the reference's pose text format (aerial-mapper-io.cc:103-121)
  frames   (F,H,W) uint8 (8UC1) or (F,H,W,3) uint8 (8UC3, OpenCV BGR order)
"""
import math

import numpy as np

_M64 = (1 << 64) - 1


def terrain_height(x, y):
    """Smooth synthetic terrain, ~400 m +- 10 m."""
    return 400.0 + 10.0 * np.sin(0.01 * x) * np.cos(0.01 * y)


def make_points(n, half_extent, seed, noise=0.05, center=(0.0, 0.0)):
    """n points uniform on [-half_extent, half_extent]^2 around `center`
    (x=easting, y=northing) on the synthetic terrain + U(-noise, noise)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    xy = rng.uniform(-half_extent, half_extent, size=(n, 2))
    z = terrain_height(xy[:, 0], xy[:, 1]) + rng.uniform(-noise, noise, size=n)
    pts = np.empty((n, 3), np.float64)
    pts[:, 0] = xy[:, 0] + center[0]
    pts[:, 1] = xy[:, 1] + center[1]
    pts[:, 2] = z
    return pts


class Mt19937_64(object):
    """std::mt19937_64 (the generator SURVEY.md 8d names for the cfg inputs), vectorised over its
    312-word state.  canonical(n) = what libstdc++'s std::generate_canonical<double, 53> makes of
    one draw each: (double)u / 2^64, a result of 1.0 replaced by the largest double below it --
    so uniform(a, b) reproduces std::uniform_real_distribution<double>(a, b)(engine) draw by draw."""
    _N, _M = 312, 156
    _A = np.uint64(0xB5026F5AA96619E9)
    _UP, _LO = np.uint64(0xFFFFFFFF80000000), np.uint64(0x7FFFFFFF)

    def __init__(self, seed):
        mt = np.empty(self._N, np.uint64)
        x = int(seed) & _M64
        for i in range(self._N):
            mt[i] = x
            x = (6364136223846793005 * (x ^ (x >> 62)) + i + 1) & _M64
        self._mt, self._left = mt, np.empty(0, np.uint64)

    def _twist(self):
        mt, N, M = self._mt, self._N, self._M
        one = np.uint64(1)

        def mix(cur, nxt, far):
            y = (cur & self._UP) | (nxt & self._LO)
            return far ^ (y >> one) ^ np.where((y & one) != 0, self._A, np.uint64(0))
        mt[:N - M] = mix(mt[:N - M], mt[1:N - M + 1], mt[M:])          # old words only
        mt[N - M:N - 1] = mix(mt[N - M:N - 1], mt[N - M + 1:], mt[:M - 1])  # far = words just made
        mt[N - 1:] = mix(mt[N - 1:], mt[:1], mt[M - 1:M])
        y = mt.copy()
        y ^= (y >> np.uint64(29)) & np.uint64(0x5555555555555555)
        y ^= (y << np.uint64(17)) & np.uint64(0x71D67FFFEDA60000)
        y ^= (y << np.uint64(37)) & np.uint64(0xFFF7EEE000000000)
        y ^= y >> np.uint64(43)
        return y

    def draws(self, n):
        """the next n outputs of operator()"""
        parts, have = [self._left], self._left.size
        while have < n:
            parts.append(self._twist())
            have += self._N
        allw = np.concatenate(parts)
        self._left = allw[n:]
        return allw[:n]

    def canonical(self, n):
        v = self.draws(n).astype(np.float64) * 2.0 ** -64    # (uint64 -> double rounds to nearest)
        return np.minimum(v, np.nextafter(1.0, 0.0))

    def uniform(self, a, b, n):
        return self.canonical(n) * (b - a) + a


def make_points_cfg1(n=1_000_000, half_extent=500.0, seed=42, noise=0.05):
    """BASELINE.json configs[0] / SURVEY.md 8d cfg1: std::mt19937_64(seed); per point, in this order,
    x, y ~ U(-half_extent, half_extent) and z = 400 + 10 sin(0.01 x) cos(0.01 y) + U(-noise, noise)."""
    u = Mt19937_64(seed).canonical(3 * n).reshape(n, 3)
    pts = np.empty((n, 3), np.float64)
    pts[:, 0] = u[:, 0] * (2.0 * half_extent) + (-half_extent)
    pts[:, 1] = u[:, 1] * (2.0 * half_extent) + (-half_extent)
    pts[:, 2] = terrain_height(pts[:, 0], pts[:, 1]) + (u[:, 2] * (2.0 * noise) + (-noise))
    return pts


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays."""
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_M64)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(_M64)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(_M64)
    return z ^ (z >> np.uint64(31))


def make_frames(num_frames, height, width, channels=1, salt=0):
    """High-frequency frames: pixel(f,v,u,c) = low byte of
    splitmix64(((salt*4+c) << 56) + (f << 32) + (v << 16) + u), so any 1-px sampling error
    changes the value."""
    f = np.arange(num_frames, dtype=np.uint64)[:, None, None]
    v = np.arange(height, dtype=np.uint64)[None, :, None]
    u = np.arange(width, dtype=np.uint64)[None, None, :]
    with np.errstate(over="ignore"):
        base = (f << np.uint64(32)) + (v << np.uint64(16)) + u
        if channels == 1:
            key = base + (np.uint64(salt * 4) << np.uint64(56))
            return (splitmix64(key) & np.uint64(0xFF)).astype(np.uint8)
        out = np.empty((num_frames, height, width, channels), np.uint8)
        for c in range(channels):
            key = base + (np.uint64(salt * 4 + c + 1) << np.uint64(56))
            out[..., c] = (splitmix64(key) & np.uint64(0xFF)).astype(np.uint8)
        return out


def _qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx])


def _axis_angle(axis, ang):
    s = math.sin(0.5 * ang)
    return np.array([math.cos(0.5 * ang), axis[0] * s, axis[1] * s, axis[2] * s])


def make_lawnmower_poses(num_frames, half_extent, altitude, seed,
                         tilt_deg=5.0, center=(0.0, 0.0), lines=None,
                         yaw_offset=math.pi / 2.0):
    """Nadir-looking lawn-mower flight over [-half_extent, half_extent]^2.

    Camera convention (pinhole, z forward): a level camera has
    R_G_C = Rz(yaw + yaw_offset) * Rx(pi), i.e. optical axis pointing at -z
    (down); with the default yaw_offset the long image side lies across track.
    Roll and pitch ~ U(-tilt, tilt) degrees are applied in the camera frame.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    if lines is None:
        lines = max(1, int(round(math.sqrt(num_frames / 2.0))))
    per_line = int(math.ceil(num_frames / lines))
    poses = np.zeros((num_frames, 7), np.float64)
    k = 0
    for ln in range(lines):
        y = -half_extent + (ln + 0.5) * (2.0 * half_extent / lines)
        forward = (ln % 2 == 0)
        for s in range(per_line):
            if k >= num_frames:
                break
            frac = (s + 0.5) / per_line
            x = (-half_extent + frac * 2.0 * half_extent) * (1.0 if forward else -1.0)
            yaw = 0.0 if forward else math.pi
            q = _axis_angle((0.0, 0.0, 1.0), yaw + yaw_offset)
            q = _qmul(q, _axis_angle((1.0, 0.0, 0.0), math.pi))
            roll = math.radians(rng.uniform(-tilt_deg, tilt_deg))
            pitch = math.radians(rng.uniform(-tilt_deg, tilt_deg))
            q = _qmul(q, _axis_angle((1.0, 0.0, 0.0), roll))
            q = _qmul(q, _axis_angle((0.0, 1.0, 0.0), pitch))
            q = q / np.linalg.norm(q)
            poses[k, 0:3] = (x + center[0], y + center[1], altitude)
            poses[k, 3:7] = q
            k += 1
    return poses


IDENTITY_POSE = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])


# ---------------------------------------------------------------------------
# torch (device-side) generators for the bench workloads
# ---------------------------------------------------------------------------
def make_points_torch(n, half_extent, seed, device, noise=0.05, center=(0.0, 0.0)):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    pts = torch.empty((n, 3), dtype=torch.float64, device=device)
    if isinstance(half_extent, (tuple, list)):     # (half extent in x, in y)
        half_extent = torch.tensor(half_extent, dtype=torch.float64, device=device)
    xy = (torch.rand((n, 2), dtype=torch.float64, device=device, generator=g) * 2.0 - 1.0) * half_extent
    z = 400.0 + 10.0 * torch.sin(0.01 * xy[:, 0]) * torch.cos(0.01 * xy[:, 1])
    z += (torch.rand(n, dtype=torch.float64, device=device, generator=g) * 2.0 - 1.0) * noise
    pts[:, 0] = xy[:, 0] + center[0]
    pts[:, 1] = xy[:, 1] + center[1]
    pts[:, 2] = z
    return pts


def make_frames_torch(num_frames, height, width, channels, seed, device):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    shape = (num_frames, height, width) if channels == 1 else (num_frames, height, width, channels)
    return torch.randint(0, 256, shape, dtype=torch.uint8, device=device, generator=g)
