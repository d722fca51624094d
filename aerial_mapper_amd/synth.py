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
