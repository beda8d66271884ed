"""Synthetic RGB-D frames + trajectories: stand-ins for the datasets BASELINE.json names, which
are not available in the build container or on the GPU box.

The output formats are exactly what the reference's dataset front-end hands to InfiniTamDriver::UpdateView
(reference: src/DenseSLAM/Input.cpp:48-162, PrecomputedDepthProvider.cpp:30-68, InfiniTamDriver.cpp:84-110):
  rgba  : uint8  [H, W, 4]  (r, g, b, 255)          -- ITMUChar4Image as filled by CvToItm
  depth : int16  [H, W]     millimetres, 0 = invalid -- ITMShortImage as filled by CvToItm
Frames are closed-form ray casts of boxes / spheres, so both machines generate identical inputs.
"""
import numpy as np


# ---------------------------------------------------------------------------------------------------
# primitives
# ---------------------------------------------------------------------------------------------------
class Box:
    def __init__(self, lo, hi, inside=False):
        self.lo, self.hi, self.inside = np.asarray(lo, np.float64), np.asarray(hi, np.float64), inside

    def hit(self, o, d, inv_d):
        # slab test per axis on 2-D arrays; o is the (single) camera centre so (lo - o) is a scalar per axis.
        # fmin/fmax ignore the NaNs of 0 * inf (ray parallel to a slab it starts on).
        tmin = tmax = None
        with np.errstate(invalid="ignore"):
            for a in range(3):
                t1 = (self.lo[a] - o[a]) * inv_d[a]
                t2 = (self.hi[a] - o[a]) * inv_d[a]
                lo_t, hi_t = np.fmin(t1, t2), np.fmax(t1, t2)
                tmin = lo_t if tmin is None else np.fmax(tmin, lo_t)
                tmax = hi_t if tmax is None else np.fmin(tmax, hi_t)
        if self.inside:  # camera inside the room: the far intersection is the wall
            t = np.where((tmax > 0) & (tmin <= tmax), tmax, np.inf)
        else:
            t = np.where((tmin > 0) & (tmin <= tmax), tmin, np.inf)
        return t


class Sphere:
    def __init__(self, c, r):
        self.c, self.r = np.asarray(c, np.float64), float(r)
        self.lo, self.hi = self.c - self.r, self.c + self.r

    def hit(self, o, d, inv_d):
        oc = o - self.c
        a = np.sum(d * d, axis=-1)
        b = 2.0 * np.sum(oc * d, axis=-1)
        c = float(np.sum(oc * oc) - self.r * self.r)
        disc = b * b - 4 * a * c
        sq = np.sqrt(np.maximum(disc, 0.0))
        t = (-b - sq) / (2 * a)
        return np.where((disc > 0) & (t > 0), t, np.inf)


def render(prims, intr, W, H, T_wc, colour_k=(3.1, 4.7, 2.3)):
    """z-depth (metres, float64, inf = nothing hit) and procedural RGB for camera->world pose T_wc."""
    fx, fy, cx, cy = intr
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    dc = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs)], axis=-1)  # z component 1 -> t == z-depth
    R, t = T_wc[:3, :3], T_wc[:3, 3]
    d = dc @ R.T
    o = np.asarray(t, np.float64)
    with np.errstate(divide="ignore"):
        inv_d = [1.0 / d[..., a] for a in range(3)]
    z = np.full((H, W), np.inf)
    for p in prims:
        z = np.minimum(z, p.hit(o, d, inv_d))
    pw = o + d * np.where(np.isfinite(z), z, 0.0)[..., None]
    rgb = np.stack([128.0 + 127.0 * np.sin(k * (pw[..., 0] + 0.7 * pw[..., 1] + 1.3 * pw[..., 2]) + i)
                    for i, k in enumerate(colour_k)], axis=-1)
    rgba = np.empty((H, W, 4), np.uint8)
    rgba[..., :3] = np.clip(rgb, 0, 255).astype(np.uint8)
    rgba[..., 3] = 255
    return z, rgba


# ---------------------------------------------------------------------------------------------------
# depth wire formats of the reference's front-end
# ---------------------------------------------------------------------------------------------------
def depth_to_mm_rgbd(z, max_depth_m=None):
    """TUM / ICL-NUIM style: millimetres, rounded; 0 = invalid."""
    mm = np.floor(1000.0 * z + 0.5)
    bad = ~np.isfinite(z) | (mm > 32000) | (mm <= 0)
    if max_depth_m is not None:
        bad |= z > max_depth_m
    return np.where(bad, 0, mm).astype(np.int16)


def depth_to_mm_kitti(z, max_depth_m=40.0):
    """KITTI path of PrecomputedDepthProvider.cpp:35,52-57: PNG holds uint16 = z*256; values above
    max_depth*256 are dropped; int16 mm = (int16)((float)v * 1000/256)."""
    zz = np.where(np.isfinite(z), z, 0.0)
    v = np.floor(zz * 256.0).astype(np.int64)
    v = np.where(v > max_depth_m * 256, 0, v)
    v = np.clip(v, 0, 32767)  # the reference reads the PNG through int16
    mm = (v.astype(np.float32) * np.float32(1000.0 / 256.0)).astype(np.int64)
    mm = np.where(mm > 32767, 0, mm)  # the reference's int16 cast overflows here; such pixels are unusable
    return mm.astype(np.int16)


# ---------------------------------------------------------------------------------------------------
# poses
# ---------------------------------------------------------------------------------------------------
def look_rotation(yaw, pitch=0.0):
    """camera->world rotation; camera looks along +z, x right, y down; yaw about world y."""
    cyw, syw = np.cos(yaw), np.sin(yaw)
    cp, sp = np.cos(pitch), np.sin(pitch)
    Ry = np.array([[cyw, 0, syw], [0, 1, 0], [-syw, 0, cyw]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    return Ry @ Rx


def pose_matrix(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def world_to_camera(T_wc):
    """M_d = pose_d->GetM() (world -> camera), float32, as set by InfiniTamDriver::SetPoseLocalMap
    (InfiniTamDriver.h:173-178: SetInvM(T_map_w * T_w_c))."""
    return np.linalg.inv(T_wc).astype(np.float32)


# ---------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------
class Workload:
    """A named scene + camera + trajectory + ITMSceneParams.  frame(i) -> (rgba, depth_mm, M_d)."""

    def __init__(self, name, W, H, intr, prims, traj, scene_kwargs, kitti_depth=False, max_depth=None,
                 cull_z=None, stereo_noise_px=0.0, stereo_baseline_m=0.54, outlier_frac=0.0):
        self.name, self.W, self.H = name, W, H
        # stereo-matching noise: the depth a KITTI pipeline hands over comes from a disparity d = fx * b / z that is
        # wrong by a fraction of a pixel (sigma_z grows with z^2) and, for a few pixels, wrong altogether
        self.stereo_noise_px, self.stereo_baseline_m, self.outlier_frac = stereo_noise_px, stereo_baseline_m, outlier_frac
        self.cull_z = cull_z  # (behind, ahead) metres along world z: only nearby primitives are ray-cast
        self.intr = np.asarray(intr, np.float32)
        self.prims, self.traj, self.scene_kwargs = prims, traj, scene_kwargs
        self.kitti_depth, self.max_depth = kitti_depth, max_depth

    def pose(self, i):
        return self.traj(i)

    def frame(self, i):
        T_wc = self.traj(i)
        prims = self.prims
        if self.cull_z is not None:
            cz = T_wc[2, 3]
            prims = [p for p in prims if p.hi[2] >= cz - self.cull_z[0] and p.lo[2] <= cz + self.cull_z[1]]
        z, rgba = render(prims, self.intr.astype(np.float64), self.W, self.H, T_wc)
        if self.stereo_noise_px > 0.0 or self.outlier_frac > 0.0:
            rng = np.random.RandomState(1234 + 7919 * i)  # per frame, the same on every machine
            fb = float(self.intr[0]) * self.stereo_baseline_m
            with np.errstate(divide="ignore", invalid="ignore"):
                disp = fb / z + self.stereo_noise_px * rng.standard_normal(z.shape)
                if self.outlier_frac > 0.0:  # mismatches: a disparity from anywhere in the search range
                    bad = rng.random_sample(z.shape) < self.outlier_frac
                    disp = np.where(bad, rng.uniform(fb / 40.0, fb / 2.0, z.shape), disp)
                z = np.where(np.isfinite(z) & (disp > 0), fb / disp, np.inf)
        if self.kitti_depth:
            mm = depth_to_mm_kitti(z, self.max_depth or 40.0)
        else:
            mm = depth_to_mm_rgbd(z, self.max_depth)
        return rgba, mm, world_to_camera(T_wc)


def s_room(W=640, H=480, scale=1.0):
    """ICL-NUIM-like indoor scene (BASELINE configs 0 and 3 stand-in): 5x3x5 m box room, a sphere and a
    cube; camera on a 0.5 m circle yawing 2 deg/frame.  Upstream default scene params (5 mm voxels)."""
    intr = (481.2 * W / 640.0, 480.0 * H / 480.0, (W - 1) / 2.0, (H - 1) / 2.0)
    prims = [Box((-2.5, -1.5, -2.5), (2.5, 1.5, 2.5), inside=True), Sphere((0.8, 0.6, 1.2), 0.5),
             Box((-1.6, 0.5, 0.9), (-0.6, 1.5, 1.9))]

    def traj(i):
        a = np.deg2rad(2.0 * i)
        t = np.array([0.5 * np.cos(a), 0.1 * np.sin(0.5 * a), 0.5 * np.sin(a)])
        return pose_matrix(look_rotation(a * 0.5, 0.05 * np.sin(a)), t)

    return Workload("S-room", W, H, intr, prims, traj,
                    dict(voxel_size=0.005 * scale, mu=0.02 * scale, max_w=100, frustum_min=0.2, frustum_max=3.0))


def s_street(W=640, H=480, n_cars=20, loop_at=None, stereo_noise_px=0.0, outlier_frac=0.0):
    """KITTI-like street (BASELINE configs 1, 2, 4 stand-in): ground plane 1.65 m below the camera, two
    facades at x = +-8 m with 3 m-period relief, parked-car boxes; camera drives forward 1 m/frame with a
    slow yaw.  5 cm voxels, mu 0.2 m, frustum 0.5-40 m (the fork's own defaults are not knowable; these give
    ~8k blocks on the first frame like the reference's memory logs)."""
    if (W, H) == (640, 480):
        intr = (480.0, 480.0, 319.5, 239.5)
    else:
        intr = (707.09 * W / 1226.0, 707.09 * W / 1226.0, (W - 1) / 2.0, (H - 1) / 2.0)
    prims = [Box((-60.0, 1.65, -50.0), (60.0, 3.0, 5000.0))]  # ground slab (y is down)
    L = 4000.0
    for side in (-1.0, 1.0):
        x0 = 8.0 * side
        prims.append(Box((min(x0, x0 + 2 * side), -12.0, -50.0), (max(x0, x0 + 2 * side), 1.65, L)))
        for k in range(0, 700):  # relief boxes, 6 m period
            z0 = k * 6.0
            xa, xb = x0 - 0.6 * side, x0
            prims.append(Box((min(xa, xb), -6.0 + (k % 3), z0), (max(xa, xb), 1.65, z0 + 1.5)))
    rng = np.random.RandomState(7)
    for k in range(n_cars * 40):
        z0 = 12.0 + k * 9.0 + rng.uniform(0, 3)
        side = -1.0 if (k % 2) else 1.0
        xc = side * 5.2
        prims.append(Box((xc - 0.9, 0.15, z0), (xc + 0.9, 1.65, z0 + 4.2)))

    def traj(i):
        j = i if loop_at is None or i < loop_at else i - loop_at
        yaw = np.deg2rad(3.0) * np.sin(0.05 * j)
        return pose_matrix(look_rotation(yaw, 0.0), np.array([0.3 * np.sin(0.03 * j), 0.0, 1.0 * j]))

    return Workload("S-street", W, H, intr, prims, traj,
                    dict(voxel_size=0.05, mu=0.2, max_w=100, frustum_min=0.5, frustum_max=40.0), kitti_depth=True,
                    max_depth=40.0, cull_z=(5.0, 48.0), stereo_noise_px=stereo_noise_px, outlier_frac=outlier_frac)


def s_tiny(W=64, H=48):
    """64x48 fixture-sized room for CPU tests and committed golden vectors."""
    w = s_room(W, H, scale=4.0)  # 2 cm voxels keep the block count small
    w.name = "S-tiny"
    return w
