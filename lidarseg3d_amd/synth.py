"""Synthetic inputs for tests and bench.py (no dataset or checkpoint is available offline).

* ``lidar_frame`` — a spinning-LiDAR model: evenly spaced beam elevations,
  uniform azimuth, range = min(ground hit, random obstacle, r_max) with multiplicative noise.
* ``random_state_dict`` — deterministic weights for a given {name: shape} manifest, so a fixture only
  has to carry a seed.  Uses numpy's PCG64 stream, which is stable across numpy versions/platforms.
* ``camera_inputs`` — image_features / camera_semantic_embeddings / points_cuv in the format produced
  by the reference pipeline (det3d/datasets/pipelines/segpreprocess.py:649-671).
"""
import numpy as np

NUSC = dict(beams=32, fov=(-30.0, 10.0), cp=5, pc_range=[-51.2, -51.2, -5.0, 51.2, 51.2, 3.0],
            voxel_size=[0.1, 0.1, 0.2], num_class=17, sensor_h=1.84, r_max=70.0)
KITTI = dict(beams=64, fov=(-25.0, 3.0), cp=4, pc_range=[-75.2, -75.2, -4.0, 75.2, 75.2, 2.0],
             voxel_size=[0.1, 0.1, 0.15], num_class=20, sensor_h=1.73, r_max=80.0)
WAYMO = dict(beams=64, fov=(-17.6, 2.4), cp=5, pc_range=[-75.2, -75.2, -2.0, 75.2, 75.2, 4.0],
             voxel_size=[0.1, 0.1, 0.15], num_class=23, sensor_h=2.0, r_max=75.0)


def lidar_frame(n, seed=0, beams=32, fov=(-30.0, 10.0), cp=5, sensor_h=1.84, r_max=70.0, **_):
    """-> points [n, cp] f32: x, y, z, intensity[, ring] (z relative to the sensor, ground at -sensor_h)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ring = rng.integers(0, beams, size=n)
    elev = np.deg2rad(fov[0] + (fov[1] - fov[0]) * ring / max(beams - 1, 1))
    azim = rng.uniform(-np.pi, np.pi, size=n)
    with np.errstate(divide="ignore"):
        r_ground = np.where(elev < 0, sensor_h / np.maximum(-np.sin(elev), 1e-6), np.inf)
    r_obst = rng.gamma(2.0, 8.0, size=n) + 1.0
    r = np.minimum(np.minimum(r_ground, r_obst), r_max) * (1.0 + rng.normal(0.0, 0.002, size=n))
    x = r * np.cos(elev) * np.cos(azim)
    y = r * np.cos(elev) * np.sin(azim)
    z = r * np.sin(elev)
    cols = [x, y, z, rng.uniform(0.0, 255.0, size=n)]
    if cp >= 5:
        cols.append(ring.astype(np.float64))
    while len(cols) < cp:
        cols.append(rng.uniform(0.0, 1.0, size=n))
    return np.stack(cols[:cp], axis=1).astype(np.float32)


def random_state_dict(shapes, seed=0):
    """{name: shape} -> {name: np.float32 array} (int64 scalar for num_batches_tracked).
    Scales keep activations O(1) through deep stacks: conv/linear weights ~ N(0, 2/fan_in),
    BN/LN scale ~ U(0.5,1.5), shift ~ N(0,0.1), running_mean ~ N(0,0.1), running_var ~ U(0.5,1.5)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = np.zeros(shp, np.int64)
        elif leaf == "running_var":
            out[name] = rng.uniform(0.5, 1.5, size=shp).astype(np.float32)
        elif leaf == "running_mean":
            out[name] = rng.normal(0.0, 0.1, size=shp).astype(np.float32)
        elif len(shp) == 1 and leaf == "weight":
            out[name] = rng.uniform(0.5, 1.5, size=shp).astype(np.float32)
        elif len(shp) == 1:
            out[name] = rng.normal(0.0, 0.1, size=shp).astype(np.float32)
        elif len(shp) == 5:  # sparse conv (kD,kH,kW,Cin,Cout): ~1/3 of the taps are active on LiDAR data
            fan = shp[0] * shp[1] * shp[2] * shp[3] / 3.0
            out[name] = rng.normal(0.0, np.sqrt(2.0 / fan), size=shp).astype(np.float32)
        else:  # Linear (out,in) / Conv1d (out,in,1) / in_proj (3E,E)
            fan = int(np.prod(shp[1:]))
            out[name] = rng.normal(0.0, np.sqrt(2.0 / fan), size=shp).astype(np.float32)
    return out


def camera_inputs(n, seed=0, ncam=6, c_img=48, h=160, w=240, num_class=17, p_valid=0.75, batch=1):
    """-> image_features [B,ncam,c_img,h,w], camera_semantic_embeddings [B,c_img,num_class,1],
    points_cuv [n,4] = (valid, cam, h, w) with cam in {1..ncam} mapped to [-1,1] as
    segpreprocess.py:655 does ((cam-1)/(ncam-1)*2-1) and (h,w) ~ U(-1,1)."""
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    img = rng.normal(0.0, 1.0, size=(batch, ncam, c_img, h, w)).astype(np.float32)
    emb = rng.normal(0.0, 1.0, size=(batch, c_img, num_class, 1)).astype(np.float32)
    valid = (rng.uniform(size=n) < p_valid).astype(np.float32)
    cam = rng.integers(1, ncam + 1, size=n).astype(np.float32)
    cam_n = ((cam - 1.0) / np.float32(max(ncam - 1, 1)) * 2.0 - 1.0).astype(np.float32)
    hw = rng.uniform(-1.0, 1.0, size=(n, 2)).astype(np.float32)
    cuv = np.stack([valid, cam_n, hw[:, 0], hw[:, 1]], axis=1).astype(np.float32)
    cuv[valid == 0, 1:] = 0.0
    return img, emb, cuv


def camera_rig(ncam=6, seed=0, im_shape=(900, 1600)):
    """nuScenes-like ring of pinhole cameras around the ego vehicle: (ref_to_global [4,4], cams_from_global [ncam,4,4],
    intrinsics [ncam,3,3]) float64"""
    rng = np.random.default_rng(seed)

    def rot(axis, a):
        c, s_ = np.cos(a), np.sin(a)
        m = np.eye(3)
        i, j = [(1, 2), (0, 2), (0, 1)][axis]
        m[i, i], m[i, j], m[j, i], m[j, j] = c, -s_, s_, c
        return m
    yaw0 = rng.uniform(-np.pi, np.pi)
    ref_to_global = np.eye(4)
    ref_to_global[:3, :3] = rot(2, yaw0)
    ref_to_global[:3, 3] = rng.uniform(-500, 500, 3) * np.array([1, 1, 0.01])
    global_from_ref = ref_to_global
    cams, Ks = [], []
    for c in range(ncam):
        yaw = 2 * np.pi * c / ncam + rng.normal() * 0.02
        # camera frame: z forward, x right, y down; the camera looks along (cos yaw, sin yaw, 0) in the ego frame
        fwd = np.array([np.cos(yaw), np.sin(yaw), 0.0])
        right = np.array([np.sin(yaw), -np.cos(yaw), 0.0])
        down = np.array([0.0, 0.0, -1.0])
        R_ego_from_cam = np.stack([right, down, fwd], axis=1)
        t = np.array([1.5 * np.cos(yaw), 1.5 * np.sin(yaw), 1.6])
        ego_from_cam = np.eye(4)
        ego_from_cam[:3, :3], ego_from_cam[:3, 3] = R_ego_from_cam, t
        cam_from_global = np.linalg.inv(global_from_ref.dot(ego_from_cam))
        cams.append(cam_from_global)
        f = 1266.0 + rng.normal() * 5
        Ks.append(np.array([[f, 0.0, im_shape[1] / 2 + rng.normal() * 5], [0.0, f, im_shape[0] / 2 + rng.normal() * 5], [0.0, 0.0, 1.0]]))
    return ref_to_global, np.stack(cams), np.stack(Ks)
