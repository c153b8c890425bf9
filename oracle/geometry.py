"""Restatement of the reference's measurement geometry.

TEST INFRASTRUCTURE -- PARITY UNPINNED for the cv2 pieces: `cv2.undistortPoints` and `cv2.Rodrigues`
(opencv-contrib-python==4.11.0.86, requirements.txt:2) are not installed here, so they are restated from OpenCV's published
algorithm (calib3d/undistort: 5 fixed-point iterations, TermCriteria(COUNT, 5, 0.01) when none is passed).  Inputs are pinned by
the reference's own data files (camera_calibration.json, extrinsics.json), copied as fixtures to tests/golden/.  Everything
else follows measurement.py line by line in float64 numpy.
"""
import numpy as np


def rodrigues(rvec):
    """cv2.Rodrigues(rvec) -> R (measurement.py:139)."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = np.linalg.norm(r)
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    c, s = np.cos(theta), np.sin(theta)
    k = r / theta
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]], dtype=np.float64)
    return c * np.eye(3) + (1 - c) * np.outer(k, k) + s * K


def compute_camera_plane(R, t):
    """measurement.py:44-48."""
    n_c = R[:, 2].astype(np.float64)
    d_c = -float(n_c.dot(t))
    return n_c, d_c


def undistort_point(u, v, K, dist, iters=5):
    """cv2.undistortPoints(pts, K, dist, P=None) for one point and the 5-coefficient model (k1,k2,p1,p2,k3)."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    k1, k2, p1, p2, k3 = (float(d) for d in np.asarray(dist).ravel()[:5])
    ifx, ify = 1.0 / fx, 1.0 / fy
    x = (u - cx) * ifx
    y = (v - cy) * ify
    x0, y0 = x, y
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2)
        if icdist < 0:
            x, y = (u - cx) * ifx, (v - cy) * ify
            break
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (x0 - dx) * icdist
        y = (y0 - dy) * icdist
    return x, y


def pixel_to_world_using_camera_plane(u, v, K, dist, R, t, n_c, d_c):
    """measurement.py:50-65."""
    x_n, y_n = undistort_point(float(u), float(v), K, dist)
    d_cam = np.array([x_n, y_n, 1.0], dtype=np.float64)
    denom = float(n_c.dot(d_cam))
    if abs(denom) < 1e-9:
        return None
    s = -d_c / denom
    X_cam = s * d_cam
    return R.T.dot(X_cam - t)


def kmeans_1d_two_clusters(values, max_iters=10):
    """measurement.py:88-113: Lloyd iterations for k = 2 on scalars, centres seeded with min / max.  The reference keeps the
    PREVIOUS assignment when it stops on unchanged centres or on an empty cluster, and compares centres with `==`."""
    v = np.asarray(values, dtype=np.float64)
    n = v.shape[0]
    if v.size < 2:
        m = float(v.mean())
        return np.zeros(n, dtype=int), (m, m)
    lo, hi = float(v.min()), float(v.max())
    assign = np.zeros(n, dtype=int)
    for _ in range(max_iters):
        nearer_hi = (np.abs(v - hi) < np.abs(v - lo)).astype(int)
        k = int(nearer_hi.sum())
        if k == 0 or k == n:
            break
        lo_new = float(v[nearer_hi == 0].mean())
        hi_new = float(v[nearer_hi == 1].mean())
        if lo_new == lo and hi_new == hi:
            break
        lo, hi, assign = lo_new, hi_new, nearer_hi
    return assign, (lo, hi)
