"""CPU checks of the geometry restatement (oracle/geometry.py) on the reference's own calibration
data (tests/golden/camera_calibration.json, extrinsics.json: copies of the reference's data files)."""
import json
import os

import numpy as np

from oracle import geometry as og

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_calib():
    c = json.load(open(os.path.join(G, "camera_calibration.json")))
    e = json.load(open(os.path.join(G, "extrinsics.json")))
    K = np.array(c["camera_matrix"], dtype=np.float64)
    dist = np.array(c["dist_coeffs"], dtype=np.float64).ravel()
    R = og.rodrigues(np.array(e["rvec"], dtype=np.float64))
    t = np.array(e["tvec"], dtype=np.float64)
    return K, dist, R, t


def _distort(x, y, dist):
    k1, k2, p1, p2, k3 = dist
    r2 = x * x + y * y
    rad = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    return x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x), y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y


def test_rodrigues_is_a_rotation_about_rvec():
    K, dist, R, t = load_calib()
    e = json.load(open(os.path.join(G, "extrinsics.json")))
    r = np.array(e["rvec"])
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(R) - 1) < 1e-14
    assert np.allclose(R @ r, r, atol=1e-14)                                    # the axis is fixed
    assert abs(np.arccos((np.trace(R) - 1) / 2) - np.linalg.norm(r)) < 1e-12    # by the angle |rvec|


def test_undistort_inverts_the_distortion_model_near_the_centre():
    K, dist, R, t = load_calib()
    for u, v in [(636.0, 422.0), (700.0, 500.0), (500.0, 300.0), (900.0, 600.0)]:
        x, y = og.undistort_point(u, v, K, dist)
        xd, yd = _distort(x, y, dist)
        # 5 fixed-point steps: converged to far below a pixel for points of the reference's ROI
        assert abs(xd * K[0, 0] + K[0, 2] - u) < 2e-3 and abs(yd * K[1, 1] + K[1, 2] - v) < 2e-3


def test_world_points_lie_on_the_fabric_plane_and_reproject():
    K, dist, R, t = load_calib()
    n_c, d_c = og.compute_camera_plane(R, t)
    for u, v in [(640.0, 480.0), (100.0, 700.0), (1200.0, 350.0)]:
        X = og.pixel_to_world_using_camera_plane(u, v, K, dist, R, t, n_c, d_c)
        assert abs(X[2]) < 1e-12                                                # world z = 0: the board plane
        Xc = R @ X + t
        x, y = og.undistort_point(u, v, K, dist)
        assert np.allclose(Xc[:2] / Xc[2], [x, y], atol=1e-12)


def test_kmeans_rows():
    vals = np.array([400.0, 402.5, 399.0, 520.0, 523.0, 518.5, 521.0])
    labels, (c0, c1) = og.kmeans_1d_two_clusters(vals)
    assert labels.tolist() == [0, 0, 0, 1, 1, 1, 1] and abs(c0 - vals[:3].mean()) < 1e-12 and abs(c1 - vals[3:].mean()) < 1e-12
    labels, (c0, c1) = og.kmeans_1d_two_clusters(np.array([5.0, 5.0, 5.0]))      # one cluster empties: previous labels kept
    assert labels.tolist() == [0, 0, 0] and (c0, c1) == (5.0, 5.0)
    labels, c = og.kmeans_1d_two_clusters(np.array([7.0]))
    assert labels.tolist() == [0] and c == (7.0, 7.0)
