"""Host-side WGS-84 inverse geodesic (vectorised numpy, fp64).

Used for synthetic-network geometry and for the per-call compatibility API, where the
caller hands `candidate_dis` to `GeneratorV4.predict` exactly as search.py:352-357 does.
The batched engine never calls this: it evaluates the same Vincenty iteration on the
device (csrc/tsg_device.h, `tsg_dev_geodesic_m`).
"""
import numpy as np

WGS84_A = 6378137.0
WGS84_F = 1.0 / 298.257223563
WGS84_B = WGS84_A * (1.0 - WGS84_F)


def geodesic_m(lat1, lon1, lat2, lon2, max_iter=20, eps=1e-12):
    """Vincenty inverse distance in metres; inputs in degrees, broadcastable arrays."""
    lat1, lon1, lat2, lon2 = np.broadcast_arrays(*[np.asarray(v, dtype=np.float64)
                                                   for v in (lat1, lon1, lat2, lon2)])
    f, a, b = WGS84_F, WGS84_A, WGS84_B
    rad = np.pi / 180.0
    U1 = np.arctan((1.0 - f) * np.tan(lat1 * rad))
    U2 = np.arctan((1.0 - f) * np.tan(lat2 * rad))
    L = (lon2 - lon1) * rad
    sinU1, cosU1, sinU2, cosU2 = np.sin(U1), np.cos(U1), np.sin(U2), np.cos(U2)
    lam = L.copy()
    active = np.ones(lam.shape, dtype=bool)
    sin_sigma = np.zeros_like(lam)
    cos_sigma = np.ones_like(lam)
    sigma = np.zeros_like(lam)
    cos_sq_alpha = np.ones_like(lam)
    cos2sm = np.zeros_like(lam)
    with np.errstate(divide='ignore', invalid='ignore'):
        for _ in range(max_iter):
            sin_lam, cos_lam = np.sin(lam), np.cos(lam)
            t1 = cosU2 * sin_lam
            t2 = cosU1 * sinU2 - sinU1 * cosU2 * cos_lam
            ss = np.sqrt(t1 * t1 + t2 * t2)
            cs = sinU1 * sinU2 + cosU1 * cosU2 * cos_lam
            sg = np.arctan2(ss, cs)
            sin_alpha = np.where(ss != 0.0, cosU1 * cosU2 * sin_lam / ss, 0.0)
            csa = 1.0 - sin_alpha * sin_alpha
            c2 = np.where(csa != 0.0, cs - 2.0 * sinU1 * sinU2 / csa, 0.0)
            C = f / 16.0 * csa * (4.0 + f * (4.0 - 3.0 * csa))
            lam_new = L + (1.0 - C) * f * sin_alpha * (
                sg + C * ss * (c2 + C * cs * (-1.0 + 2.0 * c2 * c2)))
            upd = active
            sin_sigma = np.where(upd, ss, sin_sigma)
            cos_sigma = np.where(upd, cs, cos_sigma)
            sigma = np.where(upd, sg, sigma)
            cos_sq_alpha = np.where(upd, csa, cos_sq_alpha)
            cos2sm = np.where(upd, c2, cos2sm)
            delta = np.abs(lam_new - lam)
            lam = np.where(upd, lam_new, lam)
            active = active & ~(delta < eps) & (ss != 0.0)
            if not active.any():
                break
    u_sq = cos_sq_alpha * (a * a - b * b) / (b * b)
    A = 1.0 + u_sq / 16384.0 * (4096.0 + u_sq * (-768.0 + u_sq * (320.0 - 175.0 * u_sq)))
    B = u_sq / 1024.0 * (256.0 + u_sq * (-128.0 + u_sq * (74.0 - 47.0 * u_sq)))
    d_sigma = B * sin_sigma * (cos2sm + B / 4.0 * (
        cos_sigma * (-1.0 + 2.0 * cos2sm * cos2sm)
        - B / 6.0 * cos2sm * (-3.0 + 4.0 * sin_sigma * sin_sigma) * (-3.0 + 4.0 * cos2sm * cos2sm)))
    out = b * A * (sigma - d_sigma)
    return np.where(sin_sigma == 0.0, 0.0, out)


def candidate_dis(des_gps, cand_gps_list):
    """search.py:352-357: (lon, lat) tuples in, kilometres * 10 out (fp64 numpy array)."""
    cand = np.asarray(cand_gps_list, dtype=np.float64).reshape(-1, 2)
    return geodesic_m(des_gps[1], des_gps[0], cand[:, 1], cand[:, 0]) / 1000.0 * 10
