"""WGS-84 inverse geodesic (Vincenty 1975), fp64.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Stands in for ``geopy.distance.distance(p, q).kilometers`` at the reference call
sites search.py:183,355,543,670,768.  geopy (and its geographiclib backend) is a
third-party dependency that is absent from /root/reference and from this image and
whose version the reference never pins => parity at this boundary is UNPINNED.
What is restated here is the published Vincenty inverse formula on the WGS-84
ellipsoid (a = 6378137 m, f = 1/298.257223563), which agrees with Karney's
algorithm (what geopy calls) to well under a millimetre for the non-antipodal,
city-scale distances this path sees.  The HIP kernel ``tsg_dev_geodesic_m``
(ts-trajgen_amd/csrc/tsg_device.h) implements the same iteration in fp64.
"""
import math

WGS84_A = 6378137.0
WGS84_F = 1.0 / 298.257223563
WGS84_B = WGS84_A * (1.0 - WGS84_F)
MAX_ITER = 20
EPS = 1e-12


def geodesic_m(lat1, lon1, lat2, lon2):
    """Distance in metres between (lat1, lon1) and (lat2, lon2), degrees in, fp64."""
    f = WGS84_F
    a = WGS84_A
    b = WGS84_B
    if lat1 == lat2 and lon1 == lon2:
        return 0.0
    rad = math.pi / 180.0
    U1 = math.atan((1.0 - f) * math.tan(lat1 * rad))
    U2 = math.atan((1.0 - f) * math.tan(lat2 * rad))
    L = (lon2 - lon1) * rad
    sinU1, cosU1 = math.sin(U1), math.cos(U1)
    sinU2, cosU2 = math.sin(U2), math.cos(U2)
    lam = L
    sin_sigma = cos_sigma = sigma = cos_sq_alpha = cos2sm = 0.0
    for _ in range(MAX_ITER):
        sin_lam, cos_lam = math.sin(lam), math.cos(lam)
        t1 = cosU2 * sin_lam
        t2 = cosU1 * sinU2 - sinU1 * cosU2 * cos_lam
        sin_sigma = math.sqrt(t1 * t1 + t2 * t2)
        if sin_sigma == 0.0:
            return 0.0
        cos_sigma = sinU1 * sinU2 + cosU1 * cosU2 * cos_lam
        sigma = math.atan2(sin_sigma, cos_sigma)
        sin_alpha = cosU1 * cosU2 * sin_lam / sin_sigma
        cos_sq_alpha = 1.0 - sin_alpha * sin_alpha
        if cos_sq_alpha != 0.0:
            cos2sm = cos_sigma - 2.0 * sinU1 * sinU2 / cos_sq_alpha
        else:
            cos2sm = 0.0
        C = f / 16.0 * cos_sq_alpha * (4.0 + f * (4.0 - 3.0 * cos_sq_alpha))
        lam_new = L + (1.0 - C) * f * sin_alpha * (
            sigma + C * sin_sigma * (cos2sm + C * cos_sigma * (-1.0 + 2.0 * cos2sm * cos2sm)))
        delta = abs(lam_new - lam)
        lam = lam_new
        if delta < EPS:
            break
    u_sq = cos_sq_alpha * (a * a - b * b) / (b * b)
    A = 1.0 + u_sq / 16384.0 * (4096.0 + u_sq * (-768.0 + u_sq * (320.0 - 175.0 * u_sq)))
    B = u_sq / 1024.0 * (256.0 + u_sq * (-128.0 + u_sq * (74.0 - 47.0 * u_sq)))
    d_sigma = B * sin_sigma * (cos2sm + B / 4.0 * (
        cos_sigma * (-1.0 + 2.0 * cos2sm * cos2sm)
        - B / 6.0 * cos2sm * (-3.0 + 4.0 * sin_sigma * sin_sigma) * (-3.0 + 4.0 * cos2sm * cos2sm)))
    return b * A * (sigma - d_sigma)


def candidate_dis(des_gps, cand_gps):
    """search.py:355-356: gps tuples are (lon, lat); result is kilometres * 10 (units of 100 m)."""
    return geodesic_m(des_gps[1], des_gps[0], cand_gps[1], cand_gps[0]) / 1000.0 * 10


class _Distance(object):
    """Duck-type of ``geopy.distance.distance`` for the golden generator's import stub."""

    def __init__(self, p, q):
        self.meters = geodesic_m(p[0], p[1], q[0], q[1])

    @property
    def kilometers(self):
        return self.meters / 1000.0

    @property
    def km(self):
        return self.meters / 1000.0
