"""Host-side camera conventions of the reference (data contract of GenerateCameraRay).

Mirrors Source/Camera.h:80-92 (reversed-Z projection), Source/CameraController.h:42-49 (orbit),
:160-166 (free fly) and the glm closed forms in SURVEY.md section 11.  Matrices are glm layout:
column-major, returned as float32 arrays of 16 with m[col*4+row].
"""
import math

import numpy as np


def _cm(m4):
    """4x4 ndarray (row, col indexing) -> glm column-major flat float32[16]."""
    return np.ascontiguousarray(np.asarray(m4, dtype=np.float64).T.reshape(16).astype(np.float32))


def from_cm(flat):
    return np.asarray(flat, dtype=np.float64).reshape(4, 4).T


def perspective_rh_zo(fovy, aspect, z_near, z_far):
    """glm::perspectiveRH_ZO."""
    t = math.tan(fovy / 2.0)
    m = np.zeros((4, 4))
    m[0, 0] = 1.0 / (aspect * t)
    m[1, 1] = 1.0 / t
    m[2, 2] = z_far / (z_near - z_far)
    m[3, 2] = -1.0
    m[2, 3] = -(z_far * z_near) / (z_far - z_near)
    return m


def view_to_clip(aspect, y_fov=math.pi / 2, z_near=0.01, z_far=100.0):
    """Camera::GetViewToClip (Camera.h:80-92): perspectiveRH_ZO called with near/far SWAPPED."""
    if z_far != 0.0:
        return perspective_rh_zo(y_fov, aspect, z_far, z_near)
    return perspective_rh_zo(y_fov, aspect, 100000.0, z_near)


def ortho_rh_zo(left, right, bottom, top, z_near, z_far):
    """glm::orthoRH_ZO."""
    m = np.eye(4)
    m[0, 0] = 2.0 / (right - left)
    m[1, 1] = 2.0 / (top - bottom)
    m[2, 2] = -1.0 / (z_far - z_near)
    m[0, 3] = -(right + left) / (right - left)
    m[1, 3] = -(top + bottom) / (top - bottom)
    m[2, 3] = -z_near / (z_far - z_near)
    return m


def ortho_view_to_clip(x_mag, y_mag, z_near=0.01, z_far=100.0):
    """Camera::GetViewToClip, orthographic branch (Camera.h:91): orthoRH_ZO(-1/x_mag, 1/x_mag, -1/y_mag, 1/y_mag, z_far, z_near)
    -- half extents 1/mag (sic) and near/far swapped for reversed Z."""
    return ortho_rh_zo(-1.0 / x_mag, 1.0 / x_mag, -1.0 / y_mag, 1.0 / y_mag, z_far, z_near)


def translate(v):
    m = np.eye(4)
    m[:3, 3] = v
    return m


def euler_angle_x(a):
    c, s = math.cos(a), math.sin(a)
    m = np.eye(4)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    return m


def euler_angle_z(a):
    c, s = math.cos(a), math.sin(a)
    m = np.eye(4)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


def euler_angle_xz(ax, az):
    """glm::eulerAngleXZ = Rx(ax) * Rz(az)."""
    return euler_angle_x(ax) @ euler_angle_z(az)


# world (x, y, z) -> view (x, z, -y): glm::mat3(vec3(1,0,0), vec3(0,0,-1), vec3(0,1,0)) has those COLUMNS
_M0 = np.eye(4)
_M0[:3, :3] = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float64).T


def orbit_world_to_view(centre=(0, 0, 0), radius=1.0, azimuth=0.0, inclination=0.0):
    """OrbitController::GetTransform (CameraController.h:42-49).  Defaults = the app's g_orbit."""
    return _M0 @ translate((0.0, radius, 0.0)) @ euler_angle_xz(-inclination, -azimuth) @ translate(-np.asarray(centre, dtype=np.float64))


def free_world_to_view(position=(0, -1, 0), yaw=0.0, pitch=0.0):
    """FreeController::GetTransform (CameraController.h:160-166)."""
    return _M0 @ euler_angle_xz(-pitch, -yaw) @ translate(-np.asarray(position, dtype=np.float64))


def cm(m4):
    return _cm(m4)


def inverse_transpose(m4):
    return np.linalg.inv(np.asarray(m4, dtype=np.float64)).T


def trs(translation=(0, 0, 0), rotation_xyzw=(0, 0, 0, 1), scale=(1, 1, 1)):
    """node matrix T * R * S (Gltf.cpp:1033-1035)."""
    x, y, z, w = rotation_xyzw
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 0],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w), 0],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y), 0],
                  [0, 0, 0, 1]], dtype=np.float64)
    s = np.diag([scale[0], scale[1], scale[2], 1.0])
    return translate(translation) @ r @ s


# Y-up -> Z-up root transform applied to every scene root (Gltf.cpp:1017-1022; glm column-major
# literal, so the COLUMNS are (1,0,0,0), (0,0,1,0), (0,-1,0,0), (0,0,0,1)).
Y_UP_TO_Z_UP = np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
