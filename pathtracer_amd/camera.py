"""Camera matrices for the CameraBuffer the shaders read (PathTracer/src/RayTracingLayer.h:17-24,
RayTracingLayer.cpp:403-409; RayGen.glsl:22-29,329-340).

The reference's `Camera` class lives in its absent engine submodule, so its projection convention
is unknown.  This module fixes one and every config stores the resulting matrices explicitly:
right-handed view space looking down -Z, perspective projection with depth 0..1 and the Y axis
flipped so that image row 0 (NDC y = -1 in RayGen.glsl:329-330) is the top of the picture.
Matrices are column-major 4x4 float32, as glm stores them.
"""
import math

import numpy as np


def perspective(fov_y_deg, aspect, near=0.1, far=1000.0):
    f = 1.0 / math.tan(math.radians(fov_y_deg) * 0.5)
    m = np.zeros((4, 4), np.float64)      # m[row, col]
    m[0, 0] = f / aspect
    m[1, 1] = -f
    m[2, 2] = far / (near - far)
    m[2, 3] = (near * far) / (near - far)
    m[3, 2] = -1.0
    return m


def look_at(eye, target, up=(0.0, 1.0, 0.0)):
    eye = np.asarray(eye, np.float64)
    f = np.asarray(target, np.float64) - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, np.asarray(up, np.float64))
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[0, 3], m[1, 3], m[2, 3] = -s.dot(eye), -u.dot(eye), f.dot(eye)
    return m


def from_pitch_yaw(position, pitch, yaw):
    """View matrix from the reference's CameraSpecification{pitch, yaw} (RayTracingLayer.cpp:38-42):
    yaw about +Y measured from +Z towards +X, positive pitch looks down (the engine's own
    convention is unknown; with this one the commented Sponza pose looks down the nave)."""
    cp, sp, cy, sy = math.cos(pitch), math.sin(pitch), math.cos(yaw), math.sin(yaw)
    forward = np.array([sy * cp, -sp, cy * cp])
    return look_at(position, np.asarray(position, np.float64) + forward)


def camera_buffer(view, proj):
    """The five matrices of CameraBuffer, each column-major, as one float32[80]."""
    view = np.asarray(view, np.float64)
    proj = np.asarray(proj, np.float64)
    vp = proj @ view
    mats = [vp, np.linalg.inv(vp), view, np.linalg.inv(view), np.linalg.inv(proj)]
    return np.concatenate([m.T.reshape(-1) for m in mats]).astype(np.float32)   # .T: row-major -> column-major


def cornell_camera(width, height):
    """Fixed view into the open front (+z side) of CornellBox.gltf scaled by 0.1 (RayTracingLayer.cpp:32-34)."""
    view = look_at((-0.025, 0.26, 0.75), (-0.025, 0.26, -0.3))
    return camera_buffer(view, perspective(45.0, width / height, 0.01, 100.0))


def suzanne_camera(width, height):
    """Three-quarter view of Suzanne.gltf (the head spans x in [-1.34, 1.34], y in [-0.97, 0.95], z in [-0.80, 0.83])."""
    view = look_at((0.9, 0.4, 2.5), (0.0, -0.02, 0.0))
    return camera_buffer(view, perspective(45.0, width / height, 0.01, 100.0))
