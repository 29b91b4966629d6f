"""Caller-side contract of the hot path: the per-frame uniform fill.

Reproduces Renderer::Render (/root/reference/src/rendering_backend/renderer.cpp:187-205): the previous
frame's view/projection are carried from the last call (all-zero matrices on frame 0, because the
reference keeps them in a zero-initialised function static), `frame_index` post-increments from 0, and
`camera_view_inverse` is the camera transform itself.  Projection is the reverse-Z infinite projection
of vulkan_utils.h:494-503; camera transform = T * R with R from yaw/pitch/roll
(scene_loader.cpp:60-69).
"""
import numpy as np

from . import abi


def infinite_reverse_depth_projection(yfov, aspect, znear):
    """vulkan_utils.h:494-503 as a math matrix (row, col): z_ndc = znear / -z_view, w = -z_view."""
    scale = 1.0 / np.tan(yfov * 0.5)
    m = np.zeros((4, 4))
    m[0, 0] = scale / aspect
    m[1, 1] = scale
    m[3, 2] = -1.0
    m[2, 3] = znear
    return m


def yaw_pitch_roll(yaw, pitch, roll):
    """glm::yawPitchRoll: R = Ry(yaw) * Rx(pitch) * Rz(roll)."""
    cy, sy = np.cos(yaw), np.sin(yaw)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cr, sr = np.cos(roll), np.sin(roll)
    ry = np.array([[cy, 0, sy, 0], [0, 1, 0, 0], [-sy, 0, cy, 0], [0, 0, 0, 1.0]])
    rx = np.array([[1, 0, 0, 0], [0, cp, -sp, 0], [0, sp, cp, 0], [0, 0, 0, 1.0]])
    rz = np.array([[cr, -sr, 0, 0], [sr, cr, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    return ry @ rx @ rz


def light_projview(direction):
    """scene_loader.cpp:85-94: glm::ortho(-8, 8, -8, 8, 12, 0.1) * glm::lookAt(-direction * 12, origin, +y) under the
    reference's GLM_FORCE_DEPTH_ZERO_TO_ONE (pch.h:37), right-handed: the shadow map's reverse-Z orthographic frustum
    (depth 1 at 0.1 m from the light's eye point, 0 at 12 m).  Only the rasterised shadow map and composition.frag's PCF
    read it -- the ray-traced path never does -- but the scene host fills it like the reference."""
    d = np.asarray(direction, dtype=np.float64)
    d = d / np.linalg.norm(d)
    left, right, bottom, top, z_near, z_far = -8.0, 8.0, -8.0, 8.0, 12.0, 0.1
    ortho = np.eye(4)
    ortho[0, 0] = 2.0 / (right - left)
    ortho[1, 1] = 2.0 / (top - bottom)
    ortho[2, 2] = -1.0 / (z_far - z_near)
    ortho[0, 3] = -(right + left) / (right - left)
    ortho[1, 3] = -(top + bottom) / (top - bottom)
    ortho[2, 3] = -z_near / (z_far - z_near)
    eye = -d * 12.0
    f = -eye / np.linalg.norm(eye)                       # normalize(center - eye), center = origin
    s_ = np.cross(f, [0.0, 1.0, 0.0])
    s_ = s_ / np.linalg.norm(s_)                         # a light straight down the y axis is degenerate here as in glm::lookAt
    u = np.cross(s_, f)
    view = np.eye(4)
    view[0, :3], view[1, :3], view[2, :3] = s_, u, -f
    view[0, 3], view[1, 3], view[2, 3] = -np.dot(s_, eye), -np.dot(u, eye), np.dot(f, eye)
    return ortho @ view


def directional_light(direction, color=(1.0, 1.0, 1.0), intensity=30.0):
    """scene_loader.cpp:73-99: projview (light_projview), direction vec4(w=0), colour vec4(w=1), intensity vec4(30)."""
    d = np.asarray(direction, dtype=np.float64)
    d = d / np.linalg.norm(d)
    light = np.zeros((), abi.directional_light_dtype)
    with np.errstate(invalid="ignore", divide="ignore"):
        pv = light_projview(d)
    light["projview"] = abi.mat_to_glm(pv if np.isfinite(pv).all() else np.eye(4))
    light["direction"] = [d[0], d[1], d[2], 0.0]
    light["color"] = [color[0], color[1], color[2], 1.0]
    light["intensity"] = [intensity] * 4
    return light


class FrameDriver:
    """Stateful per-frame fill, one call per rendered frame (renderer.cpp:184-205)."""

    def __init__(self, width, height, yfov, znear, light, aspect=None):
        self.width, self.height = int(width), int(height)
        # the aspect ratio comes from the glTF camera, not the window (scene_loader.cpp:47-51)
        self.aspect = (width / height) if aspect is None else aspect
        self.proj = infinite_reverse_depth_projection(yfov, self.aspect, znear)
        self.light = light
        self.frame_index = 0
        self._prev_view = np.zeros((4, 4))   # zero matrices on frame 0 (function-static zero init)
        self._prev_proj = np.zeros((4, 4))

    def next(self, position, yaw=0.0, pitch=0.0, roll=0.0, world=None):
        t = np.eye(4)
        t[:3, 3] = position
        transform = t @ yaw_pitch_roll(yaw, pitch, roll)
        if world is not None:                 # a scene that was rotated as a whole (scenes.rotated): the camera rides along
            transform = np.asarray(world, dtype=np.float64) @ transform
        view = np.linalg.inv(transform)
        pfd = np.zeros((), abi.per_frame_dtype)
        pfd["camera_view"] = abi.mat_to_glm(view)
        pfd["camera_proj"] = abi.mat_to_glm(self.proj)
        pfd["camera_view_inverse"] = abi.mat_to_glm(transform)
        pfd["camera_proj_inverse"] = abi.mat_to_glm(np.linalg.inv(self.proj))
        pfd["camera_viewproj_inverse"] = abi.mat_to_glm(np.linalg.inv(self.proj @ view))
        pfd["camera_view_prev_frame"] = abi.mat_to_glm(self._prev_view)
        pfd["camera_proj_prev_frame"] = abi.mat_to_glm(self._prev_proj)
        pfd["directional_light"] = self.light
        pfd["display_size"] = [self.width, self.height]
        pfd["display_size_inverse"] = [np.float32(1.0) / np.float32(self.width),
                                       np.float32(1.0) / np.float32(self.height)]
        pfd["frame_index"] = self.frame_index
        pfd["blue_noise_texture_index"] = 0
        self.frame_index += 1
        self._prev_view, self._prev_proj = view, self.proj
        return pfd


def dolly_frames(scene, width, height, n_frames, start_frame_index=0):
    """The fixed camera path of SURVEY.md section 8(d): `step` metres per frame along the view axis."""
    drv = FrameDriver(width, height, scene.camera["yfov"], scene.camera["znear"], scene.light,
                      aspect=scene.camera.get("aspect"))
    drv.frame_index = start_frame_index
    pos = np.asarray(scene.camera["position"], dtype=np.float64)
    step = np.asarray(scene.camera["dolly"], dtype=np.float64)
    out = []
    for i in range(n_frames):
        out.append(drv.next(pos + step * i, scene.camera["yaw"], scene.camera["pitch"], scene.camera.get("roll", 0.0), scene.camera.get("world")))
    return out
