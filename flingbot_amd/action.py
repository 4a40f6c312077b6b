"""Action selection on the device.

`select_action` is the drop-in for SimEnv.get_max_value_valid_action (environment/simEnv.py:560-661): the value maps
stay on the GPU, `fs_select_action` (csrc/fs_action.hip) validates every candidate in parallel and returns the entry the
reference's descending walk would stop at; only that one candidate is then evaluated on the host in float64 (the geometry of
environment/utils.py:134-276, formulated here from the math and pinned to the reference's bits by
tests/golden/envutils_golden.npz) to build the returned `action_params` (p1, p2, pretransform pixels).
There is no host search: without the HIP library this module raises.
"""
import ctypes as C

import threading

import numpy as np
import torch

from .sim import load_library

KINDS = {"fling": 0, "stretchdrag": 1, "drag": 2, "place": 3}


# ---------------------------------------------------------------------------------------------------------------------------
# Host geometry of the action space (what environment/utils.py:134-276 and simEnv.py:517-537 compute), written from the math.
# Convention: image points are ROW vectors (first axis, second axis, 1) and a map M acts as p' = p @ M.  The results have to
# equal the reference's float64 bits (tests/golden/envutils_golden.npz), which fixes three things and nothing else:
#   * the ORDER of the homogeneous 3x3 products in get_transform_matrix -- a closed-form composition of the same affine map
#     differs in the last bit for a third of random inputs (numpy hands 3x3 products to BLAS, which fuses multiply-adds);
#   * np.linalg.inv for the camera pose;
#   * one 4x4-by-column product per back-projected point -- the batched (4x4)@(4xK) product takes another BLAS path and
#     rounds differently for almost every generic pose.
# ---------------------------------------------------------------------------------------------------------------------------
CAMERA_FOV_DEG = 39.5978


def _shift(offset):
    m = np.eye(3)
    m[2, :2] = offset
    return m


def _linear(block):
    m = np.eye(3)
    m[:2, :2] = block
    return m


def _about_centre(block, centre):
    """p -> (p - centre) @ block + centre"""
    return _shift(-centre) @ _linear(block) @ _shift(centre)


def get_transform_matrix(original_dim, resized_dim, rotation, scale):
    """Network-input pixel -> pre-transform image pixel for one (rotation [deg], scale) entry of the action space: zoom
    about the network image's centre pixel, turn about it, then stretch to the pre-transform resolution
    (environment/utils.py:161-176)."""
    centre = float(resized_dim // 2)
    theta = np.pi * rotation / 180
    c, s = np.cos(theta), np.sin(theta)
    zoom = _about_centre(scale * np.eye(2), centre)
    turn = _about_centre(np.array([[c, -s], [s, c]]), centre)
    return zoom @ turn @ _linear((original_dim / resized_dim) * np.eye(2))


def compute_pose(pos, lookat, up=(0, 0, 1)):
    """Camera-to-world matrix of a camera at `pos` looking at `lookat` (environment/utils.py:179-201): the inverse of the
    look-at view matrix with the camera's y and z axes flipped (image y grows downwards, depth grows along the view)."""
    pos, lookat, up = (np.asarray(v, dtype=np.float64) for v in (pos, lookat, up))
    forward = lookat - pos
    forward = forward / np.linalg.norm(forward)
    side = np.cross(forward, up / np.linalg.norm(up))
    side = side / np.linalg.norm(side)
    upward = np.cross(side, forward)
    view = np.eye(4)
    view[:3, :3] = (side, upward, -forward)
    view[:3, 3] = (-np.dot(side, pos), -np.dot(upward, pos), np.dot(forward, pos))
    pose = np.linalg.inv(view)
    pose[:, 1:3] *= -1
    return pose


def compute_intrinsics(fov, image_size):
    """Pinhole intrinsics of a square image with a `fov`-degree field of view (environment/utils.py:204-210)."""
    half = float(image_size) / 2
    k = np.diag([half / np.tan((np.pi * fov / 180) / 2)] * 2 + [1.0])
    k[:2, 2] = half
    return k


def back_project(depth_im, pixels, pose_matrix, fov=CAMERA_FOV_DEG, depth_scale=1):
    """World points [K, 3] under the image points pixels[K] = (column, row) of a depth image: pinhole rays scaled by the
    depth there, moved to the world by `pose_matrix`, x mirrored (the simulator's x axis points the other way)."""
    pixels = np.asarray(pixels).reshape(-1, 2)
    half = float(depth_im.shape[0]) / 2
    focal = half / np.tan((np.pi * fov / 180) / 2)
    depth = depth_im[pixels[:, 1], pixels[:, 0]] * depth_scale  # stays float32, like the scalar the reference scales
    if not depth.all():
        raise ValueError("zero depth under a pick pixel")
    rays = np.ones((len(pixels), 4))
    rays[:, :2] = (pixels - half) * depth[:, None] / focal
    rays[:, 2] = depth
    world = np.array([np.dot(pose_matrix, ray.reshape(4, 1))[:3, 0] for ray in rays])
    world[:, 0] *= -1
    return world


def pixel_to_3d(depth_im, x, y, pose_matrix, fov=CAMERA_FOV_DEG, depth_scale=1):
    """environment/utils.py:213-234 for one pixel."""
    return back_project(depth_im, [(x, y)], pose_matrix, fov, depth_scale)[0]


def pixels_to_3d_positions(pixels, scale, rotation, pretransform_depth, transformed_depth, pose_matrix=None,
                           pretransform_pix_only=False, **kwargs):
    """environment/utils.py:237-276: the two network-image pixels of an action -> pre-transform pixels -> world points, in
    the reference's return dictionary.  The rotation enters negated, as it does there (the action space is indexed by the
    angle the OBSERVATION was turned by), and a pre-transform pixel (a, b) is read as column a, row b -- both kept on purpose:
    they decide which particle gets grasped."""
    to_source = get_transform_matrix(pretransform_depth.shape[0], transformed_depth.shape[0], -rotation, scale)
    source = (np.column_stack((pixels, np.ones(len(pixels), dtype=np.int64))) @ to_source)[:, :2].astype(int)
    out = {'valid_action': bool(((source >= 0) & (source < pretransform_depth.shape[0])).all()),
           'pretransform_pixels': source.copy()}
    if not out['valid_action']:
        out.update(p1=None, p2=None)
    elif not pretransform_pix_only:
        out['p1'], out['p2'] = back_project(pretransform_depth, source, pose_matrix)
    return out


def preprocess_obs(rgb, d):
    """environment/utils.py:579-582 on the host (the device path produces the same tensor inside fs_observe): colour as
    [0, 1] floats and depth, channels first."""
    planes = torch.cat((torch.as_tensor(rgb).float() / 255, torch.as_tensor(d).float()[..., None]), dim=-1)
    return planes.movedim(-1, 0)


def grasp_pixel_offsets(pix_grasp_dist, pix_drag_dist, pix_place_dist):
    """simEnv.py:517-537 as data: the two grasp pixels of a primitive are the arg-max pixel shifted along the first image
    axis by these amounts (fling / stretchdrag: symmetric about it; drag / place: from it to the end point)."""
    return {'fling': (pix_grasp_dist, -pix_grasp_dist), 'stretchdrag': (pix_grasp_dist, -pix_grasp_dist),
            'drag': (0, pix_drag_dist), 'place': (0, pix_place_dist)}


def get_action_params(action_primitive, max_indices, pix_grasp_dist, pix_drag_dist, pix_place_dist):
    shifts = grasp_pixel_offsets(pix_grasp_dist, pix_drag_dist, pix_place_dist).get(action_primitive)
    if shifts is None:
        raise ValueError(f"unknown action primitive {action_primitive!r}")
    anchor = np.asarray(max_indices)[1:]
    # the reference ASSIGNS the shifted coordinate into a copy of the integer index array (simEnv.py:517-537): with a
    # non-integer pixel distance the SUM is truncated into the anchor's dtype there -- the same store here, whatever the
    # config holds (integer distances, the shipped 8 / 8 / 5, are unaffected)
    out = []
    for shift in shifts:
        p = anchor.copy()
        p[0] = anchor[0] + shift
        out.append(p)
    return tuple(out)


_work = {}


class ActionSelector:
    """Holds what SimEnv keeps between calls: action space (rotations x adaptive scales), pixel distances, arm geometry."""

    def __init__(self, action_primitives, rotations, obs_dim, pix_grasp_dist, pix_drag_dist, pix_place_dist,
                 reach_distance_limit, stretchdrag_dist=0.3, grasp_height=0.02, left_arm_base=(0.765, 0, 0),
                 right_arm_base=(-0.765, 0, 0)):
        self.actions = list(action_primitives)
        self.rotations = list(rotations)
        self.obs_dim = int(obs_dim)
        self.pix_grasp_dist, self.pix_drag_dist, self.pix_place_dist = int(pix_grasp_dist), int(pix_drag_dist), int(pix_place_dist)
        self.reach_distance_limit = float(reach_distance_limit)
        self.stretchdrag_dist, self.grasp_height = float(stretchdrag_dist), float(grasp_height)
        self.left_arm_base, self.right_arm_base = np.array(left_arm_base, np.float64), np.array(right_arm_base, np.float64)
        self.pose = compute_pose(pos=[0, 2, 0], lookat=[0, 0, 0], up=[0, 0, 1])  # simEnv.py:217-221
        self.lib = load_library()
        self._mats = {}

    def _candidate(self, action, x, y, z, scales, depth):
        """Host evaluation of ONE candidate (value-map index x, pixel y, z): the checks of simEnv.py:202-260, :539-558 and
        :604-653 in float64; returns the reference's action_params dictionary, or None when the reference's walk skips it."""
        grasp_pixels = np.array(get_action_params(action, (x, y, z), self.pix_grasp_dist, self.pix_drag_dist,
                                                  self.pix_place_dist))
        if ((grasp_pixels < 0) | (grasp_pixels >= self.obs_dim)).any():
            return None
        rotation_idx, scale_idx = divmod(x, len(scales))
        scale, rotation = scales[scale_idx], self.rotations[rotation_idx]
        found = pixels_to_3d_positions(pixels=grasp_pixels, scale=scale, rotation=rotation, pretransform_depth=depth,
                                       transformed_depth=np.empty((self.obs_dim, 0)), pose_matrix=self.pose)
        if not found['valid_action']:
            return None
        points = np.array([found['p1'], found['p2']])
        arms = {'left': self.left_arm_base, 'right': self.right_arm_base}

        def within_reach(which, targets):  # [arm][point]: is the point closer to that arm's base than the reach limit
            return [np.linalg.norm(arms[which] - t) < self.reach_distance_limit for t in targets]

        arm = None
        if action in ('fling', 'stretchdrag'):  # one arm per point: left takes the first, right the second
            ok = within_reach('left', points[:1])[0] and within_reach('right', points[1:])[0]
            if action == 'stretchdrag':  # ... and again at the end of the drag, at grasp height
                points[:, 1] = self.grasp_height
                sideways = np.cross(points[0] - points[1], np.array([0, 1, 0]))
                sideways = self.stretchdrag_dist * sideways / np.linalg.norm(sideways)
                ok = (within_reach('left', points[:1] + sideways)[0]
                      and within_reach('right', points[1:] + sideways)[0]) and ok
        else:  # drag / place: a single arm has to reach both points; the left one is asked first
            arm = next((name for name in arms if all(within_reach(name, points))), None)
            ok = arm is not None
        if not ok:
            return None
        return dict(valid_action=True, p1=points[0], p2=points[1], pretransform_pixels=found['pretransform_pixels'],
                    left_or_right=arm, scale=scale, rotation=rotation, max_indices=np.array([x, y, z]))

    def select(self, value_maps, adaptive_scale_factors, pretransform_depth, depth_device=None):
        """value_maps: {primitive: CUDA float32 [T, D, D]} or a stacked CUDA tensor [P, T, D, D] in self.actions order;
        pretransform_depth: [S, S] float32 (numpy or tensor); depth_device: the same plane as a CUDA tensor when the caller
        still has it there (the observation stage leaves it on the device: saves re-uploading 640 KB per action).
        Returns (action, action_params) like the reference, or (None, None)."""
        if isinstance(value_maps, dict):
            stacked = torch.stack(tuple(value_maps[a] for a in self.actions))
        else:
            stacked = value_maps
        if not stacked.is_cuda:
            raise RuntimeError("select_action runs on the GPU: pass CUDA value maps")
        stacked = stacked.contiguous().float()
        P, T, D, _ = stacked.shape
        scales = np.asarray(adaptive_scale_factors, np.float64)
        assert D == self.obs_dim and T == len(self.rotations) * len(scales) and P == len(self.actions)
        depth_np = pretransform_depth.detach().cpu().numpy() if torch.is_tensor(pretransform_depth) else np.asarray(pretransform_depth)
        depth_np = np.ascontiguousarray(depth_np, np.float32)
        S = depth_np.shape[0]
        if depth_device is not None and depth_device.is_cuda and tuple(depth_device.shape) == depth_np.shape:
            d_depth = depth_device.contiguous().float()
        elif torch.is_tensor(pretransform_depth) and pretransform_depth.is_cuda:
            d_depth = pretransform_depth.contiguous().float()
        else:
            d_depth = torch.from_numpy(depth_np).to(stacked.device)
        mkey = (S, D, scales.tobytes())
        mats = self._mats.get(mkey)
        if mats is None:  # adaptive scale factors change per observation, but only between a handful of values
            mats = np.ascontiguousarray([get_transform_matrix(original_dim=S, resized_dim=D, rotation=-r, scale=s)
                                         for r in self.rotations for s in scales], np.float64)
            self._mats[mkey] = mats
        kinds = np.ascontiguousarray([KINDS[a] for a in self.actions], np.int32)
        fx = float(compute_intrinsics(CAMERA_FOV_DEG, S)[0, 0])
        nbytes = int(self.lib.fs_select_action_work_bytes(T))
        key = (stacked.device.index, T, threading.get_ident())  # scratch per host thread
        work = _work.get(key)
        if work is None or work.numel() < nbytes:
            work = torch.empty(nbytes, dtype=torch.uint8, device=stacked.device)
            _work[key] = work
        dp = C.POINTER(C.c_double)
        g = self.pix_grasp_dist
        W = D - 2 * g
        pose = np.ascontiguousarray(self.pose, np.float64)
        values = stacked
        for _ in range(8):  # re-run only if the host evaluation of the winner disagrees at a rounding boundary
            best, bval = C.c_longlong(-1), C.c_float(0.0)
            with torch.cuda.device(stacked.device):
                stream = torch.cuda.current_stream().cuda_stream
                rc = self.lib.fs_select_action(
                    C.c_void_p(values.data_ptr()), P, kinds.ctypes.data_as(C.POINTER(C.c_int)), T, D, g,
                    self.pix_drag_dist, self.pix_place_dist, mats.ctypes.data_as(dp), C.c_void_p(d_depth.data_ptr()), S, fx,
                    pose.ctypes.data_as(dp), self.left_arm_base.ctypes.data_as(dp), self.right_arm_base.ctypes.data_as(dp),
                    self.reach_distance_limit, self.stretchdrag_dist, self.grasp_height, C.byref(best), C.byref(bval),
                    C.c_void_p(work.data_ptr()), C.c_void_p(stream))
            if rc != 0:
                raise RuntimeError("fs_select_action: " + self.lib.fs_last_error().decode())
            k = int(best.value)
            if k < 0:
                return None, None
            pidx, x, yy, zz = np.unravel_index(k, (P, T, W, W))
            action = self.actions[pidx]
            params = self._candidate(action, int(x), int(yy) + g, int(zz) + g, scales, depth_np)
            if params is not None:
                params["flat_index"] = k
                params["value"] = float(bval.value)
                return action, params
            if values is stacked:
                values = stacked.clone()
            values[pidx, x, yy + g, zz + g] = float("-inf")  # boundary disagreement: exclude and select again
        raise RuntimeError("select_action: device and host validation keep disagreeing")
