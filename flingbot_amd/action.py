"""Action selection on the device.

`select_action` is the drop-in for SimEnv.get_max_value_valid_action (environment/simEnv.py:560-661): the value maps
stay on the GPU, `fs_select_action` (csrc/fs_action.hip) validates every candidate in parallel and returns the entry the
reference's descending walk would stop at; only that one candidate is then evaluated on the host, with the reference's own
numpy expressions (environment/utils.py:134-276), to build the returned `action_params` (p1, p2, pretransform pixels).
There is no host search: without the HIP library this module raises.
"""
import ctypes as C

import threading

import numpy as np
import torch

from .sim import load_library

KINDS = {"fling": 0, "stretchdrag": 1, "drag": 2, "place": 3}


# ---- environment/utils.py:134-176, :179-234 (same numpy calls, hence the same bits)
def rot2d(angle, degrees=True):
    if degrees:
        angle = np.pi * angle / 180
    return np.array([[np.cos(angle), np.sin(angle), 0], [-np.sin(angle), np.cos(angle), 0], [0, 0, 1]]).T


def translate2d(translation):
    return np.array([[1, 0, translation[0]], [0, 1, translation[1]], [0, 0, 1]]).T


def scale2d(scale):
    return np.array([[scale, 0, 0], [0, scale, 0], [0, 0, 1]]).T


def get_transform_matrix(original_dim, resized_dim, rotation, scale):
    resize_mat = scale2d(original_dim / resized_dim)
    scale_mat = np.matmul(np.matmul(translate2d(-np.ones(2) * (resized_dim // 2)), scale2d(scale)),
                          translate2d(np.ones(2) * (resized_dim // 2)))
    rot_mat = np.matmul(np.matmul(translate2d(-np.ones(2) * (resized_dim // 2)), rot2d(rotation)),
                        translate2d(np.ones(2) * (resized_dim // 2)))
    return np.matmul(np.matmul(scale_mat, rot_mat), resize_mat)


def compute_pose(pos, lookat, up=(0, 0, 1)):
    norm = np.linalg.norm
    lookat, pos, up = np.array(lookat), np.array(pos), np.array(up)
    f = (lookat - pos)
    f = f / norm(f)
    u = up / norm(up)
    s = np.cross(f, u)
    s = s / norm(s)
    u = np.cross(s, f)
    view_matrix = [s[0], u[0], -f[0], 0, s[1], u[1], -f[1], 0, s[2], u[2], -f[2], 0,
                   -np.dot(s, pos), -np.dot(u, pos), np.dot(f, pos), 1]
    view_matrix = np.array(view_matrix).reshape(4, 4).T
    pose_matrix = np.linalg.inv(view_matrix)
    pose_matrix[:, 1:3] = -pose_matrix[:, 1:3]
    return pose_matrix


def compute_intrinsics(fov, image_size):
    image_size = float(image_size)
    focal_length = (image_size / 2) / np.tan((np.pi * fov / 180) / 2)
    return np.array([[focal_length, 0, image_size / 2], [0, focal_length, image_size / 2], [0, 0, 1]])


def pixel_to_3d(depth_im, x, y, pose_matrix, fov=39.5978, depth_scale=1):
    intrinsics_matrix = compute_intrinsics(fov, depth_im.shape[0])
    click_z = depth_im[y, x]
    click_z *= depth_scale
    click_x = (x - intrinsics_matrix[0, 2]) * click_z / intrinsics_matrix[0, 0]
    click_y = (y - intrinsics_matrix[1, 2]) * click_z / intrinsics_matrix[1, 1]
    if click_z == 0:
        raise Exception('Invalid pick point')
    point_3d = np.asarray([click_x, click_y, click_z])
    point_3d = np.append(point_3d, 1.0).reshape(4, 1)
    target_position = np.dot(pose_matrix, point_3d)
    target_position = target_position[0:3, 0]
    target_position[0] = - target_position[0]
    return target_position


def pixels_to_3d_positions(pixels, scale, rotation, pretransform_depth, transformed_depth, pose_matrix=None,
                           pretransform_pix_only=False, **kwargs):
    """environment/utils.py:232-276: network pixels -> pretransform pixels -> world points (same return dictionary)."""
    mat = get_transform_matrix(original_dim=pretransform_depth.shape[0], resized_dim=transformed_depth.shape[0],
                               rotation=-rotation,  # the reference's sign ("TODO bug", environment/utils.py:244), kept
                               scale=scale)
    pixels = np.concatenate((pixels, np.array([[1], [1]])), axis=1)
    pixels = np.matmul(pixels, mat)[:, :2].astype(int)
    pix_1, pix_2 = pixels
    max_idx = pretransform_depth.shape[0]
    if (pixels < 0).any() or (pixels >= max_idx).any():
        return {'valid_action': False, 'p1': None, 'p2': None, 'pretransform_pixels': np.array([pix_1, pix_2])}
    if pretransform_pix_only:
        return {'valid_action': True, 'pretransform_pixels': np.array([pix_1, pix_2])}
    x, y = pix_1  # "this order of x, y is not a bug"
    p1 = pixel_to_3d(depth_im=pretransform_depth, x=x, y=y, pose_matrix=pose_matrix)
    x, y = pix_2
    p2 = pixel_to_3d(depth_im=pretransform_depth, x=x, y=y, pose_matrix=pose_matrix)
    return {'valid_action': p1 is not None and p2 is not None, 'p1': p1, 'p2': p2,
            'pretransform_pixels': np.array([pix_1, pix_2])}


def preprocess_obs(rgb, d):
    """environment/utils.py:579-582 on the host (the device path produces the same tensor inside fs_observe)."""
    return torch.cat((torch.tensor(rgb).float() / 255, torch.tensor(d).unsqueeze(dim=2).float()), dim=2).permute(2, 0, 1)


def get_action_params(action_primitive, max_indices, pix_grasp_dist, pix_drag_dist, pix_place_dist):
    """simEnv.py:517-537"""
    x, y, z = max_indices
    if action_primitive in ('fling', 'stretchdrag'):
        center = np.array([x, y, z])
        p1 = center[1:].copy()
        p1[0] = p1[0] + pix_grasp_dist
        p2 = center[1:].copy()
        p2[0] = p2[0] - pix_grasp_dist
    elif action_primitive == 'drag':
        p1 = np.array([y, z])
        p2 = p1.copy()
        p2[0] += pix_drag_dist
    elif action_primitive == 'place':
        p1 = np.array([y, z])
        p2 = p1.copy()
        p2[0] += pix_place_dist
    else:
        raise Exception(f'Action Primitive not supported: {action_primitive}')
    return p1, p2


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
        """check_action / reachability for ONE candidate with the reference's expressions; None when it is skipped."""
        reach_points = np.array(get_action_params(action, (x, y, z), self.pix_grasp_dist, self.pix_drag_dist,
                                                  self.pix_place_dist))
        if any(((p < 0).any() or (p >= self.obs_dim).any()) for p in reach_points):
            return None
        p1, p2 = reach_points[:2]
        num_scales = len(scales)
        rotation_idx = x // num_scales
        scale_idx = x - rotation_idx * num_scales
        scale, rotation = scales[scale_idx], self.rotations[rotation_idx]
        r3d = pixels_to_3d_positions(pixels=np.array([p1, p2]), scale=scale, rotation=rotation, pretransform_depth=depth,
                                     transformed_depth=np.empty((self.obs_dim, 0)), pose_matrix=self.pose)
        if not r3d['valid_action']:
            return None
        P1, P2 = r3d['p1'], r3d['p2']
        pix_1, pix_2 = r3d['pretransform_pixels']

        def reach(base, pos):
            return np.linalg.norm(base - pos) < self.reach_distance_limit
        left, right = self.left_arm_base, self.right_arm_base
        left_or_right = None
        if action in ('fling', 'stretchdrag'):
            reachable = reach(left, P1) and reach(right, P2)
        elif reach(left, P1) and reach(left, P2):
            reachable, left_or_right = True, 'left'
        elif reach(right, P1) and reach(right, P2):
            reachable, left_or_right = True, 'right'
        else:
            reachable = False
        if action == 'stretchdrag':
            P1[1] = self.grasp_height
            P2[1] = self.grasp_height
            drag_direction = np.cross(P1 - P2, np.array([0, 1, 0]))
            drag_direction = self.stretchdrag_dist * drag_direction / np.linalg.norm(drag_direction)
            reachable = (reach(left, P1 + drag_direction) and reach(right, P2 + drag_direction)) and reachable
        if not reachable:
            return None
        return dict(valid_action=True, p1=P1, p2=P2, pretransform_pixels=np.array([pix_1, pix_2]),
                    left_or_right=left_or_right, scale=scale, rotation=rotation, max_indices=np.array([x, y, z]))

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
        fx = float(compute_intrinsics(39.5978, S)[0, 0])
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
