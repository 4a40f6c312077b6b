"""CPU ORACLE (test infrastructure): numpy restatement of the reference's action selection.

Follows, expression by expression (same dtypes, same np.matmul / np.dot / np.linalg.norm calls):
    SimEnv.get_max_value_valid_action   environment/simEnv.py:560-661
    SimEnv.get_action_params            environment/simEnv.py:517-537
    SimEnv.check_action                 environment/simEnv.py:202-260   (conservative_grasp_radius = 0 branch)
    SimEnv.check_action_reachability    environment/simEnv.py:539-558
    pixels_to_3d_positions, get_transform_matrix, rot2d / translate2d / scale2d, compute_pose, compute_intrinsics,
    pixel_to_3d                         environment/utils.py:134-276
Pinned by tests/golden/action_golden.npz: the reference's own SimEnv method run on synthetic value maps / depth images
(tests/golden/make_golden.py action).  Exhaustive and slow (sorts all candidates, Python loop) -- small cases only.
"""
import numpy as np


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


def get_action_params(action_primitive, max_indices, pix_grasp_dist, pix_drag_dist, pix_place_dist):
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


def evaluate_candidate(action, x, y, z, cfg):
    """Everything the reference does for one candidate (primitive `action`, transform x, pixel (y, z)).
    Returns None when the candidate is skipped, else dict(p1, p2, pretransform_pixels, left_or_right)."""
    reach_points = np.array(get_action_params(action, (x, y, z), cfg['pix_grasp_dist'], cfg['pix_drag_dist'],
                                              cfg['pix_place_dist']))
    if any(((p < 0).any() or (p >= cfg['obs_dim']).any()) for p in reach_points):
        return None
    p1, p2 = reach_points[:2]
    num_scales = len(cfg['scales'])
    rotation_idx = x // num_scales
    scale_idx = x - rotation_idx * num_scales
    scale = cfg['scales'][scale_idx]
    rotation = cfg['rotations'][rotation_idx]
    depth = cfg['depth']
    mat = get_transform_matrix(original_dim=depth.shape[0], resized_dim=cfg['obs_dim'], rotation=-rotation, scale=scale)
    pixels = np.concatenate((np.array([p1, p2]), np.array([[1], [1]])), axis=1)
    pixels = np.matmul(pixels, mat)[:, :2].astype(int)
    pix_1, pix_2 = pixels
    if (pixels < 0).any() or (pixels >= depth.shape[0]).any():
        return None
    pose = compute_pose(pos=[0, 2, 0], lookat=[0, 0, 0], up=[0, 0, 1])
    xx, yy = pix_1
    P1 = pixel_to_3d(depth_im=depth.copy(), x=xx, y=yy, pose_matrix=pose)
    xx, yy = pix_2
    P2 = pixel_to_3d(depth_im=depth.copy(), x=xx, y=yy, pose_matrix=pose)

    def reach(base, pos):
        return np.linalg.norm(base - pos) < cfg['reach_distance_limit']
    left, right = cfg['left_arm_base'], cfg['right_arm_base']
    left_or_right = None
    if action in ('fling', 'stretchdrag'):
        reachable = reach(left, P1) and reach(right, P2)
    else:
        if reach(left, P1) and reach(left, P2):
            reachable, left_or_right = True, 'left'
        elif reach(right, P1) and reach(right, P2):
            reachable, left_or_right = True, 'right'
        else:
            reachable = False
    if action == 'stretchdrag':
        P1[1] = cfg['grasp_height']
        P2[1] = cfg['grasp_height']
        drag_direction = np.cross(P1 - P2, np.array([0, 1, 0]))
        drag_direction = cfg['stretchdrag_dist'] * drag_direction / np.linalg.norm(drag_direction)
        reachable = (reach(left, P1 + drag_direction) and reach(right, P2 + drag_direction)) and reachable
    if not reachable:
        return None
    return dict(p1=P1, p2=P2, pretransform_pixels=np.array([pix_1, pix_2]), left_or_right=left_or_right)


def get_max_value_valid_action(values, actions, cfg):
    """values: float32 [P, T, D, D] (stacked value maps in the order of `actions`).  Returns
    (action, dict(p1, p2, ...), flat index into the edge-cropped stack) or (None, None, -1)."""
    g = cfg['pix_grasp_dist']
    cropped = values[:, :, g:-g, g:-g]
    flat = cropped.reshape(-1)
    order = np.argsort(-flat, kind='stable')  # descending value, ascending flat index among equals
    shape = cropped.shape
    for k in order:
        pidx, x, yy, zz = np.unravel_index(k, shape)
        res = evaluate_candidate(actions[pidx], int(x), int(yy) + g, int(zz) + g, cfg)
        if res is not None:
            return actions[pidx], res, int(k)
    return None, None, -1
