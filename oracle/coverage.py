"""CPU ORACLE (test infrastructure): coverage reward.

numpy restatement of get_current_covered_area (reference environment/flex_utils.py:358-395, helpers
vectorized_range :262-268 and vectorized_meshgrid :255-260).  Pinned against vectors generated from the reference's own
function in this container (tests/golden/coverage_golden.npz, made by tests/golden/make_golden.py).
"""
import numpy as np


def _vrange(start, end):
    n = int(np.max(end - start)) + 1
    return np.floor(np.arange(n) * (end - start)[:, None] / n + start[:, None]).astype("int")


def _vmeshgrid(vx, vy):
    n, k, d = vx.shape[0], vx.shape[1], vy.shape[1]
    vx = np.tile(vx[:, None, :], [1, d, 1]).reshape(n, -1)
    vy = np.tile(vy[:, :, None], [1, 1, k]).reshape(n, -1)
    return vx, vy


def covered_area(pos, cloth_particle_radius=0.00625):
    """pos: float[4N] or [N,4] in the dtype the caller holds (float32 from pyflex.get_positions)."""
    pos = np.reshape(pos, [-1, 4])
    min_x, min_y = np.min(pos[:, 0]), np.min(pos[:, 2])
    max_x, max_y = np.max(pos[:, 0]), np.max(pos[:, 2])
    init = np.array([min_x, min_y])
    span = np.array([max_x - min_x, max_y - min_y]) / 100.
    pos2d = pos[:, [0, 2]]
    offset = pos2d - init
    x_lo = np.maximum(np.round((offset[:, 0] - cloth_particle_radius) / span[0]).astype(int), 0)
    x_hi = np.minimum(np.round((offset[:, 0] + cloth_particle_radius) / span[0]).astype(int), 100)
    y_lo = np.maximum(np.round((offset[:, 1] - cloth_particle_radius) / span[1]).astype(int), 0)
    y_hi = np.minimum(np.round((offset[:, 1] + cloth_particle_radius) / span[1]).astype(int), 100)
    grid = np.zeros(10000)
    xx, yy = _vmeshgrid(_vrange(x_lo, x_hi), _vrange(y_lo, y_hi))
    idx = np.clip((xx * 100 + yy).flatten(), 0, 9999)
    grid[idx] = 1
    return np.sum(grid) * span[0] * span[1]
