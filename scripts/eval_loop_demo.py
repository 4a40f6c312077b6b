"""BASELINE.json configs[4] in miniature: the act -> fling loop with every heavy stage on the GPU, for E episodes.

  render (fs_render at S x S)  ->  preprocess_obs  ->  prepare_image (fs_prepare_image, 96 transforms)  ->
  SpatialValueNet forward (PyTorch-ROCm, random-init weights)  ->  action selection (fs_select_action)  ->
  pick_and_fling for all episodes at once (fs_movep_batch + device feedback loops)  ->  wait_until_stable  ->  coverage

Simplifications, stated: the observation is rendered directly at S x S (the reference renders 720 x 720 and resizes with
cv2, absent here); adaptive scaling and the HSV cloth mask (cv2 / skimage) are left out; the grasp-on-cloth flags come from
the depth image (depth != 2.0 at the two pretransform pixels).  No parity claim is attached to this script -- its stages
are pinned one by one in tests/; it measures what the composition costs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
from fling_helpers import picker_centres
from flingbot_amd import sim as fsim, nets
from flingbot_amd.action import ActionSelector
from flingbot_amd.primitives import FlingPrimitives


def main(E=32, S=128, D=64, n_actions=2, seed=0):
    dev = torch.device("cuda:0")
    ctx = fsim.FlingSim(n_envs=E, camera_width=S, camera_height=S, solver=0)
    for e in range(E):
        env = ctx.env(e)
        bench.setup_episode(env, e)
        cp = ctx.get_camera_params(e)                       # [w, h, pos, angle] -> render at S x S
        ctx.set_camera_params(e, [*cp[2:8], S, S])
        for c in picker_centres():
            env.add_sphere(0.02, c, [1, 0, 0, 0])
        st = np.array(env.get_shape_states()).reshape(-1, 14)
        for i, c in enumerate([[0.5, 0.5, -0.5], [-0.5, 0.5, -0.5]]):  # reset_end_effectors pose
            st[i] = np.hstack([c, c, [1, 0, 0, 0], [1, 0, 0, 0]])
        env.set_shape_states(st)
        ctx.picker_reset(e)
    ctx.step(60)                                   # let the sheets fall and crumple
    ctx.wait_until_stable(range(E), max_steps=200)
    torch.manual_seed(seed)
    scales = [1.0, 1.25, 1.5, 1.75, 2.0, 2.25, 2.5, 2.75]
    policy = nets.MaximumValuePolicy(action_primitives=["fling"], num_rotations=12, scale_factors=scales, obs_dim=D,
                                     pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, action_expl_prob=0.0,
                                     action_expl_decay=1.0, value_expl_prob=0.0, value_expl_decay=1.0, device=dev)
    net = policy.value_nets["fling"].fold_batchnorm()
    selector = ActionSelector(["fling"], policy.rotations, D, 8, 8, 5, reach_distance_limit=1.0)
    tf = [(r, s) for r in policy.rotations for s in scales]
    prim = FlingPrimitives(ctx, range(E))
    cov0 = np.array(ctx.coverage())
    t = dict(render=0.0, prepare=0.0, net=0.0, select=0.0, fling=0.0, settle=0.0)
    ctx.sync(); torch.cuda.synchronize(); t_all = time.perf_counter()
    flung = 0
    for a in range(n_actions):
        p1s, p2s, g1, g2 = [], [], [], []
        for e in range(E):
            t0 = time.perf_counter()
            rgba, depth = ctx.render(e)
            rgb = np.flip(rgba.reshape(S, S, 4), 0)[:, :, :3]
            d = np.flip(depth.reshape(S, S), 0).copy()
            obs = torch.cat((torch.tensor(rgb.copy()).float() / 255, torch.tensor(d).unsqueeze(2)), dim=2).permute(2, 0, 1).to(dev)
            t1 = time.perf_counter()
            stack = nets.prepare_image(obs, tf, D)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            with torch.no_grad():
                vmap = net(stack).squeeze(1)[None]           # [1, T, D, D]
            torch.cuda.synchronize(); t3 = time.perf_counter()
            action, params = selector.select(vmap, scales, d)
            t4 = time.perf_counter()
            t["render"] += t1 - t0; t["prepare"] += t2 - t1; t["net"] += t3 - t2; t["select"] += t4 - t3
            if action is None:
                p1s.append([0, 0, 0]); p2s.append([0, 0, 0]); g1.append(False); g2.append(False)
                continue
            pix = params["pretransform_pixels"]
            p1s.append(params["p1"]); p2s.append(params["p2"])
            g1.append(bool(d[pix[0][1], pix[0][0]] != 2.0)); g2.append(bool(d[pix[1][1], pix[1][0]] != 2.0))
        t0 = time.perf_counter()
        out = prim.pick_and_fling(np.array(p1s), np.array(p2s), g1, g2)
        ctx.sync(); t1 = time.perf_counter()
        ctx.wait_until_stable(range(E), max_steps=300)
        ctx.sync(); t2 = time.perf_counter()
        t["fling"] += t1 - t0; t["settle"] += t2 - t1
        flung += sum(1 for o in out if o["dist"] is not None)
    total = time.perf_counter() - t_all
    cov1 = np.array(ctx.coverage())
    print("E=%d episodes x %d actions: %.2f s (%.1f actions/s); flings executed %d; simulation steps %d; mean coverage "
          "%.4f -> %.4f m^2" % (E, n_actions, total, E * n_actions / total, flung, prim.sim_steps, cov0.mean(), cov1.mean()))
    print("  per stage [s]: " + ", ".join("%s %.2f" % kv for kv in t.items()))
    return dict(total=total, flung=flung, cov0=cov0, cov1=cov1, stages=t)


if __name__ == "__main__":
    main(E=int(sys.argv[1]) if len(sys.argv) > 1 else 32)
