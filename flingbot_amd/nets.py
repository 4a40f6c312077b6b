"""Value-map networks and the rotate/scale observation stack, with the module surface of the reference's
learning/nets.py (BasicBlock :12, ResidualBlock :44, SpatialValueNet :81, crop_center :144, pad :150, transform :155,
prepare_image :180, Policy :196, MaximumValuePolicy :232) so that `flingbot.pth` checkpoints load unchanged
(state_dict keys: SURVEY.md 8b) and callers (run_sim.py, environment/simEnv.py) keep working.

MI355X notes: the CNN is 18 3x3 convolutions with 16 channels on 64x64 maps -- memory/launch bound, so the forward
runs channels-last on PyTorch-ROCm (MIOpen -> MFMA) with eval-mode BatchNorm folded into the convolutions
(`SpatialValueNet.fold_batchnorm`), and `MaximumValuePolicy.act` batches all environments into ONE forward instead of
looping per environment (nets.py:228-229).  cv2 / ray are not needed: padding and nearest resize are restated with
numpy following OpenCV's conventions (BORDER_REPLICATE; INTER_NEAREST source index = floor(dst * src/dst)).
"""
import random
import threading
from time import time
from typing import List

import numpy as np
import torch
import torch.nn as nn
from scipy import ndimage as nd


class BasicBlock(nn.Module):
    """conv3x3 (no bias) [+ BatchNorm + non-linearity]; sub-module name `net` is part of the checkpoint layout."""

    def __init__(self, inplanes, planes, kernel_size, stride, padding=1, norm_layer=None, non_linearity=nn.LeakyReLU):
        super().__init__()
        layers = [nn.Conv2d(inplanes, planes, kernel_size=kernel_size, stride=stride, padding=padding, bias=False)]
        if non_linearity is not None:
            layers += [nn.BatchNorm2d(planes), non_linearity()]
        self.net = nn.Sequential(*layers)

    def forward(self, input):
        return self.net(input)


class ResidualBlock(nn.Module):
    """y = relu(bn2(conv2(relu(bn1(conv1(x))))) + x); attribute names conv1/bn1/relu/conv2/bn2 are checkpoint keys."""

    def __init__(self, inplanes, planes, kernel_size, stride, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self.planes = planes
        self.stride = stride
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=kernel_size, stride=stride, padding=1, bias=False)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=kernel_size, stride=stride, padding=1, bias=False)
        self.bn2 = norm_layer(planes)

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        out = out + x
        return self.relu(out)


_vn_work = {}


class SpatialValueNet(nn.Module):
    def __init__(self, rgb_only=False, depth_only=False, steps=0, device='cuda', **kwargs):
        super().__init__()
        self.device = device
        self.rgb_only = rgb_only
        self.depth_only = depth_only
        self.input_channels = 3 if rgb_only else (1 if depth_only else 4)
        self.net = self.setup_net()
        mean = torch.tensor([0.18, 0.18, 0.18, 1.99])
        std = torch.tensor([0.1, 0.1, 0.1, 0.006])
        if rgb_only:
            mean, std = mean[:3], std[:3]
        elif depth_only:
            mean, std = mean[3], std[3]
        self.mean, self.std = mean, std  # plain attributes, not buffers: they are not in the reference's state_dict
        self.steps = nn.parameter.Parameter(torch.tensor(steps), requires_grad=False)
        self._folded = None
        self._hip = None
        self._fold_stale = False  # parameters may have changed since fold_batchnorm(): re-fold before the next eval forward
        self._fold_hip = None     # the `hip` argument of the last fold_batchnorm() call

    # The folded copies are derived data.  Everything that can change the parameters they were derived from marks them
    # stale -- loading a state_dict (also through a parent module: nn.Module.load_state_dict calls this hook on every
    # submodule), a train() phase (optimizer steps), .to() / .cuda() / .float() (which also moves nothing that is not
    # registered) -- and the next eval-mode forward folds again.  Writing to `p.data` by hand in eval mode is the one case
    # left to the caller: call fold_batchnorm() again.
    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._fold_stale = True

    def train(self, mode=True):
        if mode:
            self._fold_stale = True
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._fold_stale = True
        return out

    def setup_net(self):
        blocks = [BasicBlock(self.input_channels, 16, 3, 1)]
        blocks += [ResidualBlock(16, 16, 3, 1) for _ in range(8)]
        blocks += [BasicBlock(16, 1, 3, 1, non_linearity=None)]
        return nn.Sequential(*blocks)

    def preprocess_obs(self, obs):
        assert obs.dim() == 4
        c = obs.shape[1]
        if self.rgb_only:
            if c == 4:
                obs = obs[:, :3]
            elif c != 3:
                raise Exception
        elif self.depth_only:
            obs = obs[:, 3:4] if c == 4 else obs.squeeze().unsqueeze(dim=-3)
        mean = self.mean.to(obs.device).reshape(1, -1, 1, 1)
        std = self.std.to(obs.device).reshape(1, -1, 1, 1)
        return (obs - mean) / std

    def forward(self, obs):
        if self._folded is not None and not self.training:
            if self._fold_stale:
                self.fold_batchnorm(hip=self._fold_hip)
            if self._hip is not None and obs.is_cuda and obs.dim() == 4 and tuple(obs.shape[-2:]) == (64, 64):
                return self._forward_hip(obs)
            return self._folded(self.preprocess_obs(obs).contiguous(memory_format=torch.channels_last))
        return self.net(self.preprocess_obs(obs))

    def _forward_hip(self, obs):
        """The whole forward (normalisation included) in libflingsim's fs_value_net_forward (csrc/fs_valuenet.hip):
        10 launches, the 16 -> 16 convolutions of a residual block fused in LDS on fp32 MFMA."""
        import ctypes as C
        lib, params = self._hip
        c = int(obs.shape[1])
        if self.rgb_only:
            if c not in (3, 4):
                raise Exception
            off = 0
        elif self.depth_only:
            if c not in (1, 4):
                raise Exception
            off = 3 if c == 4 else 0
        else:
            off = 0
        if off + self.input_channels > c:
            raise Exception
        obs = obs.contiguous().float()
        if params.device != obs.device:
            params = params.to(obs.device)
            object.__setattr__(self, '_hip', (lib, params))
        batch = int(obs.shape[0])
        out = torch.empty((batch, 1, 64, 64), dtype=torch.float32, device=obs.device)
        if batch == 0:
            return out
        nbytes = int(lib.fs_value_net_work_bytes(batch, 64))
        key = (obs.device.index, threading.get_ident())  # per host thread: launches of two threads interleave
        work = _vn_work.get(key)
        if work is None or work.numel() < nbytes:
            work = torch.empty(nbytes, dtype=torch.uint8, device=obs.device)
            _vn_work[key] = work
        with torch.cuda.device(obs.device):
            stream = torch.cuda.current_stream().cuda_stream
            rc = lib.fs_value_net_forward(C.c_void_p(params.data_ptr()), C.c_void_p(obs.data_ptr()), c, off,
                                          self.input_channels, batch, 64, C.c_void_p(out.data_ptr()),
                                          C.c_void_p(work.data_ptr()), C.c_void_p(stream))
        if rc != 0:
            raise RuntimeError("fs_value_net_forward: " + lib.fs_last_error().decode())
        return out

    # ---- inference fast path -------------------------------------------------------------------------------------
    def fold_batchnorm(self, hip=None):
        """Build an eval-only copy of `net` with every BatchNorm folded into the preceding convolution
        (w' = w * g / sqrt(var + eps), b' = beta - mean * g / sqrt(var + eps)) in channels-last layout.
        18 conv + 17 BN + activations become 18 conv(+bias) launches.  The parameters of `net` are untouched, so
        state_dict() keeps the reference layout.
        hip: also pack the folded weights for the hand-written forward (fs_value_net_forward), which then serves CUDA
        observations of 64 x 64 pixels; default = whenever the parameters live on a GPU.  Call again after loading
        new weights."""
        self.eval()
        self._fold_hip = hip

        def fold(conv, bn):
            out = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, bias=True)
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            out.weight.data = conv.weight.data * scale.reshape(-1, 1, 1, 1)
            out.bias.data = bn.bias.data - bn.running_mean * scale
            return out

        class FoldedResidual(nn.Module):
            def __init__(self, blk):
                super().__init__()
                self.c1, self.c2 = fold(blk.conv1, blk.bn1), fold(blk.conv2, blk.bn2)

            def forward(self, x):
                return torch.relu(self.c2(torch.relu(self.c1(x))) + x)

        first, last = self.net[0].net, self.net[-1].net
        layers = [fold(first[0], first[1]), type(first[2])()]
        layers += [FoldedResidual(b) for b in list(self.net)[1:-1]]
        tail = nn.Conv2d(last[0].in_channels, last[0].out_channels, 3, 1, 1, bias=False)
        tail.weight.data = last[0].weight.data.clone()
        layers.append(tail)
        folded = nn.Sequential(*layers).to(next(self.net.parameters()).device).eval()
        folded = folded.to(memory_format=torch.channels_last)
        for p in folded.parameters():
            p.requires_grad_(False)
        object.__setattr__(self, '_folded', folded)  # not registered: keeps state_dict identical to the reference
        dev = next(self.net.parameters()).device
        if hip is None:
            hip = dev.type == 'cuda'
        object.__setattr__(self, '_hip', self._pack_hip(folded, dev) if hip else None)
        self._fold_stale = False
        return self

    def _pack_hip(self, folded, dev):
        import ctypes as C
        from .sim import load_library
        lib = load_library()  # raises when libflingsim is missing: no silent change of path
        fp = C.POINTER(C.c_float)

        def host(t):
            return np.ascontiguousarray(t.detach().cpu().numpy(), np.float32)

        blocks = list(folded)[2:-1]
        w_blocks = np.stack([host(c.weight) for b in blocks for c in (b.c1, b.c2)])
        b_blocks = np.stack([host(c.bias) for b in blocks for c in (b.c1, b.c2)])
        assert w_blocks.shape == (16, 16, 16, 3, 3) and b_blocks.shape == (16, 16)
        mean = np.ascontiguousarray(np.atleast_1d(self.mean.numpy()), np.float32)
        std = np.ascontiguousarray(np.atleast_1d(self.std.numpy()), np.float32)
        w_first, b_first, w_last = host(folded[0].weight), host(folded[0].bias), host(folded[-1].weight)
        packed = np.zeros(int(lib.fs_value_net_param_floats()), np.float32)
        rc = lib.fs_value_net_pack(self.input_channels, mean.ctypes.data_as(fp), std.ctypes.data_as(fp),
                                   w_first.ctypes.data_as(fp), b_first.ctypes.data_as(fp), w_blocks.ctypes.data_as(fp),
                                   b_blocks.ctypes.data_as(fp), w_last.ctypes.data_as(fp), packed.ctypes.data_as(fp))
        if rc != 0:
            raise RuntimeError("fs_value_net_pack: " + lib.fs_last_error().decode())
        params = torch.from_numpy(packed)
        return lib, (params.to(dev) if dev.type == 'cuda' else params)


def _centre_window(size, extent):
    """First index of the `extent`-wide window that crop_center cuts out of `size` samples."""
    return size // 2 - extent // 2


def crop_center(img, crop):
    """Module surface of learning/nets.py (:144-147), semantics fixed by it: the crop x crop window around the centre of
    the first two axes (centre = size // 2, window start = centre - crop // 2)."""
    r0, c0 = _centre_window(img.shape[0], crop), _centre_window(img.shape[1], crop)
    return img[r0:r0 + crop, c0:c0 + crop, ...]


def pad(img, size):
    """cv2.copyMakeBorder(img, n, n, n, n, BORDER_REPLICATE) with n = (size - h) // 2 (nets.py:150-152)."""
    n = (size - img.shape[0]) // 2
    widths = [(n, n), (n, n)] + [(0, 0)] * (img.ndim - 2)
    return np.pad(img, widths, mode='edge')


def resize_nearest(img, dim):
    """cv2.resize(img, (dim, dim), interpolation=INTER_NEAREST): source index = floor(dst * src / dst_size)."""
    h, w = img.shape[:2]
    ys = np.minimum((np.arange(dim) * (h / dim)).astype(np.int64), h - 1)
    xs = np.minimum((np.arange(dim) * (w / dim)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def scale_window_indices(size, scale, dim):
    """Which of the `size` samples of a rotated plane each of the `dim` output samples shows after the reference's
    crop / pad + nearest-resize chain (nets.py:163-171), as ONE index vector: the chain cuts (scale < 1) or replicates
    (scale > 1) the plane to a window [start, start + extent) and the nearest resize then picks window sample
    floor(d * extent / dim); replicated border samples are the clamped ones.  fs_prepare_image's gather kernel uses the
    same map (csrc/fs_image.hip)."""
    target = int(scale * size)
    if scale < 1:
        start = _centre_window(size, target)
        extent = min(target, size - start)
    elif scale > 1:
        border = (target - size) // 2
        start, extent = -border, size + 2 * border
    else:
        start, extent = 0, size
    picked = np.minimum((np.arange(dim) * (extent / dim)).astype(np.int64), extent - 1)
    return np.clip(start + picked, 0, size - 1)


def transform(img, rotation: float, scale: float, dim: int):
    """One rotated / scaled copy of the observation, the host form of learning/nets.py:155-174 (the hot path is
    fs_prepare_image on the device): channel-first square tensor -> (W, H, C) array (the reference's permute(2, 1, 0)),
    scipy's cubic-spline rotation about the centre with mode='nearest', then the crop / pad + nearest-resize chain as
    one gather (scale_window_indices), back to channel-first."""
    channel_first = len(img.shape) == 3 and (img.shape[-1] == img.shape[-2])
    plane = img.permute(2, 1, 0) if channel_first else img
    rotated = nd.rotate(input=plane, angle=rotation, reshape=False, mode='nearest')
    rows = scale_window_indices(rotated.shape[0], scale, dim)
    cols = scale_window_indices(rotated.shape[1], scale, dim) if rotated.shape[1] != rotated.shape[0] else rows
    out = rotated[rows][:, cols]
    if out.ndim == 3:
        out = out.swapaxes(-1, 0)
    return torch.tensor(np.ascontiguousarray(out))


def transform_async(*args, **kwargs):
    """The reference wraps `transform` in ray.remote; without ray this is the same function run inline."""
    return transform(*args, **kwargs)


def rotation_matrices(rotations, size):
    """Matrix rows and offset of scipy.ndimage.rotate(angle, reshape=False) for a size x size plane (scipy
    _interpolation.py: c, s = cosdg, sindg; [[c, s], [-s, c]]; offset = centre - matrix @ centre)."""
    from scipy import special
    mats, offs = [], []
    for ang in rotations:
        c, s_ = special.cosdg(ang), special.sindg(ang)
        m = np.array([[c, s_], [-s_, c]])
        centre = (np.array([size, size]) - 1) / 2
        mats.append(m.ravel())
        offs.append(centre - m @ centre)
    return np.ascontiguousarray(mats, np.float64), np.ascontiguousarray(offs, np.float64)


_prep_work = {}
_prep_mats = {}


def prepare_image_device(img, transformations, dim: int):
    """prepare_image (nets.py:177-193) for an observation that already lives on the GPU: one spline prefilter of the
    observation + one gather kernel for all transforms (libflingsim fs_prepare_image, csrc/fs_image.hip) instead of
    len(transformations) full-size scipy rotations on the host.  Returns a float32 CUDA tensor [T, C, dim, dim]."""
    import ctypes as C
    from .sim import load_library

    lib = load_library()
    assert img.is_cuda and img.dim() == 3 and img.shape[-1] == img.shape[-2], "expects a CUDA (C, S, S) observation"
    img = img.contiguous().float()
    ch, size = int(img.shape[0]), int(img.shape[-1])
    rots = tuple(float(t[0]) for t in transformations)
    scales = np.ascontiguousarray([float(t[1]) for t in transformations], np.float64)
    mkey = (rots, size)
    if mkey not in _prep_mats:  # a policy asks for the same rotations at every observation
        if len(_prep_mats) > 16:
            _prep_mats.clear()
        _prep_mats[mkey] = rotation_matrices(rots, size)
    mats, offs = _prep_mats[mkey]
    n = len(rots)
    key = (img.device.index, ch, size, n, threading.get_ident())  # scratch per host thread
    nbytes = int(lib.fs_prepare_image_work_bytes(ch, size, n))
    work = _prep_work.get(key)
    if work is None or work.numel() < nbytes:
        work = torch.empty(nbytes, dtype=torch.uint8, device=img.device)
        _prep_work[key] = work
    out = torch.empty((n, ch, dim, dim), dtype=torch.float32, device=img.device)
    dp = C.POINTER(C.c_double)
    with torch.cuda.device(img.device):
        stream = torch.cuda.current_stream().cuda_stream
        rc = lib.fs_prepare_image(C.c_void_p(img.data_ptr()), ch, size, n, mats.ctypes.data_as(dp), offs.ctypes.data_as(dp),
                                  scales.ctypes.data_as(dp), int(dim), C.c_void_p(out.data_ptr()),
                                  C.c_void_p(work.data_ptr()), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError("fs_prepare_image: " + lib.fs_last_error().decode())
    return out


def prepare_image(img, transformations, dim: int, parallelize=False, log=False):
    if log:
        start = time()
        print('preparing images')
    if torch.is_tensor(img) and img.is_cuda:  # device-resident observation: the HIP path (no host copy, no scipy)
        retval = prepare_image_device(img, transformations, dim)
        if log:
            print(f'prepare_image took {float(time() - start):.02f}s')
        return retval
    imgs = [transform(img, *t, dim=dim) for t in transformations]
    retval = torch.stack(imgs).float()
    if log:
        print(f'prepare_image took {float(time() - start):.02f}s')
    return retval


class Policy:
    def __init__(self, action_primitives: List[str], num_rotations: int, scale_factors: List[float], obs_dim: int,
                 pix_grasp_dist: int, pix_drag_dist: int, pix_place_dist: int, **kwargs):
        assert len(action_primitives) > 0
        self.action_primitives = action_primitives
        # rotation angles in degrees, counter-clockwise: fling covers [-90, 90], the others the full circle
        if 'fling' in action_primitives:
            self.rotations = [(2 * i / (num_rotations - 1) - 1) * 90 for i in range(num_rotations)]
        else:
            self.rotations = [(2 * i / num_rotations - 1) * 180 for i in range(num_rotations)]
        self.scale_factors = scale_factors
        self.num_transforms = len(self.rotations) * len(self.scale_factors)
        self.obs_dim = obs_dim
        self.pix_grasp_dist = pix_grasp_dist
        self.pix_drag_dist = pix_drag_dist
        self.pix_place_dist = pix_place_dist

    def get_action_single(self, obs):
        raise NotImplementedError()

    def act(self, obs):
        return [self.get_action_single(o) for o in obs]


class MaximumValuePolicy(nn.Module, Policy):
    def __init__(self, action_expl_prob: float, action_expl_decay: float, value_expl_prob: float,
                 value_expl_decay: float, device=None, **kwargs):
        super().__init__()
        Policy.__init__(self, **kwargs)
        if device is None:
            self.device = torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')
        else:
            self.device = torch.device(device)
        as_param = lambda v: nn.parameter.Parameter(torch.tensor(v), requires_grad=False)
        self.action_expl_prob = as_param(action_expl_prob)
        self.action_expl_decay = as_param(action_expl_decay)
        self.value_expl_prob = as_param(value_expl_prob)
        self.value_expl_decay = as_param(value_expl_decay)
        # one value net per action primitive
        self.value_nets = nn.ModuleDict({key: SpatialValueNet(device=self.device, **kwargs).to(self.device)
                                         for key in self.action_primitives})
        self.should_explore_action = lambda: self.action_expl_prob > random.random()
        self.should_explore_value = lambda: self.value_expl_prob > random.random()
        self.eval()

    def decay_exploration(self):
        self.action_expl_prob *= self.action_expl_decay
        self.value_expl_prob *= self.value_expl_decay

    def random_value_map(self, device=None):
        return torch.rand(len(self.rotations) * len(self.scale_factors), self.obs_dim, self.obs_dim, device=device)

    def _explore(self, value_maps):
        """Value / action exploration on one environment's dict of value maps (nets.py:279-293)."""
        value_maps = {k: (v if not self.should_explore_value() else self.random_value_map(v.device))
                      for k, v in value_maps.items()}
        if self.should_explore_action():
            random_action, action_val_map = random.choice(list(value_maps.items()))
            min_val = action_val_map.min()
            value_maps = {k: (v if k == random_action else torch.ones(v.size(), device=v.device) * min_val)
                          for k, v in value_maps.items()}
        return value_maps

    def get_action_single(self, obs):
        return self.act([obs])[0]

    def act(self, obs, keep_on_device=False):
        """list of [T,4,D,D] observation stacks -> list of {primitive: [T,D,D] cpu tensor}.  All environments go through
        each value net in ONE batched forward (the reference loops per environment).  keep_on_device (additive): leave the
        value maps on the policy's device for a consumer that selects the action there (action.ActionSelector)."""
        if len(obs) == 0:
            return []
        with torch.no_grad():
            sizes = [o.shape[0] for o in obs]
            batch = torch.cat([o.to(self.device, non_blocking=True) for o in obs], dim=0)
            outs = {}
            for k, net in self.value_nets.items():
                maps = net(batch).squeeze(1)
                outs[k] = (maps if keep_on_device else maps.cpu()).split(sizes)
            return [self._explore({k: outs[k][e] for k in outs}) for e in range(len(obs))]

    def steps(self):
        return sum([net.steps for net in self.value_nets.values()])

    def forward(self, obs):
        return self.act(obs)
