"""flingbot_amd -- MI355X-native FlingBot cloth hot path (solver + rasteriser + value-map CNN).

The compute lives in hand-written HIP kernels behind the C-ABI of include/flingsim.h; this package is the thin host
side: `sim` (batched ctypes binding), `pyflex_native/pyflex` (pybind11 module with the reference's pyflex surface),
`nets` (learning/nets.py module surface on PyTorch-ROCm, incl. the device `prepare_image`), `primitives` (batched
pick-and-fling on the device-side movep / feedback loops), `action` (device action selection), `tasks` (batched task generation), `env` (SimEnv.reset / step for many episodes), `distributed` (one process
per GPU helpers).
"""
__version__ = "0.1.0"
