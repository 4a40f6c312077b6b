"""flingbot_amd -- MI355X-native FlingBot cloth hot path (solver + rasteriser + value-map CNN).

The compute lives in hand-written HIP kernels behind the C-ABI of include/flingsim.h; this package is the thin host
side:
  sim          batched ctypes binding (FlingSim: pyflex-shaped calls per episode, batched step / movep / feedback loops,
               device-side observation stage)
  pyflex_native/pyflex   pybind11 module with the reference's `pyflex` surface
  nets         learning/nets.py module surface (PyTorch-ROCm modules, reference state_dict layout); inference forward on
               the hand-written fp32-MFMA kernels, `prepare_image` on the device
  action       device action selection (SimEnv.get_max_value_valid_action)
  primitives   batched pick-and-fling / drag / place / stretch-drag on the device-side movep
  tasks        batched task generation, quad-mesh .obj loading
  env          SimEnv.reset / step for many episodes
  evaluate     the run_sim.py evaluation loop + collect_stats' statistics, optionally sharded over ranks
  distributed  one-process-per-GPU helpers (episode sharding, coverage all_gather)
"""
__version__ = "0.1.0"
