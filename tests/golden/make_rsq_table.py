"""Dump what v_rsq_f32 returns on this chip for every (exponent parity, mantissa) -- the 2^24 inputs in [1, 4) -- as the
difference in ulps from the CPU formula  float32(1 / sqrt(float64(x)))  (IEEE double sqrt and division + one rounding: the
same bits on every x86-64), check that other exponents only rescale the result, and look at zero / denormal inputs.
Run on the GPU box (the shipped library's fs_rsqrt IS the hardware instruction), then copy the file to oracle/:
    python tests/golden/make_rsq_table.py gpurun_out/rsq/v_rsq_f32_gfx950.npz
"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # the repository root
import numpy as np
from flingbot_amd import sim as fsim


def cpu_formula(x):
    return (1.0 / np.sqrt(x.astype(np.float64))).astype(np.float32)


out = sys.argv[1]
ctx = fsim.FlingSim(n_envs=1)
bits = (np.arange(1 << 24, dtype=np.uint32) + np.uint32(127 << 23))          # [1, 2) then [2, 4): parity | mantissa
x = bits.view(np.float32)
hw = np.concatenate([ctx.eval_rsqrt(x[k:k + (1 << 22)]) for k in range(0, 1 << 24, 1 << 22)])
ref = cpu_formula(x)
delta = (hw.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
print("delta range", delta.min(), delta.max(), "histogram", {int(v): int((delta == v).sum()) for v in np.unique(delta)})
assert delta.min() >= -2 and delta.max() <= 1, "the 2-bit field (delta + 2) holds -2 .. +1 only"
d8 = delta.astype(np.int8)
packed = ((d8[0::4] + 2).astype(np.uint8) | ((d8[1::4] + 2).astype(np.uint8) << 2) | ((d8[2::4] + 2).astype(np.uint8) << 4)
          | ((d8[3::4] + 2).astype(np.uint8) << 6))
print("packed bytes", packed.nbytes, "zlib", len(zlib.compress(packed.tobytes(), 9)))
np.savez_compressed(out, delta2bit=packed, arch=np.array("gfx950"), note=np.array("v_rsq_f32(max(x, FLT_MIN)) on gfx950 minus float32(1/sqrt(float64(x))) in ulps, "
                                                          "+2, four 2-bit fields per byte, index = (exponent & 1) << 23 | mantissa"))
# scale invariance: rsq(x * 4^k) == rsq(x) * 2^-k exactly, for normal x
rng = np.random.RandomState(0)
rb = rng.randint(1 << 23, (255 << 23), size=1 << 22).astype(np.uint32)          # every normal exponent
xr = rb.view(np.float32)
hwr = ctx.eval_rsqrt(xr)
e = (rb >> 23).astype(np.int64) - 127
par = e & 1
idx = (par << 23 | (rb & 0x7fffff)).astype(np.int64)
base = (ref.view(np.int32).astype(np.int64)[idx] + delta[idx]).astype(np.int32).view(np.float32)
pred = np.ldexp(base, (-(e - par) // 2).astype(np.int32)).astype(np.float32)
bad = hwr.view(np.uint32) != pred.view(np.uint32)
print("scale-invariance mismatches", int(bad.sum()), "of", xr.size)
if bad.any():
    k = np.where(bad)[0][:10]
    print(xr[k], hwr[k], pred[k])
# special inputs (through fs_rsqrt = v_rsq_f32(max(x, FLT_MIN)))
sp = np.array([0.0, 1e-45, 1e-40, 1.17549435e-38, 1.1754942e-38, 3.4e38, np.inf], np.float32)
print("special", list(zip(sp.tolist(), ctx.eval_rsqrt(sp).tolist())))
