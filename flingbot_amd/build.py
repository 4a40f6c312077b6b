"""In-tree build of the HIP extension(s) for gfx950.

    python -m flingbot_amd.build          # libflingsim.so (+ the pyflex pybind11 module)

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract: it makes
the kernels' fp32 results independent of FMA fusion so they match the CPU oracle bit for bit.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libflingsim.so")
ARCH = "gfx950"

LIB_SOURCES = ["fs_capi.hip", "fs_solver.hip", "fs_render.hip", "fs_picker.hip", "fs_loops.hip", "fs_image.hip", "fs_action.hip", "fs_valuenet.hip", "fs_observe.hip", "fs_hostapi.hip", "fs_scene.cpp", "fs_tenants.cpp"]
# -amdgpu-kernarg-preload-count: the first 16 kernel-argument dwords arrive in SGPRs with the wave instead of behind an
# s_load -- one scalar round trip less in front of every kernel, which the streaming back-end's ~130 dependent launches per
# frame feel (measured: 64x64 cloths x 8 / 64 episodes 1.04 / 1.424 -> 0.98 / 1.392 ms per step)
HIP_FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", f"--offload-arch={ARCH}", "-Wall",
             "-Wno-unused-function", "-Wno-unused-result", "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _all_sources():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(ROOT, "include", "flingsim.h"))
    deps.append(os.path.abspath(__file__))  # the compile flags live here
    return deps


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def build_lib(force=False, verbose=False):
    """Compile libflingsim.so (HIP kernels + C-ABI)."""
    if not force and not _newer(LIB, _all_sources()):
        return LIB
    cmd = [hipcc()] + HIP_FLAGS + ["-shared", "-o", LIB] + [os.path.join(CSRC, s) for s in LIB_SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def pyflex_module_path():
    suffix = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    return os.path.join(HERE, "pyflex_native", "pyflex" + suffix)


def build_pyflex(force=False, verbose=False):
    """Compile the pybind11 module named `pyflex` (drop-in for PyFlex/bindings/pyflex.cpp) against libflingsim."""
    import pybind11

    src = os.path.join(CSRC, "pyflex_module.cpp")
    if not os.path.exists(src):
        return None
    out = pyflex_module_path()
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not force and not _newer(out, [src, os.path.join(ROOT, "include", "flingsim.h")]):
        return out
    build_lib(force=False, verbose=verbose)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-o", out, src,
           "-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"], "-I", os.path.join(ROOT, "include"),
           "-L", HERE, "-lflingsim", "-Wl,-rpath,$ORIGIN/.."]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_all(force=False, verbose=False):
    lib = build_lib(force=force, verbose=verbose)
    mod = build_pyflex(force=force, verbose=verbose)
    return lib, mod


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
