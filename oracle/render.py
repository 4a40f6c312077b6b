"""ctypes front-end of oracle/raster_oracle.c (TEST INFRASTRUCTURE).  The camera / light matrices are inputs: tests pass
the set-up that is pinned against the reference's core/maths.h (tests/golden/camera_golden.json)."""
import ctypes as C

import numpy as np

from .flex import _load, _f, _i, _fp, _ip


def _spheres11(shape_states, radii):
    """pyflex shape states float[14 * S] (pos3, prevPos3, quat4, prevQuat4) + radii -> the oracle's float[11 * S]:
    current xyz, previous xyz, radius, PREVIOUS quaternion (the reference draws shapes at their previous transform,
    main.cpp:1737-1740)."""
    st = np.asarray(shape_states, np.float32).reshape(-1, 14)
    sph = np.zeros((st.shape[0], 11), np.float32)
    if st.shape[0]:
        sph[:, 0:3], sph[:, 3:6], sph[:, 6], sph[:, 7:11] = st[:, 0:3], st[:, 3:6], np.asarray(radii, np.float32), st[:, 10:14]
    return np.ascontiguousarray(sph.ravel())


def sphere_mesh(shape_states, radii):
    """(verts float32[441 S, 4], normals float32[441 S, 4], tris int32[800 S, 3]) of the picker meshes as the reference
    draws them (core/mesh.cpp:858-902 + main.cpp:1739-1751)."""
    lib = _load()
    lib.orc_sphere_mesh.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                    C.POINTER(C.c_int)]
    lib.orc_sphere_mesh.restype = None
    sph = _spheres11(shape_states, radii)
    ns = sph.size // 11
    verts, nrms = np.empty((441 * ns, 4), np.float32), np.empty((441 * ns, 4), np.float32)
    tris = np.empty((800 * ns, 3), np.int32)
    if ns:
        lib.orc_sphere_mesh(_fp(sph), ns, _fp(verts), _fp(nrms), _ip(tris))
    return verts, nrms, tris


def render(mats54, cam_pos, width, height, positions, normals, faces, shape_states=(), radii=()):
    """mats54: view(16) proj(16) lightTransform(16) lightPos(3) lightDir(3), row-major.  shape_states: float[14*S] in
    the pyflex layout (pos3, prevPos3, ...); radii: float[S].  Returns (rgba uint8[H*W*4] bottom-up, depth float32[H*W])."""
    lib = _load()
    lib.orc_render.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_float),
                               C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_float), C.c_int,
                               C.POINTER(C.c_ubyte), C.POINTER(C.c_float)]
    m, cp, p, nr, f = _f(mats54), _f(cam_pos), _f(positions), _f(normals), _i(faces)
    st = np.asarray(shape_states, np.float32).reshape(-1, 14)
    sph = _spheres11(st, radii)
    rgba = np.empty(width * height * 4, np.uint8)
    depth = np.empty(width * height, np.float32)
    rc = lib.orc_render(_fp(m), _fp(cp), width, height, _fp(p), _fp(nr), p.size // 4, _ip(f), f.size // 3,
                        _fp(sph) if sph.size else None, st.shape[0], rgba.ctypes.data_as(C.POINTER(C.c_ubyte)), _fp(depth))
    assert rc == 0
    return rgba, depth
