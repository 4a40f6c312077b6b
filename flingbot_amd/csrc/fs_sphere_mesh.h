// fs_sphere_mesh.h -- the triangle mesh of a kinematic sphere shape, shared by the rasteriser's kernel
// (fs_raster_kernels.h) and the host-only entry point fs_host_sphere_mesh (fs_hostapi.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#include "fs_types.h"

#define FS_SPHERE_SLICES 20
#define FS_SPHERE_SEGMENTS 20
#define FS_SPHERE_VERTS ((FS_SPHERE_SLICES + 1) * (FS_SPHERE_SEGMENTS + 1))
#define FS_SPHERE_TRIS (FS_SPHERE_SLICES * FS_SPHERE_SEGMENTS * 2)

// Picker sphere meshes the way the reference draws them (main.cpp:1739-1751): CreateSphere(20, 20, radius)
// (core/mesh.cpp:858-902: vertex (i, j) = direction(theta_i, phi_j) * radius, normal = direction) moved by
// Translation(PREVIOUS position) * Rotation(PREVIOUS quaternion) (Mesh::Transform, core/mesh.cpp:650-657).  FlingBot adds
// its pickers with the quaternion [1, 0, 0, 0] (flex_utils.py:82-83), half a turn about x, so vertex 0 is the SOUTH pole.
// The reference evaluates sinf / cosf of the 21 + 21 angles with the host's libm; so does the host here (FsSphereTrig)
// and the kernel only multiplies -- the mesh is then the reference's bit for bit (tests/golden/sphere_golden.json,
// recorded from the reference's own mesh.cpp).
struct FsSphereTrig {
    float sin_t[FS_SPHERE_SLICES + 1], cos_t[FS_SPHERE_SLICES + 1], sin_p[FS_SPHERE_SEGMENTS + 1], cos_p[FS_SPHERE_SEGMENTS + 1];
};
struct FsSphereRot { float a[FS_MAX_SHAPES][9]; };  // a[q][3 c + r]: image of unit axis c under the shape's previous rotation

inline void fs_sphere_trig(FsSphereTrig &t) {
    const float kPi = 3.141592653589f;  // core/maths.h
    const float d_theta = kPi / FS_SPHERE_SLICES, d_phi = (2.0f * kPi) / FS_SPHERE_SEGMENTS;
    for (int i = 0; i <= FS_SPHERE_SLICES; ++i) { t.sin_t[i] = sinf(d_theta * i); t.cos_t[i] = cosf(d_theta * i); }
    for (int j = 0; j <= FS_SPHERE_SEGMENTS; ++j) { t.sin_p[j] = sinf(d_phi * j); t.cos_p[j] = cosf(d_phi * j); }
}
// columns of RotationMatrix(Quat) (maths.h:555-566) = Rotate(q, axis) (quat.h:162-165), terms summed left to right
inline void fs_quat_axes(const float *q, float *a) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float diag = 2.0f * w * w - 1.0f;
    const float e[3][3] = {{1.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 1.0f}};
    for (int c = 0; c < 3; ++c) {
        const float *u = e[c];
        const float cr[3] = {y * u[2] - z * u[1], z * u[0] - x * u[2], x * u[1] - y * u[0]};
        const float dot = x * u[0] + y * u[1] + z * u[2];
        a[3 * c + 0] = u[0] * diag + cr[0] * w * 2.0f + x * dot * 2.0f;
        a[3 * c + 1] = u[1] * diag + cr[1] * w * 2.0f + y * dot * 2.0f;
        a[3 * c + 2] = u[2] * diag + cr[2] * w * 2.0f + z * dot * 2.0f;
    }
}

// vertex v = i * 21 + j of a sphere mesh: Matrix44 * Point3 / * Vec3 (mat44.h:181-201) evaluated left to right (the
// library is built with -ffp-contract=off, so the host and the device round alike)
__host__ __device__ inline void fs_sphere_vertex(const FsSphereTrig &trig, const float *a, float radius, float tx, float ty,
                                                 float tz, int v, FsVec4 &vert, FsVec4 &nrm) {
    const int i = v / (FS_SPHERE_SEGMENTS + 1), j = v % (FS_SPHERE_SEGMENTS + 1);
    const float x = trig.sin_t[i] * trig.cos_p[j], y = trig.cos_t[i], z = trig.sin_t[i] * trig.sin_p[j];
    const float px = x * radius, py = y * radius, pz = z * radius;
    vert = FsVec4{px * a[0] + py * a[3] + pz * a[6] + tx, px * a[1] + py * a[4] + pz * a[7] + ty,
                  px * a[2] + py * a[5] + pz * a[8] + tz, 1.0f};
    nrm = FsVec4{x * a[0] + y * a[3] + z * a[6], x * a[1] + y * a[4] + z * a[7], x * a[2] + y * a[5] + z * a[8], 0.0f};
}

__host__ __device__ inline void fs_sphere_tri(int t, int &a, int &b, int &c) {
    // quad (i, j), i in 1..slices, j in 1..segments; tris (b,a,d) and (b,d,c)
    const int q = t / FS_SPHERE_TRIS, r = t % FS_SPHERE_TRIS;
    const int quad = r >> 1, half = r & 1;
    const int i = quad / FS_SPHERE_SEGMENTS + 1, j = quad % FS_SPHERE_SEGMENTS + 1;
    const int row = FS_SPHERE_SEGMENTS + 1, base = q * FS_SPHERE_VERTS;
    const int va = i * row + j, vb = (i - 1) * row + j, vc = (i - 1) * row + j - 1, vd = i * row + j - 1;
    if (half == 0) { a = base + vb; b = base + va; c = base + vd; }
    else { a = base + vb; b = base + vd; c = base + vc; }
}
