// fs_camera.h -- host-side camera / light set-up of the rasteriser (no device code; safe to include anywhere).
// Restates RenderScene's set-up (reference PyFlex/bindings/main.cpp:1411-1438) and the matrix helpers it uses
// (PyFlex/core/maths.h:507-598) in row-major / column-vector form.
#pragma once
#include <cmath>

struct FsRasterFrame {
    float view[16], proj[16], vp[16];  // row-major, column vectors: clip = vp * (p, 1)
    float light_vp[16];
    float cam_pos[3];
    float light_pos[3], light_dir[3];  // light_dir = normalize(target - pos) (shadersGL.cpp:849-850)
    float znear, zfar, fog;
    float tan_half_fov, aspect;
    float inv_rot[9];                  // eye -> world rotation (transpose of the view rotation)
    float plane[4];                    // ground plane (main.cpp:882)
    float col_plane[3], col_shape[3], col_cloth[3];
    float bias_shape;                  // g_shadowBias for DrawMesh, 0 for planes / cloth (shadersGL.cpp:250,1119,1178)
    int W, H;
};

// ---------------------------------------------------------------- host: matrices (core/maths.h:507-598 restated)
static inline void fs_mat_mul(const float *a, const float *b, float *o) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += a[4 * r + k] * b[4 * k + c];
            o[4 * r + c] = s;
        }
}
static inline void fs_mat_rotation(float angle, float ax, float ay, float az, float *m) {
    float l = sqrtf(ax * ax + ay * ay + az * az);
    ax /= l; ay /= l; az /= l;
    float s = sinf(angle), c = cosf(angle);
    float r[16] = {ax * ax + (1.0f - ax * ax) * c, ax * ay * (1.0f - c) - az * s, ax * az * (1.0f - c) + ay * s, 0.0f,
                   ax * ay * (1.0f - c) + az * s, ay * ay + (1.0f - ay * ay) * c, ay * az * (1.0f - c) - ax * s, 0.0f,
                   ax * az * (1.0f - c) - ay * s, ay * az * (1.0f - c) + ax * s, az * az + (1.0f - az * az) * c, 0.0f,
                   0.0f, 0.0f, 0.0f, 1.0f};
    for (int i = 0; i < 16; ++i) m[i] = r[i];
}
static inline void fs_mat_projection(float fov_deg, float aspect, float zn, float zf, float *m) {
    const float kPi = 3.141592653589f;
    float f = 1.0f / tanf((fov_deg * 0.5f) * (kPi / 180.0f));
    float zd = zn - zf;
    float r[16] = {f / aspect, 0, 0, 0, 0, f, 0, 0, 0, 0, (zf + zn) / zd, (2.0f * zn * zf) / zd, 0, 0, -1.0f, 0};
    for (int i = 0; i < 16; ++i) m[i] = r[i];
}
static inline void fs_mat_lookat(const float *eye, const float *target, float *m) {
    float fx = -(target[0] - eye[0]), fy = -(target[1] - eye[1]), fz = -(target[2] - eye[2]);
    float l = sqrtf(fx * fx + fy * fy + fz * fz);
    fx /= l; fy /= l; fz /= l;
    // left = normalize(cross(up, forward)), up = (0,1,0)
    float lx = 1.0f * fz - 0.0f * fy, ly = 0.0f * fx - 0.0f * fz, lz = 0.0f * fy - 1.0f * fx;
    l = sqrtf(lx * lx + ly * ly + lz * lz);
    lx /= l; ly /= l; lz /= l;
    float ux = fy * lz - fz * ly, uy = fz * lx - fx * lz, uz = fx * ly - fy * lx;
    // inverse of the affine [left up forward eye]
    float r[16] = {lx, ly, lz, -(lx * eye[0] + ly * eye[1] + lz * eye[2]),
                   ux, uy, uz, -(ux * eye[0] + uy * eye[1] + uz * eye[2]),
                   fx, fy, fz, -(fx * eye[0] + fy * eye[1] + fz * eye[2]),
                   0, 0, 0, 1};
    for (int i = 0; i < 16; ++i) m[i] = r[i];
}

// RenderScene camera + light setup (main.cpp:1411-1438); scene bounds are the Init-time ones (quirk: main.cpp:875-879)
static inline void fs_raster_setup(FsRasterFrame &fr, const float *cam_pos, const float *cam_angle, int W, int H,
                                   const float *scene_lower, const float *scene_upper) {
    const float kPi = 3.141592653589f;
    const float fov = kPi * 39.5978f / 180.0f;  // main.cpp:474
    fr.W = W; fr.H = H;
    fr.znear = 0.01f; fr.zfar = 3.0f; fr.fog = 0.005f;  // main.cpp:741-742,735
    fr.aspect = float(W) / float(H);
    const float fov_deg = fov * (180.0f / kPi);
    fs_mat_projection(fov_deg, fr.aspect, fr.znear, fr.zfar, fr.proj);
    fr.tan_half_fov = tanf((fov_deg * 0.5f) * (kPi / 180.0f));
    float r1[16], r2[16], tr[16], tmp[16];
    fs_mat_rotation(-cam_angle[0], 0.0f, 1.0f, 0.0f, r1);
    fs_mat_rotation(-cam_angle[1], cosf(-cam_angle[0]), 0.0f, sinf(-cam_angle[0]), r2);
    for (int i = 0; i < 16; ++i) tr[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    tr[3] = -cam_pos[0]; tr[7] = -cam_pos[1]; tr[11] = -cam_pos[2];
    fs_mat_mul(r1, r2, tmp);
    fs_mat_mul(tmp, tr, fr.view);
    fs_mat_mul(fr.proj, fr.view, fr.vp);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) fr.inv_rot[3 * r + c] = fr.view[4 * c + r];
    for (int k = 0; k < 3; ++k) fr.cam_pos[k] = cam_pos[k];
    // light
    float lo[3], up[3];
    const float lo_min[3] = {-2.0f, 0.0f, -2.0f}, up_max[3] = {2.0f, 2.0f, 2.0f};
    for (int k = 0; k < 3; ++k) {
        lo[k] = scene_lower[k] < lo_min[k] ? scene_lower[k] : lo_min[k];
        up[k] = scene_upper[k] > up_max[k] ? scene_upper[k] : up_max[k];
    }
    float ext[3], cen[3];
    for (int k = 0; k < 3; ++k) { ext[k] = up[k] - lo[k]; cen[k] = 0.5f * (up[k] + lo[k]); }
    float ld[3] = {5.0f, 15.0f, 7.5f};
    float l = sqrtf(ld[0] * ld[0] + ld[1] * ld[1] + ld[2] * ld[2]);
    for (int k = 0; k < 3; ++k) ld[k] /= l;
    const float ext_len = sqrtf(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2]);
    for (int k = 0; k < 3; ++k) fr.light_pos[k] = cen[k] + ld[k] * ext_len * 10.0f;  // g_lightDistance = 10
    float d1[3] = {up[0] - cen[0], up[1] - cen[1], up[2] - cen[2]};
    float d2[3] = {fr.light_pos[0] - cen[0], fr.light_pos[1] - cen[1], fr.light_pos[2] - cen[2]};
    float light_fov = 2.0f * atanf(sqrtf(d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2]) /
                                   sqrtf(d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2]));
    const float fmin = 25.0f * (kPi / 180.0f), fmax = 65.0f * (kPi / 180.0f);
    light_fov = light_fov < fmin ? fmin : (light_fov > fmax ? fmax : light_fov);
    float lp[16], lv[16];
    fs_mat_projection(light_fov * (180.0f / kPi), 1.0f, 1.0f, 1000.0f, lp);
    fs_mat_lookat(fr.light_pos, cen, lv);
    fs_mat_mul(lp, lv, fr.light_vp);
    float tl[3] = {cen[0] - fr.light_pos[0], cen[1] - fr.light_pos[1], cen[2] - fr.light_pos[2]};
    l = sqrtf(tl[0] * tl[0] + tl[1] * tl[1] + tl[2] * tl[2]);
    for (int k = 0; k < 3; ++k) fr.light_dir[k] = tl[k] / l;
    fr.plane[0] = 0.0f; fr.plane[1] = 1.0f; fr.plane[2] = 0.0f; fr.plane[3] = 0.0f;
    for (int k = 0; k < 3; ++k) { fr.col_plane[k] = 0.001f; fr.col_shape[k] = 0.9f; }  // shader.cpp:221, main.cpp:502
    fr.col_cloth[0] = 0.612f * 1.5f; fr.col_cloth[1] = 0.194f * 1.5f; fr.col_cloth[2] = 0.394f * 1.5f;  // main.cpp:198
    fr.bias_shape = 0.05f;
}

