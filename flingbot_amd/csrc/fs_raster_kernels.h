// fs_raster_kernels.h -- HIP software rasteriser standing in for the reference's OpenGL path
// (pyflex_render PyFlex/bindings/pyflex.cpp:924-1133, RenderScene main.cpp:1339-1582, GLSL shadersGL.cpp:692-839).
//
// Pipeline per frame: sphere meshes for the pickers (core/mesh.cpp:858-902, drawn at their PREVIOUS transform,
// main.cpp:1737-1751) -> 2048^2 shadow depth from the light (polygon offset 8,8; shadersGL.cpp:1002-1004) -> camera
// depth + primitive id with one 64-bit atomicMin per covered pixel -> per-pixel shading (Lambert x PCF shadow x spot
// attenuation + ambient, fog, gamma; shadersGL.cpp:795-839) -> RGBA8 + linearised depth (pyflex.cpp:1046-1054).
// Rules fixed here (GL leaves them to the implementation): 8 sub-pixel bits, integer edge functions with a top-left
// fill rule, 24-bit depth with round-to-nearest, pixel centres at +0.5.  The ground plane is shaded analytically.
// Cloth triangles are ~3 px wide at 720^2, so one thread per triangle walking its bounding box is the right shape.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#include "fs_types.h"

#define FS_SHADOW_RES 2048
#define FS_DEPTH_MAX 16777215.0  // 2^24 - 1

#include "fs_camera.h"
#include "fs_sphere_mesh.h"

// ---------------------------------------------------------------- device
struct FsClipVert { float x, y, z, w; };

__host__ __device__ inline FsClipVert fs_xform(const float *m, float x, float y, float z) {
    FsClipVert c;
    c.x = m[0] * x + m[1] * y + m[2] * z + m[3];
    c.y = m[4] * x + m[5] * y + m[6] * z + m[7];
    c.z = m[8] * x + m[9] * y + m[10] * z + m[11];
    c.w = m[12] * x + m[13] * y + m[14] * z + m[15];
    return c;
}

__global__ void fs_k_sphere_mesh(const FsShapesDev *sh, const FsSphereTrig trig, const FsSphereRot rot, FsVec4 *verts,
                                 FsVec4 *nrms) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= sh->count * FS_SPHERE_VERTS) return;
    const int q = g / FS_SPHERE_VERTS;
    fs_sphere_vertex(trig, rot.a[q], sh->pos[q].w, sh->prev[q].x, sh->prev[q].y, sh->prev[q].z, g % FS_SPHERE_VERTS, verts[g],
                     nrms[g]);
}

struct FsSetupTri {
    long long x0, y0, x1, y1, x2, y2;  // 24.8 fixed-point window coords
    long long area;                    // twice the signed area (x256^2), > 0 after orientation fix
    float d0, d1, d2;                  // window depth in [0,1]
    float w0, w1, w2;                  // clip w
    int minx, maxx, miny, maxy;
    bool front, valid;
};

__host__ __device__ inline long long fs_edge(long long ax, long long ay, long long bx, long long by, long long px,
                                             long long py) {
    return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}
// top-left rule on a CCW (y-up) triangle: an edge owns its boundary pixels when it is a left edge (going down) or a
// top edge (horizontal, pointing to -x)
__host__ __device__ inline bool fs_owns(long long ax, long long ay, long long bx, long long by) {
    return (by < ay) || (by == ay && bx < ax);
}

__host__ __device__ inline FsSetupTri fs_setup_tri(const float *m, int W, int H, const FsVec4 p0, const FsVec4 p1,
                                                   const FsVec4 p2) {
    FsSetupTri s;
    s.valid = false;
    FsClipVert c0 = fs_xform(m, p0.x, p0.y, p0.z), c1 = fs_xform(m, p1.x, p1.y, p1.z), c2 = fs_xform(m, p2.x, p2.y, p2.z);
    if (!(c0.w > 1e-6f) || !(c1.w > 1e-6f) || !(c2.w > 1e-6f)) return s;  // no near-plane clipping
    const float i0 = 1.0f / c0.w, i1 = 1.0f / c1.w, i2 = 1.0f / c2.w;
    const float fx0 = (c0.x * i0 * 0.5f + 0.5f) * (float)W, fy0 = (c0.y * i0 * 0.5f + 0.5f) * (float)H;
    const float fx1 = (c1.x * i1 * 0.5f + 0.5f) * (float)W, fy1 = (c1.y * i1 * 0.5f + 0.5f) * (float)H;
    const float fx2 = (c2.x * i2 * 0.5f + 0.5f) * (float)W, fy2 = (c2.y * i2 * 0.5f + 0.5f) * (float)H;
    const float lim = 1.0e6f;
    if (!(fabsf(fx0) < lim && fabsf(fy0) < lim && fabsf(fx1) < lim && fabsf(fy1) < lim && fabsf(fx2) < lim && fabsf(fy2) < lim))
        return s;
    s.x0 = (long long)rintf(fx0 * 256.0f); s.y0 = (long long)rintf(fy0 * 256.0f);
    s.x1 = (long long)rintf(fx1 * 256.0f); s.y1 = (long long)rintf(fy1 * 256.0f);
    s.x2 = (long long)rintf(fx2 * 256.0f); s.y2 = (long long)rintf(fy2 * 256.0f);
    s.d0 = c0.z * i0 * 0.5f + 0.5f; s.d1 = c1.z * i1 * 0.5f + 0.5f; s.d2 = c2.z * i2 * 0.5f + 0.5f;
    s.w0 = c0.w; s.w1 = c1.w; s.w2 = c2.w;
    long long area = fs_edge(s.x0, s.y0, s.x1, s.y1, s.x2, s.y2);
    if (area == 0) return s;
    s.front = area > 0;
    if (area < 0) {  // make CCW by swapping 1 <-> 2
        long long tx = s.x1, ty = s.y1; s.x1 = s.x2; s.y1 = s.y2; s.x2 = tx; s.y2 = ty;
        float td = s.d1; s.d1 = s.d2; s.d2 = td;
        float tw = s.w1; s.w1 = s.w2; s.w2 = tw;
        area = -area;
    }
    s.area = area;
    long long mnx = s.x0 < s.x1 ? s.x0 : s.x1; mnx = mnx < s.x2 ? mnx : s.x2;
    long long mxx = s.x0 > s.x1 ? s.x0 : s.x1; mxx = mxx > s.x2 ? mxx : s.x2;
    long long mny = s.y0 < s.y1 ? s.y0 : s.y1; mny = mny < s.y2 ? mny : s.y2;
    long long mxy = s.y0 > s.y1 ? s.y0 : s.y1; mxy = mxy > s.y2 ? mxy : s.y2;
    // pixel centres at (px*256 + 128)
    long long a = (mnx - 128 + 255) >> 8, b = (mxx - 128) >> 8, c = (mny - 128 + 255) >> 8, d = (mxy - 128) >> 8;
    s.minx = (int)(a < 0 ? 0 : a); s.maxx = (int)(b > W - 1 ? W - 1 : b);
    s.miny = (int)(c < 0 ? 0 : c); s.maxy = (int)(d > H - 1 ? H - 1 : d);
    s.valid = s.minx <= s.maxx && s.miny <= s.maxy;
    return s;
}

// coverage + barycentric weights of pixel (px, py); returns false if outside
__host__ __device__ inline bool fs_cover(const FsSetupTri &s, int px, int py, double &l0, double &l1, double &l2) {
    const long long cx = ((long long)px << 8) + 128, cy = ((long long)py << 8) + 128;
    const long long e0 = fs_edge(s.x1, s.y1, s.x2, s.y2, cx, cy);  // weight of vertex 0
    const long long e1 = fs_edge(s.x2, s.y2, s.x0, s.y0, cx, cy);
    const long long e2 = fs_edge(s.x0, s.y0, s.x1, s.y1, cx, cy);
    if (e0 < 0 || e1 < 0 || e2 < 0) return false;
    if (e0 == 0 && !fs_owns(s.x1, s.y1, s.x2, s.y2)) return false;
    if (e1 == 0 && !fs_owns(s.x2, s.y2, s.x0, s.y0)) return false;
    if (e2 == 0 && !fs_owns(s.x0, s.y0, s.x1, s.y1)) return false;
    const double inv = 1.0 / (double)s.area;
    l0 = (double)e0 * inv; l1 = (double)e1 * inv; l2 = (double)e2 * inv;
    return true;
}

__host__ __device__ inline unsigned int fs_quant24(double d) {
    double q = d * FS_DEPTH_MAX + 0.5;
    if (q < 0.0) q = 0.0;
    if (q > FS_DEPTH_MAX) q = FS_DEPTH_MAX;
    return (unsigned int)q;
}

// primitive ids: 0 = ground plane, 1 .. n_sph_tris = sphere triangles, then cloth triangles (draw order)
__global__ __launch_bounds__(64) void fs_k_raster_camera(FsRasterFrame fr, const FsVec4 *pos, const int *tris, int n_cloth,
                                                         const FsVec4 *sph, int n_sph_tris, unsigned long long *zbuf) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n_cloth + n_sph_tris) return;
    FsVec4 p0, p1, p2;
    const bool is_sphere = t < n_sph_tris;
    if (is_sphere) {
        int a, b, c;
        fs_sphere_tri(t, a, b, c);
        p0 = sph[a]; p1 = sph[b]; p2 = sph[c];
    } else {
        const int k = t - n_sph_tris;
        p0 = pos[tris[3 * k]]; p1 = pos[tris[3 * k + 1]]; p2 = pos[tris[3 * k + 2]];
    }
    const FsSetupTri s = fs_setup_tri(fr.vp, fr.W, fr.H, p0, p1, p2);
    if (!s.valid) return;
    if (is_sphere && !s.front) return;  // back-face culling is on for meshes, off for cloth (shadersGL.cpp:1169-1170)
    const unsigned long long id = (unsigned long long)(t + 1);
    for (int py = s.miny; py <= s.maxy; ++py)
        for (int px = s.minx; px <= s.maxx; ++px) {
            double l0, l1, l2;
            if (!fs_cover(s, px, py, l0, l1, l2)) continue;
            const double d = l0 * (double)s.d0 + l1 * (double)s.d1 + l2 * (double)s.d2;
            if (d < 0.0 || d > 1.0) continue;
            const unsigned long long key = ((unsigned long long)fs_quant24(d) << 32) | id;
            atomicMin(&zbuf[(size_t)py * fr.W + px], key);
        }
}

__global__ __launch_bounds__(64) void fs_k_raster_shadow(FsRasterFrame fr, const FsVec4 *pos, const int *tris, int n_cloth,
                                                         const FsVec4 *sph, int n_sph_tris, unsigned int *shadow) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n_cloth + n_sph_tris) return;
    FsVec4 p0, p1, p2;
    if (t < n_sph_tris) {
        int a, b, c;
        fs_sphere_tri(t, a, b, c);
        p0 = sph[a]; p1 = sph[b]; p2 = sph[c];
    } else {
        const int k = t - n_sph_tris;
        p0 = pos[tris[3 * k]]; p1 = pos[tris[3 * k + 1]]; p2 = pos[tris[3 * k + 2]];
    }
    const FsSetupTri s = fs_setup_tri(fr.light_vp, FS_SHADOW_RES, FS_SHADOW_RES, p0, p1, p2);
    if (!s.valid) return;
    // glPolygonOffset(8, 8): o = 8 * max(|dz/dx|, |dz/dy|) + 8 * 2^-24, slopes from the depth plane in window space
    const double ax = (double)(s.x1 - s.x0) / 256.0, ay = (double)(s.y1 - s.y0) / 256.0;
    const double bx = (double)(s.x2 - s.x0) / 256.0, by = (double)(s.y2 - s.y0) / 256.0;
    const double az = (double)s.d1 - (double)s.d0, bz = (double)s.d2 - (double)s.d0;
    const double det = ax * by - ay * bx;
    double dzdx = 0.0, dzdy = 0.0;
    if (det != 0.0) { dzdx = (az * by - bz * ay) / det; dzdy = (bz * ax - az * bx) / det; }
    const double slope = fmax(fabs(dzdx), fabs(dzdy));
    const double offset = 8.0 * slope + 8.0 / 16777216.0;
    for (int py = s.miny; py <= s.maxy; ++py)
        for (int px = s.minx; px <= s.maxx; ++px) {
            double l0, l1, l2;
            if (!fs_cover(s, px, py, l0, l1, l2)) continue;
            const double d = l0 * (double)s.d0 + l1 * (double)s.d1 + l2 * (double)s.d2 + offset;
            if (d < 0.0) continue;
            atomicMin(&shadow[(size_t)py * FS_SHADOW_RES + px], fs_quant24(d > 1.0 ? 1.0 : d));
        }
}

// one bilinear PCF tap (GL_LINEAR on a GL_COMPARE_R_TO_TEXTURE / GL_LEQUAL depth texture, shadersGL.cpp:969-976)
__device__ inline float fs_shadow_tap(const unsigned int *shadow, float u, float v, float ref) {
    const float x = u * (float)FS_SHADOW_RES - 0.5f, y = v * (float)FS_SHADOW_RES - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float ax = x - fx0, ay = y - fy0;
    float acc = 0.0f;
    for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
            int ix = (int)fx0 + dx, iy = (int)fy0 + dy;
            ix = ix < 0 ? 0 : (ix > FS_SHADOW_RES - 1 ? FS_SHADOW_RES - 1 : ix);  // clamp-to-edge
            iy = iy < 0 ? 0 : (iy > FS_SHADOW_RES - 1 ? FS_SHADOW_RES - 1 : iy);
            const unsigned int q = shadow[(size_t)iy * FS_SHADOW_RES + ix];
            const float texel = q == 0xffffffffu ? 1.0f : (float)((double)q / FS_DEPTH_MAX);
            const float lit = ref <= texel ? 1.0f : 0.0f;
            acc += lit * (dx ? ax : 1.0f - ax) * (dy ? ay : 1.0f - ay);
        }
    return acc;
}

__device__ inline void fs_shade(const FsRasterFrame &fr, const unsigned int *shadow, float px, float py, float pz, float nx,
                                float ny, float nz, const float *color, float bias, float &r, float &g, float &b) {
    const float taps[12][2] = {{-0.326212f, -0.40581f}, {-0.840144f, -0.07358f}, {-0.695914f, 0.457137f},
                               {-0.203345f, 0.620716f}, {0.96234f, -0.194983f}, {0.473434f, -0.480026f},
                               {0.519456f, 0.767022f}, {0.185461f, -0.893124f}, {0.507431f, 0.064425f},
                               {0.89642f, 0.412458f}, {-0.32194f, -0.932615f}, {-0.791559f, -0.59771f}};
    const FsClipVert lc = fs_xform(fr.light_vp, px + nx * bias, py + ny * bias, pz + nz * bias);
    const float lx = lc.x / lc.w, ly = lc.y / lc.w, lz = lc.z / lc.w;
    const float u = lx * 0.5f + 0.5f, v = ly * 0.5f + 0.5f, wz = lz * 0.5f + 0.5f;
    float sh = 1.0f;
    if (!(u < 0.0f || u > 1.0f || v < 0.0f || v > 1.0f)) {
        float s = 0.0f;
        for (int k = 0; k < 12; ++k) s += fs_shadow_tap(shadow, u + taps[k][0] * 0.002f, v + taps[k][1] * 0.002f, wz);
        sh = s / 12.0f;
    }
    sh = fmaxf(sh, 0.5f);
    // attenuation = max(smoothstep(spotMax = 1.0, spotMin = 0.5, r^2), 0.05)
    float tt = (lx * lx + ly * ly - 1.0f) / (0.5f - 1.0f);
    tt = tt < 0.0f ? 0.0f : (tt > 1.0f ? 1.0f : tt);
    const float att = fmaxf(tt * tt * (3.0f - 2.0f * tt), 0.05f);
    const float ndl = -(fr.light_dir[0] * nx + fr.light_dir[1] * ny + fr.light_dir[2] * nz);
    const float diff = fmaxf(0.0f, ndl * sh) * att;
    const float mixv = ndl * 0.5f + 0.5f;
    const float light[3] = {0.03f * 1.5f, 0.025f * 1.5f, 0.025f * 1.5f}, dark[3] = {0.025f, 0.025f, 0.03f};
    const FsClipVert ev = fs_xform(fr.view, px, py, pz);
    const float fogf = expf(ev.z * fr.fog);
    float out[3];
    for (int k = 0; k < 3; ++k) {
        const float amb = 4.0f * color[k] * (dark[k] * (1.0f - mixv) + light[k] * mixv) * att;
        const float lit = color[k] * diff + amb;
        const float fogged = 0.0f * (1.0f - fogf) + lit * fogf;  // fog colour = clear colour = black
        out[k] = powf(fmaxf(fogged, 0.0f), 1.0f / 2.2f);
    }
    r = out[0]; g = out[1]; b = out[2];
}

__device__ inline unsigned char fs_to_u8(float c) {
    c = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
    return (unsigned char)(c * 255.0f + 0.5f);
}

__global__ __launch_bounds__(256) void fs_k_shade(FsRasterFrame fr, const FsVec4 *pos, const FsVec4 *nrm, const int *tris,
                                                  int n_cloth, const FsVec4 *sph, const FsVec4 *sph_n, int n_sph_tris,
                                                  const unsigned long long *zbuf, const unsigned int *shadow,
                                                  unsigned char *rgba, float *depth) {
    const int px = blockIdx.x * 16 + threadIdx.x, py = blockIdx.y * 16 + threadIdx.y;
    if (px >= fr.W || py >= fr.H) return;
    const size_t pix = (size_t)py * fr.W + px;
    unsigned long long key = zbuf[pix];
    // ground plane, analytic: ray through the pixel centre
    const float xn = (((float)px + 0.5f) / (float)fr.W) * 2.0f - 1.0f, yn = (((float)py + 0.5f) / (float)fr.H) * 2.0f - 1.0f;
    const float ex = xn * fr.tan_half_fov * fr.aspect, ey = yn * fr.tan_half_fov, ez = -1.0f;
    const float dx = fr.inv_rot[0] * ex + fr.inv_rot[1] * ey + fr.inv_rot[2] * ez;
    const float dy = fr.inv_rot[3] * ex + fr.inv_rot[4] * ey + fr.inv_rot[5] * ez;
    const float dz = fr.inv_rot[6] * ex + fr.inv_rot[7] * ey + fr.inv_rot[8] * ez;
    const float denom = fr.plane[0] * dx + fr.plane[1] * dy + fr.plane[2] * dz;
    float hx = 0.0f, hy = 0.0f, hz = 0.0f;
    bool plane_hit = false;
    if (denom < 0.0f) {  // front face of the plane only (cull mode on, main.cpp:1510)
        const float num = -(fr.plane[0] * fr.cam_pos[0] + fr.plane[1] * fr.cam_pos[1] + fr.plane[2] * fr.cam_pos[2] + fr.plane[3]);
        const float tpar = num / denom;
        if (tpar > 0.0f) {
            hx = fr.cam_pos[0] + dx * tpar; hy = fr.cam_pos[1] + dy * tpar; hz = fr.cam_pos[2] + dz * tpar;
            const FsClipVert c = fs_xform(fr.vp, hx, hy, hz);
            const double d = (double)(c.z / c.w) * 0.5 + 0.5;
            if (c.w > 0.0f && d >= 0.0 && d <= 1.0) {
                const unsigned long long pk = (unsigned long long)fs_quant24(d) << 32;
                if (pk < key) { key = pk; plane_hit = true; }
            }
        }
    }
    float r = 0.0f, g = 0.0f, b = 0.0f;  // clear colour (0,0,0) -> pow(0, 1/2.2) = 0
    double dwin = 1.0;                   // cleared depth
    if (key != 0xffffffffffffffffull) {
        dwin = (double)(unsigned int)(key >> 32) / FS_DEPTH_MAX;
        const int id = (int)(key & 0xffffffffu);
        if (plane_hit && id == 0) {
            fs_shade(fr, shadow, hx, hy, hz, fr.plane[0], fr.plane[1], fr.plane[2], fr.col_plane, 0.0f, r, g, b);
        } else {
            const int t = id - 1;
            FsVec4 p0, p1, p2, n0, n1, n2;
            const bool is_sphere = t < n_sph_tris;
            if (is_sphere) {
                int a, bb, c;
                fs_sphere_tri(t, a, bb, c);
                p0 = sph[a]; p1 = sph[bb]; p2 = sph[c];
                n0 = sph_n[a]; n1 = sph_n[bb]; n2 = sph_n[c];
            } else {
                const int k = t - n_sph_tris;
                const int a = tris[3 * k], bb = tris[3 * k + 1], c = tris[3 * k + 2];
                p0 = pos[a]; p1 = pos[bb]; p2 = pos[c];
                n0 = nrm[a]; n1 = nrm[bb]; n2 = nrm[c];
            }
            FsSetupTri s = fs_setup_tri(fr.vp, fr.W, fr.H, p0, p1, p2);
            if (!s.front) {  // setup swapped vertices 1 <-> 2
                FsVec4 tp = p1; p1 = p2; p2 = tp;
                FsVec4 tn = n1; n1 = n2; n2 = tn;
            }
            double l0, l1, l2;
            fs_cover(s, px, py, l0, l1, l2);
            // perspective-correct weights
            double q0 = l0 / (double)s.w0, q1 = l1 / (double)s.w1, q2 = l2 / (double)s.w2;
            const double qs = q0 + q1 + q2;
            const float b0 = (float)(q0 / qs), b1 = (float)(q1 / qs), b2 = (float)(q2 / qs);
            const float wx = b0 * p0.x + b1 * p1.x + b2 * p2.x, wy = b0 * p0.y + b1 * p1.y + b2 * p2.y,
                        wz = b0 * p0.z + b1 * p1.z + b2 * p2.z;
            float nx = b0 * n0.x + b1 * n1.x + b2 * n2.x, ny = b0 * n0.y + b1 * n1.y + b2 * n2.y,
                  nz = b0 * n0.z + b1 * n1.z + b2 * n2.z;
            if (!s.front) { nx = -nx; ny = -ny; nz = -nz; }  // gl_FrontFacing == false: flipped normal, secondary colour
            fs_shade(fr, shadow, wx, wy, wz, nx, ny, nz, is_sphere ? fr.col_shape : fr.col_cloth,
                     is_sphere ? fr.bias_shape : 0.0f, r, g, b);
        }
    }
    rgba[4 * pix + 0] = fs_to_u8(r);
    rgba[4 * pix + 1] = fs_to_u8(g);
    rgba[4 * pix + 2] = fs_to_u8(b);
    rgba[4 * pix + 3] = (key != 0xffffffffffffffffull) ? 255 : 0;  // shader writes alpha 1, clear alpha 0
    // pyflex.cpp:1053 depth linearisation
    const float dw = (float)dwin;
    depth[pix] = 2.0f * fr.zfar * fr.znear / (fr.zfar + fr.znear - (2.0f * dw - 1.0f) * (fr.zfar - fr.znear));
}
