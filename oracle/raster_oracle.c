/*
 * raster_oracle.c -- CPU ORACLE (test infrastructure, NOT product code) of the observation renderer.
 *
 * Scalar restatement of pyflex_render (PyFlex/bindings/pyflex.cpp:924-1133) + RenderScene (main.cpp:1339-1582) + the
 * solid shader (opengl/shadersGL.cpp:692-839) for the cloth scene: picker spheres (core/mesh.cpp:858-902, drawn at their
 * previous position main.cpp:1737-1751), 2048^2 shadow pass with glPolygonOffset(8,8) (shadersGL.cpp:1002-1004), colour
 * pass planes -> shapes -> cloth (main.cpp:1510-1527), RGBA8 + depth linearisation (pyflex.cpp:1046-1054).
 *
 * PARITY UNPINNED at pixel level: the reference needs EGL + a GL driver, which this container lacks.  The rules OpenGL
 * leaves to the implementation are fixed here (8 sub-pixel bits, integer edge functions, top-left fill rule, 24-bit
 * round-to-nearest depth, pixel centres at +0.5); the HIP rasteriser must reproduce this file: depth bit-exact, colour
 * within 1 LSB (expf / powf differ in the last ulp between libm and the device library).
 * Camera / light matrices are passed in by the caller (tests take them from the reference-pinned set-up).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SHADOW_RES 2048
#define SPH_SLICES 20
#define SPH_SEGS 20
#define SPH_VERTS ((SPH_SLICES + 1) * (SPH_SEGS + 1))
#define SPH_TRIS (SPH_SLICES * SPH_SEGS * 2)
#define DEPTH_MAX 16777215.0

typedef struct { float x, y, z, w; } v4;

typedef struct {
    long long x[3], y[3];
    long long area;
    float d[3], w[3];
    int minx, maxx, miny, maxy, front, valid;
    int order[3]; /* original vertex index of slot 0,1,2 */
} tri_setup;

static v4 xform(const float *m, float x, float y, float z) {
    v4 c;
    c.x = m[0] * x + m[1] * y + m[2] * z + m[3];
    c.y = m[4] * x + m[5] * y + m[6] * z + m[7];
    c.z = m[8] * x + m[9] * y + m[10] * z + m[11];
    c.w = m[12] * x + m[13] * y + m[14] * z + m[15];
    return c;
}
static long long edge(long long ax, long long ay, long long bx, long long by, long long px, long long py) {
    return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}
static int owns(long long ax, long long ay, long long bx, long long by) { return (by < ay) || (by == ay && bx < ax); }

static tri_setup setup(const float *m, int W, int H, const v4 *p) {
    tri_setup s;
    memset(&s, 0, sizeof(s));
    float fx[3], fy[3];
    for (int k = 0; k < 3; ++k) {
        v4 c = xform(m, p[k].x, p[k].y, p[k].z);
        if (!(c.w > 1e-6f)) return s;
        float inv = 1.0f / c.w;
        fx[k] = (c.x * inv * 0.5f + 0.5f) * (float)W;
        fy[k] = (c.y * inv * 0.5f + 0.5f) * (float)H;
        if (!(fabsf(fx[k]) < 1.0e6f && fabsf(fy[k]) < 1.0e6f)) return s;
        s.d[k] = c.z * inv * 0.5f + 0.5f;
        s.w[k] = c.w;
        s.order[k] = k;
    }
    for (int k = 0; k < 3; ++k) { s.x[k] = (long long)rintf(fx[k] * 256.0f); s.y[k] = (long long)rintf(fy[k] * 256.0f); }
    long long area = edge(s.x[0], s.y[0], s.x[1], s.y[1], s.x[2], s.y[2]);
    if (area == 0) return s;
    s.front = area > 0;
    if (area < 0) { /* swap 1 <-> 2 to make it counter-clockwise */
        long long t;
        t = s.x[1]; s.x[1] = s.x[2]; s.x[2] = t;
        t = s.y[1]; s.y[1] = s.y[2]; s.y[2] = t;
        float f = s.d[1]; s.d[1] = s.d[2]; s.d[2] = f;
        f = s.w[1]; s.w[1] = s.w[2]; s.w[2] = f;
        s.order[1] = 2; s.order[2] = 1;
        area = -area;
    }
    s.area = area;
    long long mnx = s.x[0], mxx = s.x[0], mny = s.y[0], mxy = s.y[0];
    for (int k = 1; k < 3; ++k) {
        if (s.x[k] < mnx) mnx = s.x[k];
        if (s.x[k] > mxx) mxx = s.x[k];
        if (s.y[k] < mny) mny = s.y[k];
        if (s.y[k] > mxy) mxy = s.y[k];
    }
    long long a = (mnx - 128 + 255) >> 8, b = (mxx - 128) >> 8, c = (mny - 128 + 255) >> 8, d = (mxy - 128) >> 8;
    s.minx = (int)(a < 0 ? 0 : a); s.maxx = (int)(b > W - 1 ? W - 1 : b);
    s.miny = (int)(c < 0 ? 0 : c); s.maxy = (int)(d > H - 1 ? H - 1 : d);
    s.valid = s.minx <= s.maxx && s.miny <= s.maxy;
    return s;
}
static int cover(const tri_setup *s, int px, int py, double *l) {
    long long cx = ((long long)px << 8) + 128, cy = ((long long)py << 8) + 128;
    long long e0 = edge(s->x[1], s->y[1], s->x[2], s->y[2], cx, cy);
    long long e1 = edge(s->x[2], s->y[2], s->x[0], s->y[0], cx, cy);
    long long e2 = edge(s->x[0], s->y[0], s->x[1], s->y[1], cx, cy);
    if (e0 < 0 || e1 < 0 || e2 < 0) return 0;
    if (e0 == 0 && !owns(s->x[1], s->y[1], s->x[2], s->y[2])) return 0;
    if (e1 == 0 && !owns(s->x[2], s->y[2], s->x[0], s->y[0])) return 0;
    if (e2 == 0 && !owns(s->x[0], s->y[0], s->x[1], s->y[1])) return 0;
    double inv = 1.0 / (double)s->area;
    l[0] = (double)e0 * inv; l[1] = (double)e1 * inv; l[2] = (double)e2 * inv;
    return 1;
}
static unsigned quant24(double d) {
    double q = d * DEPTH_MAX + 0.5;
    if (q < 0.0) q = 0.0;
    if (q > DEPTH_MAX) q = DEPTH_MAX;
    return (unsigned)q;
}
static void sphere_tri(int t, int *a, int *b, int *c) {
    int q = t / SPH_TRIS, r = t % SPH_TRIS, quad = r >> 1, half = r & 1;
    int i = quad / SPH_SEGS + 1, j = quad % SPH_SEGS + 1, row = SPH_SEGS + 1, base = q * SPH_VERTS;
    int va = i * row + j, vb = (i - 1) * row + j, vc = (i - 1) * row + j - 1, vd = i * row + j - 1;
    if (half == 0) { *a = base + vb; *b = base + va; *c = base + vd; }
    else { *a = base + vb; *b = base + vd; *c = base + vc; }
}

/*
 * The mesh the reference draws for a kinematic sphere shape (bindings/main.cpp:1739-1751):
 *   CreateSphere(20, 20, radius)                    core/mesh.cpp:858-902  (unit direction * radius, normal = direction)
 *   ->Transform(Translation(prevPos) * Rotation(prevQuat))   core/mesh.cpp:650-657, maths.h:555-573, quat.h:162-165
 * spheres: float[11 * ns] = current xyz, previous xyz, radius, previous rotation quaternion (x, y, z, w).
 * FlingBot adds its pickers with the quaternion [1, 0, 0, 0] (flex_utils.py:82-83) -- half a turn about x -- so the
 * mesh it draws starts at the SOUTH pole.  Pinned against the reference's own mesh.cpp (tests/golden/sphere_golden.json).
 * verts / nrms: float4[441 * ns] (w = 1 / 0); tris (may be NULL): int[3 * 800 * ns], indices into the concatenated meshes.
 */
static void quat_axes(const float *q, float *R /* R[3 * c + r]: image of unit axis c */) {
    /* quat.h:162-165  Rotate(q, x) = x (2 w w - 1) + cross(q.xyz, x) w 2 + q.xyz dot(q.xyz, x) 2, summed left to right */
    const float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    const float s = 2.0f * qw * qw - 1.0f;
    for (int c = 0; c < 3; ++c) {
        const float ex = c == 0 ? 1.0f : 0.0f, ey = c == 1 ? 1.0f : 0.0f, ez = c == 2 ? 1.0f : 0.0f;
        const float cx = qy * ez - qz * ey, cy = qz * ex - qx * ez, cz = qx * ey - qy * ex;
        const float d = qx * ex + qy * ey + qz * ez;
        R[3 * c + 0] = ex * s + cx * qw * 2.0f + qx * d * 2.0f;
        R[3 * c + 1] = ey * s + cy * qw * 2.0f + qy * d * 2.0f;
        R[3 * c + 2] = ez * s + cz * qw * 2.0f + qz * d * 2.0f;
    }
}
void orc_sphere_mesh(const float *spheres, int ns, float *verts, float *nrms, int *tris) {
    const float kPi = 3.141592653589f; /* core/maths.h */
    const float dTheta = kPi / SPH_SLICES, dPhi = (2.0f * kPi) / SPH_SEGS;
    for (int q = 0; q < ns; ++q) {
        const float *S = spheres + 11 * q;
        const float r = S[6];
        float R[9];
        quat_axes(S + 7, R);
        for (int i = 0; i <= SPH_SLICES; ++i)
            for (int j = 0; j <= SPH_SEGS; ++j) {
                const float theta = dTheta * i, phi = dPhi * j;
                const float x = sinf(theta) * cosf(phi), y = cosf(theta), z = sinf(theta) * sinf(phi);
                const float px = x * r, py = y * r, pz = z * r;
                float *V = verts + 4 * (q * SPH_VERTS + i * (SPH_SEGS + 1) + j);
                float *N = nrms + 4 * (q * SPH_VERTS + i * (SPH_SEGS + 1) + j);
                /* mat44.h:181-201: v.x m[0] + v.y m[4] + v.z m[8] (+ m[12]), left to right */
                V[0] = px * R[0] + py * R[3] + pz * R[6] + S[3];
                V[1] = px * R[1] + py * R[4] + pz * R[7] + S[4];
                V[2] = px * R[2] + py * R[5] + pz * R[8] + S[5];
                V[3] = 1.0f;
                N[0] = x * R[0] + y * R[3] + z * R[6];
                N[1] = x * R[1] + y * R[4] + z * R[7];
                N[2] = x * R[2] + y * R[5] + z * R[8];
                N[3] = 0.0f;
            }
    }
    if (tris)
        for (int t = 0; t < ns * SPH_TRIS; ++t) sphere_tri(t, &tris[3 * t], &tris[3 * t + 1], &tris[3 * t + 2]);
}

typedef struct {
    const float *view, *vp, *light_vp, *cam_pos, *light_dir;
    float znear, zfar, fog, tan_half_fov, aspect;
    float inv_rot[9];
    int W, H;
} frame;

static float shadow_tap(const unsigned *sh, float u, float v, float ref) {
    float x = u * (float)SHADOW_RES - 0.5f, y = v * (float)SHADOW_RES - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y), ax = x - fx0, ay = y - fy0, acc = 0.0f;
    for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
            int ix = (int)fx0 + dx, iy = (int)fy0 + dy;
            ix = ix < 0 ? 0 : (ix > SHADOW_RES - 1 ? SHADOW_RES - 1 : ix);
            iy = iy < 0 ? 0 : (iy > SHADOW_RES - 1 ? SHADOW_RES - 1 : iy);
            unsigned q = sh[(size_t)iy * SHADOW_RES + ix];
            float texel = q == 0xffffffffu ? 1.0f : (float)((double)q / DEPTH_MAX);
            float lit = ref <= texel ? 1.0f : 0.0f;
            acc += lit * (dx ? ax : 1.0f - ax) * (dy ? ay : 1.0f - ay);
        }
    return acc;
}

/* shadersGL.cpp:795-839 */
static void shade(const frame *fr, const unsigned *sh, float px, float py, float pz, float nx, float ny, float nz,
                  const float *color, float bias, float *out) {
    static const float taps[12][2] = {{-0.326212f, -0.40581f}, {-0.840144f, -0.07358f}, {-0.695914f, 0.457137f},
                                      {-0.203345f, 0.620716f}, {0.96234f, -0.194983f}, {0.473434f, -0.480026f},
                                      {0.519456f, 0.767022f}, {0.185461f, -0.893124f}, {0.507431f, 0.064425f},
                                      {0.89642f, 0.412458f}, {-0.32194f, -0.932615f}, {-0.791559f, -0.59771f}};
    v4 lc = xform(fr->light_vp, px + nx * bias, py + ny * bias, pz + nz * bias);
    float lx = lc.x / lc.w, ly = lc.y / lc.w, lz = lc.z / lc.w;
    float u = lx * 0.5f + 0.5f, v = ly * 0.5f + 0.5f, wz = lz * 0.5f + 0.5f, shd = 1.0f;
    if (!(u < 0.0f || u > 1.0f || v < 0.0f || v > 1.0f)) {
        float s = 0.0f;
        for (int k = 0; k < 12; ++k) s += shadow_tap(sh, u + taps[k][0] * 0.002f, v + taps[k][1] * 0.002f, wz);
        shd = s / 12.0f;
    }
    shd = fmaxf(shd, 0.5f);
    float tt = (lx * lx + ly * ly - 1.0f) / (0.5f - 1.0f);
    tt = tt < 0.0f ? 0.0f : (tt > 1.0f ? 1.0f : tt);
    float att = fmaxf(tt * tt * (3.0f - 2.0f * tt), 0.05f);
    float ndl = -(fr->light_dir[0] * nx + fr->light_dir[1] * ny + fr->light_dir[2] * nz);
    float diff = fmaxf(0.0f, ndl * shd) * att, mixv = ndl * 0.5f + 0.5f;
    const float light[3] = {0.03f * 1.5f, 0.025f * 1.5f, 0.025f * 1.5f}, dark[3] = {0.025f, 0.025f, 0.03f};
    v4 ev = xform(fr->view, px, py, pz);
    float fogf = expf(ev.z * fr->fog);
    for (int k = 0; k < 3; ++k) {
        float amb = 4.0f * color[k] * (dark[k] * (1.0f - mixv) + light[k] * mixv) * att;
        float lit = color[k] * diff + amb;
        float fogged = 0.0f * (1.0f - fogf) + lit * fogf;
        out[k] = powf(fmaxf(fogged, 0.0f), 1.0f / 2.2f);
    }
}
static unsigned char to_u8(float c) {
    c = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
    return (unsigned char)(c * 255.0f + 0.5f);
}

/*
 * mats: [0:16] view, [16:32] proj, [32:48] light view-proj (row-major, column vectors), [48:51] lightPos, [51:54] lightDir
 * pos/nrm: float4[n]; tris int[3t]; spheres: float[11*ns] = current xyz, previous xyz, radius, previous quaternion xyzw (see orc_sphere_mesh).
 * Outputs rgba[W*H*4] (bottom-up) and linear depth[W*H].
 */
int orc_render(const float *mats, const float *cam_pos, int W, int H, const float *pos, const float *nrm, int n,
               const int *tris, int t, const float *spheres, int ns, unsigned char *rgba, float *depth) {
    (void)n;
    const float *view = mats, *proj = mats + 16, *light_vp = mats + 32, *light_dir = mats + 51;
    float vp[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += proj[4 * r + k] * view[4 * k + c];
            vp[4 * r + c] = s;
        }
    frame fr;
    fr.view = view; fr.vp = vp; fr.light_vp = light_vp; fr.cam_pos = cam_pos; fr.light_dir = light_dir;
    fr.znear = 0.01f; fr.zfar = 3.0f; fr.fog = 0.005f;
    fr.aspect = (float)W / (float)H;
    fr.tan_half_fov = 1.0f / proj[5];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) fr.inv_rot[3 * r + c] = view[4 * c + r];
    fr.W = W; fr.H = H;
    /* proj[5] = f = 1/tan(fov/2): recover tan exactly the way the product does (tanf of the same angle) */
    {
        const float kPi = 3.141592653589f;
        const float fov = kPi * 39.5978f / 180.0f;
        const float fov_deg = fov * (180.0f / kPi);
        fr.tan_half_fov = tanf((fov_deg * 0.5f) * (kPi / 180.0f));
    }
    const float plane[4] = {0.0f, 1.0f, 0.0f, 0.0f};
    const float col_plane[3] = {0.001f, 0.001f, 0.001f}, col_shape[3] = {0.9f, 0.9f, 0.9f};
    const float col_cloth[3] = {0.612f * 1.5f, 0.194f * 1.5f, 0.394f * 1.5f};
    const v4 *P = (const v4 *)pos, *N = (const v4 *)nrm;

    /* picker meshes */
    int nsv = ns * SPH_VERTS, nst = ns * SPH_TRIS;
    v4 *sv = (v4 *)malloc(sizeof(v4) * (nsv + 1)), *sn = (v4 *)malloc(sizeof(v4) * (nsv + 1));
    orc_sphere_mesh(spheres, ns, (float *)sv, (float *)sn, NULL);
    size_t npx = (size_t)W * H;
    unsigned long long *zb = (unsigned long long *)malloc(sizeof(unsigned long long) * npx);
    unsigned *shm = (unsigned *)malloc(sizeof(unsigned) * (size_t)SHADOW_RES * SHADOW_RES);
    memset(zb, 0xff, sizeof(unsigned long long) * npx);
    memset(shm, 0xff, sizeof(unsigned) * (size_t)SHADOW_RES * SHADOW_RES);

    int total = nst + t;
    for (int k = 0; k < total; ++k) {
        v4 p[3];
        int is_sphere = k < nst;
        if (is_sphere) { int a, b, c; sphere_tri(k, &a, &b, &c); p[0] = sv[a]; p[1] = sv[b]; p[2] = sv[c]; }
        else { int q = k - nst; p[0] = P[tris[3 * q]]; p[1] = P[tris[3 * q + 1]]; p[2] = P[tris[3 * q + 2]]; }
        /* shadow pass */
        tri_setup s = setup(light_vp, SHADOW_RES, SHADOW_RES, p);
        if (s.valid) {
            double ax = (double)(s.x[1] - s.x[0]) / 256.0, ay = (double)(s.y[1] - s.y[0]) / 256.0;
            double bx = (double)(s.x[2] - s.x[0]) / 256.0, by = (double)(s.y[2] - s.y[0]) / 256.0;
            double az = (double)s.d[1] - (double)s.d[0], bz = (double)s.d[2] - (double)s.d[0];
            double det = ax * by - ay * bx, dzdx = 0.0, dzdy = 0.0;
            if (det != 0.0) { dzdx = (az * by - bz * ay) / det; dzdy = (bz * ax - az * bx) / det; }
            double slope = fmax(fabs(dzdx), fabs(dzdy)), offset = 8.0 * slope + 8.0 / 16777216.0;
            for (int py = s.miny; py <= s.maxy; ++py)
                for (int px = s.minx; px <= s.maxx; ++px) {
                    double l[3];
                    if (!cover(&s, px, py, l)) continue;
                    double d = l[0] * (double)s.d[0] + l[1] * (double)s.d[1] + l[2] * (double)s.d[2] + offset;
                    if (d < 0.0) continue;
                    unsigned q = quant24(d > 1.0 ? 1.0 : d);
                    unsigned *dst = &shm[(size_t)py * SHADOW_RES + px];
                    if (q < *dst) *dst = q;
                }
        }
        /* camera pass */
        s = setup(vp, W, H, p);
        if (!s.valid) continue;
        if (is_sphere && !s.front) continue;
        unsigned long long id = (unsigned long long)(k + 1);
        for (int py = s.miny; py <= s.maxy; ++py)
            for (int px = s.minx; px <= s.maxx; ++px) {
                double l[3];
                if (!cover(&s, px, py, l)) continue;
                double d = l[0] * (double)s.d[0] + l[1] * (double)s.d[1] + l[2] * (double)s.d[2];
                if (d < 0.0 || d > 1.0) continue;
                unsigned long long key = ((unsigned long long)quant24(d) << 32) | id;
                unsigned long long *dst = &zb[(size_t)py * W + px];
                if (key < *dst) *dst = key;
            }
    }
    /* shading */
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            size_t pix = (size_t)py * W + px;
            unsigned long long key = zb[pix];
            float xn = (((float)px + 0.5f) / (float)W) * 2.0f - 1.0f, yn = (((float)py + 0.5f) / (float)H) * 2.0f - 1.0f;
            float ex = xn * fr.tan_half_fov * fr.aspect, ey = yn * fr.tan_half_fov, ez = -1.0f;
            float dx = fr.inv_rot[0] * ex + fr.inv_rot[1] * ey + fr.inv_rot[2] * ez;
            float dy = fr.inv_rot[3] * ex + fr.inv_rot[4] * ey + fr.inv_rot[5] * ez;
            float dz = fr.inv_rot[6] * ex + fr.inv_rot[7] * ey + fr.inv_rot[8] * ez;
            float denom = plane[0] * dx + plane[1] * dy + plane[2] * dz, hx = 0, hy = 0, hz = 0;
            int plane_hit = 0;
            if (denom < 0.0f) {
                float num = -(plane[0] * cam_pos[0] + plane[1] * cam_pos[1] + plane[2] * cam_pos[2] + plane[3]);
                float tp = num / denom;
                if (tp > 0.0f) {
                    hx = cam_pos[0] + dx * tp; hy = cam_pos[1] + dy * tp; hz = cam_pos[2] + dz * tp;
                    v4 c = xform(vp, hx, hy, hz);
                    double d = (double)(c.z / c.w) * 0.5 + 0.5;
                    if (c.w > 0.0f && d >= 0.0 && d <= 1.0) {
                        unsigned long long pk = (unsigned long long)quant24(d) << 32;
                        if (pk < key) { key = pk; plane_hit = 1; }
                    }
                }
            }
            float col[3] = {0.0f, 0.0f, 0.0f};
            double dwin = 1.0;
            if (key != 0xffffffffffffffffull) {
                dwin = (double)(unsigned)(key >> 32) / DEPTH_MAX;
                int id = (int)(key & 0xffffffffu);
                if (plane_hit && id == 0) {
                    shade(&fr, shm, hx, hy, hz, plane[0], plane[1], plane[2], col_plane, 0.0f, col);
                } else {
                    int k = id - 1, is_sphere = k < nst;
                    v4 p[3], nn[3];
                    if (is_sphere) {
                        int a, b, c; sphere_tri(k, &a, &b, &c);
                        p[0] = sv[a]; p[1] = sv[b]; p[2] = sv[c]; nn[0] = sn[a]; nn[1] = sn[b]; nn[2] = sn[c];
                    } else {
                        int q = k - nst, a = tris[3 * q], b = tris[3 * q + 1], c = tris[3 * q + 2];
                        p[0] = P[a]; p[1] = P[b]; p[2] = P[c]; nn[0] = N[a]; nn[1] = N[b]; nn[2] = N[c];
                    }
                    tri_setup s = setup(vp, W, H, p);
                    double l[3];
                    cover(&s, px, py, l);
                    double q0 = l[0] / (double)s.w[0], q1 = l[1] / (double)s.w[1], q2 = l[2] / (double)s.w[2], qs = q0 + q1 + q2;
                    float b0 = (float)(q0 / qs), b1 = (float)(q1 / qs), b2 = (float)(q2 / qs);
                    const v4 *pa = &p[s.order[0]], *pb = &p[s.order[1]], *pc = &p[s.order[2]];
                    const v4 *na = &nn[s.order[0]], *nb = &nn[s.order[1]], *nc = &nn[s.order[2]];
                    float wx = b0 * pa->x + b1 * pb->x + b2 * pc->x, wy = b0 * pa->y + b1 * pb->y + b2 * pc->y,
                          wz = b0 * pa->z + b1 * pb->z + b2 * pc->z;
                    float nx = b0 * na->x + b1 * nb->x + b2 * nc->x, ny = b0 * na->y + b1 * nb->y + b2 * nc->y,
                          nz = b0 * na->z + b1 * nb->z + b2 * nc->z;
                    if (!s.front) { nx = -nx; ny = -ny; nz = -nz; }
                    shade(&fr, shm, wx, wy, wz, nx, ny, nz, is_sphere ? col_shape : col_cloth, is_sphere ? 0.05f : 0.0f, col);
                }
            }
            rgba[4 * pix + 0] = to_u8(col[0]); rgba[4 * pix + 1] = to_u8(col[1]); rgba[4 * pix + 2] = to_u8(col[2]);
            rgba[4 * pix + 3] = (key != 0xffffffffffffffffull) ? 255 : 0;
            float dw = (float)dwin;
            depth[pix] = 2.0f * fr.zfar * fr.znear / (fr.zfar + fr.znear - (2.0f * dw - 1.0f) * (fr.zfar - fr.znear));
        }
    free(sv); free(sn); free(zb); free(shm);
    return 0;
}
