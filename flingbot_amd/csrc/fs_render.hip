// fs_render.hip -- vertex normals, coverage reward and the software rasteriser of libflingsim.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/flingsim.h"
#include "fs_context.h"
#include "fs_raster_kernels.h"

#define HIP_TRY(call)                                     \
    do {                                                  \
        if (!fs_hip_ok((call), #call)) return FS_ERR_HIP; \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------
// Vertex normals: area-weighted sum of incident triangle normals, normalised, fallback (0,1,0).
// Same formula as the reference's host code (main.cpp:904-919) and what NvFlexGetNormals returns (main.cpp:2286).
// Gather over the vertex->triangle CSR in ascending triangle id = the accumulation order of a sequential loop over
// triangles, so the result is deterministic.
__global__ __launch_bounds__(256) void fs_k_vertex_normals(const FsVec4 *__restrict__ pos, const int *__restrict__ tris,
                                                           const int *__restrict__ vt_off,
                                                           const int *__restrict__ vt_tri, FsVec4 *__restrict__ nrm, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    for (int q = vt_off[i]; q < vt_off[i + 1]; ++q) {
        const int t = vt_tri[q];
        const FsVec4 v0 = pos[tris[3 * t]], v1 = pos[tris[3 * t + 1]], v2 = pos[tris[3 * t + 2]];
        float ax = v1.x - v0.x, ay = v1.y - v0.y, az = v1.z - v0.z;
        float bx = v2.x - v0.x, by = v2.y - v0.y, bz = v2.z - v0.z;
        sx += ay * bz - az * by;
        sy += az * bx - ax * bz;
        sz += ax * by - ay * bx;
    }
    float l = sx * sx + sy * sy + sz * sz;
    FsVec4 o;
    if (l > 0.0f) {
        float inv = 1.0f / sqrtf(l);
        o = FsVec4{sx * inv, sy * inv, sz * inv, 0.0f};
    } else {
        o = FsVec4{0.0f, 1.0f, 0.0f, 0.0f};
    }
    nrm[i] = o;
}

static int ensure_scratch(fs_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->render_scratch_bytes) return FS_OK;
    if (ctx->render_scratch) (void)hipFree(ctx->render_scratch);
    ctx->render_scratch = nullptr;
    ctx->render_scratch_bytes = 0;
    HIP_TRY(hipMalloc(&ctx->render_scratch, bytes));
    ctx->render_scratch_bytes = bytes;
    return FS_OK;
}

// normals of env into device buffer `d_nrm` (float4[n])
static int launch_normals(fs_ctx *ctx, const FsEnv &e, FsVec4 *d_nrm) {
    const int n = e.host.n;
    hipLaunchKernelGGL(fs_k_vertex_normals, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, e.dev.pos, e.topo->tris,
                       e.topo->vt_off, e.topo->vt_tri, d_nrm, n);
    HIP_TRY(hipGetLastError());
    return FS_OK;
}

int fs_normals_env(fs_ctx *ctx, int env, float *out4n) {
    FsEnv &e = ctx->envs[env];
    const size_t bytes = size_t(16) * e.host.n;
    int rc = ensure_scratch(ctx, bytes);
    if (rc != FS_OK) return rc;
    rc = launch_normals(ctx, e, (FsVec4 *)ctx->render_scratch);
    if (rc != FS_OK) return rc;
    void *st = fs_stage(ctx, bytes);
    if (!st) return FS_ERR_HIP;
    HIP_TRY(hipMemcpyAsync(st, ctx->render_scratch, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(out4n, st, bytes);
    return FS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Coverage reward (reference environment/flex_utils.py:358-395, pos=None path => float32 positions):
//   bbox of (x, z); span = extent / 100 (fp32); per particle the slots round((offset -/+ r) / span) clipped to
//   [0, 100]; vectorized_range(start, end) samples floor(k * (end - start) / N + start), k < N = max(end - start) + 1
//   (fp64); cells idx = x * 100 + y clipped to [0, 9999]; area = #cells * span_x * span_y (fp64).
// One workgroup per episode; the 100x100 occupancy grid lives in LDS.
__device__ __forceinline__ float fs_wave_min(float v) {
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float fs_wave_max(float v) {
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int fs_wave_max_i(int v) {
    for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

#define FS_COV_THREADS 1024  // one workgroup per episode: 16 waves go through the three passes over the particles
__global__ __launch_bounds__(FS_COV_THREADS) void fs_k_coverage(const FsEnvDev *envs, double *out, float particle_radius) {
    const FsEnvDev &E = envs[blockIdx.x];
    __shared__ unsigned char grid[10000];
    __shared__ float red[4][FS_COV_THREADS / 64];
    __shared__ int redi[2][FS_COV_THREADS / 64];
    __shared__ int total;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (!E.has_scene || E.n <= 0) {
        if (t == 0) out[blockIdx.x] = 0.0;
        return;
    }
    const int n = E.n;
    float mnx = 3.402823466e+38f, mnz = 3.402823466e+38f, mxx = -3.402823466e+38f, mxz = -3.402823466e+38f;
    for (int i = t; i < n; i += FS_COV_THREADS) {
        const FsVec4 p = E.pos[i];
        mnx = fminf(mnx, p.x); mxx = fmaxf(mxx, p.x);
        mnz = fminf(mnz, p.z); mxz = fmaxf(mxz, p.z);
    }
    mnx = fs_wave_min(mnx); mnz = fs_wave_min(mnz); mxx = fs_wave_max(mxx); mxz = fs_wave_max(mxz);
    if (lane == 0) { red[0][wave] = mnx; red[1][wave] = mnz; red[2][wave] = mxx; red[3][wave] = mxz; }
    for (int q = t; q < 10000; q += FS_COV_THREADS) grid[q] = 0;
    if (t == 0) total = 0;
    __syncthreads();
    for (int w = 0; w < FS_COV_THREADS / 64; ++w) {  // (min / max: the order of the reduction cannot change the result)
        mnx = fminf(mnx, red[0][w]); mnz = fminf(mnz, red[1][w]);
        mxx = fmaxf(mxx, red[2][w]); mxz = fmaxf(mxz, red[3][w]);
    }
    const float span0 = (mxx - mnx) / 100.0f, span1 = (mxz - mnz) / 100.0f;
    // pass A: N = max(end - start) + 1 per axis
    int ex = -2147483647, ez = -2147483647;
    for (int i = t; i < n; i += FS_COV_THREADS) {
        const FsVec4 p = E.pos[i];
        const float ox = p.x - mnx, oz = p.z - mnz;
        long long xl = (long long)rintf((ox - particle_radius) / span0), xh = (long long)rintf((ox + particle_radius) / span0);
        long long zl = (long long)rintf((oz - particle_radius) / span1), zh = (long long)rintf((oz + particle_radius) / span1);
        if (xl < 0) xl = 0; if (xh > 100) xh = 100; if (zl < 0) zl = 0; if (zh > 100) zh = 100;
        int dx = (int)(xh - xl), dz = (int)(zh - zl);
        ex = dx > ex ? dx : ex; ez = dz > ez ? dz : ez;
    }
    ex = fs_wave_max_i(ex); ez = fs_wave_max_i(ez);
    if (lane == 0) { redi[0][wave] = ex; redi[1][wave] = ez; }
    __syncthreads();
    for (int w = 0; w < FS_COV_THREADS / 64; ++w) { ex = redi[0][w] > ex ? redi[0][w] : ex; ez = redi[1][w] > ez ? redi[1][w] : ez; }
    const int Nx = ex + 1, Nz = ez + 1;
    // pass B: mark cells
    for (int i = t; i < n; i += FS_COV_THREADS) {
        const FsVec4 p = E.pos[i];
        const float ox = p.x - mnx, oz = p.z - mnz;
        long long xl = (long long)rintf((ox - particle_radius) / span0), xh = (long long)rintf((ox + particle_radius) / span0);
        long long zl = (long long)rintf((oz - particle_radius) / span1), zh = (long long)rintf((oz + particle_radius) / span1);
        if (xl < 0) xl = 0; if (xh > 100) xh = 100; if (zl < 0) zl = 0; if (zh > 100) zh = 100;
        for (int kz = 0; kz < Nz; ++kz) {
            const long long iz = (long long)floor((double)(kz * (zh - zl)) / (double)Nz + (double)zl);
            for (int kx = 0; kx < Nx; ++kx) {
                const long long ix = (long long)floor((double)(kx * (xh - xl)) / (double)Nx + (double)xl);
                long long idx = ix * 100 + iz;
                idx = idx < 0 ? 0 : (idx > 9999 ? 9999 : idx);
                grid[idx] = 1;
            }
        }
    }
    __syncthreads();
    int cnt = 0;
    for (int q = t; q < 10000; q += FS_COV_THREADS) cnt += grid[q];
    atomicAdd(&total, cnt);
    __syncthreads();
    if (t == 0) out[blockIdx.x] = (double)total * (double)span0 * (double)span1;
}

int fs_coverage_all(fs_ctx *ctx, double *out) {
    hipLaunchKernelGGL(fs_k_coverage, dim3(ctx->n_envs), dim3(FS_COV_THREADS), 0, ctx->stream, ctx->d_envs, ctx->d_coverage,
                       0.00625f);
    HIP_TRY(hipGetLastError());
    const size_t bytes = sizeof(double) * ctx->n_envs;
    void *st = fs_stage(ctx, bytes);
    if (!st) return FS_ERR_HIP;
    HIP_TRY(hipMemcpyAsync(st, ctx->d_coverage, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(out, st, bytes);
    return FS_OK;
}

// picker meshes of `env` into d_sv / d_sn (float4[441 * shapes]); the shapes' previous rotations are host state
// (pyflex.add_sphere / set_shape_states; the device-side picker moves centres only)
static int launch_sphere_mesh(fs_ctx *ctx, int env, FsVec4 *d_sv, FsVec4 *d_sn) {
    const FsEnv &e = ctx->envs[env];
    static const FsSphereTrig trig = [] { FsSphereTrig t; fs_sphere_trig(t); return t; }();
    FsSphereRot rot;
    memset(&rot, 0, sizeof(rot));
    for (int q = 0; q < e.shapes.count && q < FS_MAX_SHAPES; ++q) fs_quat_axes(e.shape_prev_rot[q], rot.a[q]);
    const size_t sph_verts = size_t(e.shapes.count) * FS_SPHERE_VERTS;
    hipLaunchKernelGGL(fs_k_sphere_mesh, dim3((unsigned)((sph_verts + 255) / 256)), dim3(256), 0, ctx->stream,
                       ctx->d_shapes + env, trig, rot, d_sv, d_sn);
    HIP_TRY(hipGetLastError());
    return FS_OK;
}

// The picker meshes alone (test hook of the render path: what fs_render_device rasterises for the shapes).
// verts4 / nrms4: float[4 * 441 * shapes]; tris: int[3 * 800 * shapes] (either may be null).
int fs_sphere_mesh_env(fs_ctx *ctx, int env, float *verts4, float *nrms4, int *tris) {
    const FsEnv &e = ctx->envs[env];
    const int n_sph = e.shapes.count;
    if (n_sph <= 0) return FS_OK;
    const size_t nv = size_t(n_sph) * FS_SPHERE_VERTS, bytes = nv * 16;
    int rc = ensure_scratch(ctx, 2 * bytes + 512);
    if (rc != FS_OK) return rc;
    FsVec4 *d_sv = (FsVec4 *)ctx->render_scratch, *d_sn = d_sv + nv;
    rc = launch_sphere_mesh(ctx, env, d_sv, d_sn);
    if (rc != FS_OK) return rc;
    char *stg = (char *)fs_stage(ctx, 2 * bytes);
    if (!stg) return FS_ERR_HIP;
    HIP_TRY(hipMemcpyAsync(stg, d_sv, 2 * bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (verts4) memcpy(verts4, stg, bytes);
    if (nrms4) memcpy(nrms4, stg + bytes, bytes);
    if (tris)
        for (int t = 0; t < n_sph * FS_SPHERE_TRIS; ++t) fs_sphere_tri(t, tris[3 * t], tris[3 * t + 1], tris[3 * t + 2]);
    return FS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// pyflex.render -- see fs_raster_kernels.h
// Renders into the context's scratch and leaves the frame there: *d_rgba_out (uint8 RGBA, bottom-up rows) and
// *d_depth_out (float32) stay valid until the next render on this context; the work is enqueued on ctx->stream.
int fs_render_device(fs_ctx *ctx, int env, unsigned char **d_rgba_out, float **d_depth_out) {
    FsEnv &e = ctx->envs[env];
    const int W = e.cam.width, H = e.cam.height;
    const int n = e.host.n, T = e.host.t;
    if (W <= 0 || H <= 0 || W > 4096 || H > 4096) { fs_set_error("bad camera size"); return FS_ERR_ARG; }
    FsRasterFrame fr;
    fs_raster_setup(fr, e.cam.pos, e.cam.angle, W, H, e.host.scene_lower, e.host.scene_upper);
    // scratch carve: normals | sphere verts | zbuf (u64 per pixel) | shadow (u32 per texel) | rgba | depth
    const int n_sph = e.shapes.count;
    const size_t sph_verts = size_t(n_sph) * FS_SPHERE_VERTS;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~size_t(255); return o; };
    const size_t o_nrm = take(size_t(16) * n);
    const size_t o_sv = take(size_t(16) * (sph_verts + 1));
    const size_t o_sn = take(size_t(16) * (sph_verts + 1));
    const size_t o_z = take(size_t(8) * W * H);
    const size_t o_sh = take(size_t(4) * FS_SHADOW_RES * FS_SHADOW_RES);
    const size_t o_rgba = take(size_t(4) * W * H);
    const size_t o_depth = take(size_t(4) * W * H);
    int rc = ensure_scratch(ctx, off);
    if (rc != FS_OK) return rc;
    char *base = (char *)ctx->render_scratch;
    FsVec4 *d_nrm = (FsVec4 *)(base + o_nrm);
    FsVec4 *d_sv = (FsVec4 *)(base + o_sv), *d_sn = (FsVec4 *)(base + o_sn);
    unsigned long long *d_z = (unsigned long long *)(base + o_z);
    unsigned int *d_shadow = (unsigned int *)(base + o_sh);
    unsigned char *d_rgba = (unsigned char *)(base + o_rgba);
    float *d_depth = (float *)(base + o_depth);
    hipStream_t st = ctx->stream;

    rc = launch_normals(ctx, e, d_nrm);
    if (rc != FS_OK) return rc;
    HIP_TRY(hipMemsetAsync(d_z, 0xff, size_t(8) * W * H, st));
    HIP_TRY(hipMemsetAsync(d_shadow, 0xff, size_t(4) * FS_SHADOW_RES * FS_SHADOW_RES, st));
    if (n_sph > 0) {
        rc = launch_sphere_mesh(ctx, env, d_sv, d_sn);
        if (rc != FS_OK) return rc;
    }
    const int n_sph_tris = n_sph * FS_SPHERE_TRIS;
    // shadow pass (depth only, from the light), then camera pass (depth + primitive id), then shading
    const int total_tris = T + n_sph_tris;
    if (total_tris > 0) {
        hipLaunchKernelGGL(fs_k_raster_shadow, dim3((total_tris + 63) / 64), dim3(64), 0, st, fr, e.dev.pos, e.topo->tris, T,
                           d_sv, n_sph_tris, d_shadow);
        hipLaunchKernelGGL(fs_k_raster_camera, dim3((total_tris + 63) / 64), dim3(64), 0, st, fr, e.dev.pos, e.topo->tris, T,
                           d_sv, n_sph_tris, d_z);
    }
    hipLaunchKernelGGL(fs_k_shade, dim3((W + 15) / 16, (H + 15) / 16), dim3(16, 16), 0, st, fr, e.dev.pos, d_nrm,
                       e.topo->tris, T, d_sv, d_sn, n_sph_tris, d_z, d_shadow, d_rgba, d_depth);
    HIP_TRY(hipGetLastError());
    *d_rgba_out = d_rgba;
    *d_depth_out = d_depth;
    return FS_OK;
}

int fs_render_env(fs_ctx *ctx, int env, unsigned char *rgba, float *depth) {
    unsigned char *d_rgba = nullptr;
    float *d_depth = nullptr;
    int rc = fs_render_device(ctx, env, &d_rgba, &d_depth);
    if (rc != FS_OK) return rc;
    const FsEnv &e = ctx->envs[env];
    hipStream_t st = ctx->stream;
    const size_t px = size_t(e.cam.width) * e.cam.height;
    char *stg = (char *)fs_stage(ctx, px * 8);
    if (!stg) return FS_ERR_HIP;
    HIP_TRY(hipMemcpyAsync(stg, d_rgba, px * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(stg + px * 4, d_depth, px * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(rgba, stg, px * 4);
    memcpy(depth, stg + px * 4, px * 4);
    return FS_OK;
}
