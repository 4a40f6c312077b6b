// fs_fused_kernel.h -- fused LDS-resident solver: ONE workgroup advances ONE cloth episode by whole frames.
//
// MI355X mapping: a 64x64 cloth is 4096 particles; its predicted positions (float4, 64 KiB), substep-start
// positions (3 x 16 KiB) and the spatial-hash bins (24 KiB) all fit the 160 KiB LDS of one CU, so the 4 substeps x
// 30 Jacobi iterations of a frame (reference softgym_cloth.h:154-155) run without touching HBM for particle state:
// 1024 threads (16 waves, 4 per SIMD), 4 particles per thread held in registers across the frame, neighbour
// positions gathered from LDS with ds_read_b128.  HBM/L2 traffic is the read-only spring adjacency (shared by all
// episodes of the same cloth, L2-resident) and the per-particle contact candidate lists.  Grid = #episodes: 256 CUs
// advance 256 episodes concurrently.
//
// Arithmetic is the same per-particle code as the streaming kernels (fs_constraints.h), so both back-ends and the
// CPU oracle agree bit for bit.
#pragma once
#include "fs_constraints.h"

#define FS_FUSED_THREADS 1024
#define FS_FUSED_PPT 4
#define FS_FUSED_MAX_PARTICLES (FS_FUSED_THREADS * FS_FUSED_PPT)
#define FS_FUSED_MAX_DEG 64
#define FS_FUSED_BUCKETS 4096  // 32 x 4 x 32 wrapped cells (>= 3 per axis: a 3x3x3 block never aliases itself)

// LDS carve (bytes): X float4[4096] | X0x,X0y,X0z float[4096] | cursor int[4096] | items ushort[4096] | scan int[16]
#define FS_FUSED_OFF_X 0
#define FS_FUSED_OFF_X0 (FS_FUSED_MAX_PARTICLES * 16)
#define FS_FUSED_OFF_CUR (FS_FUSED_OFF_X0 + FS_FUSED_MAX_PARTICLES * 12)
#define FS_FUSED_OFF_ITEMS (FS_FUSED_OFF_CUR + FS_FUSED_BUCKETS * 4)
#define FS_FUSED_OFF_SCAN (FS_FUSED_OFF_ITEMS + FS_FUSED_MAX_PARTICLES * 2)
#define FS_FUSED_LDS_BYTES (FS_FUSED_OFF_SCAN + 64)

__device__ __forceinline__ int fs_fused_bucket(int cx, int cy, int cz) {
    return (cx & 31) | ((cy & 3) << 5) | ((cz & 31) << 7);
}

__global__ __launch_bounds__(FS_FUSED_THREADS) void fs_k_fused_step(const FsEnvDev *envs, const FsShapesDev *shapes,
                                                                    const int *ids, int n_steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    FsVec4 *X = (FsVec4 *)(smem + FS_FUSED_OFF_X);
    float *X0x = (float *)(smem + FS_FUSED_OFF_X0);
    float *X0y = X0x + FS_FUSED_MAX_PARTICLES;
    float *X0z = X0y + FS_FUSED_MAX_PARTICLES;
    int *cursor = (int *)(smem + FS_FUSED_OFF_CUR);
    unsigned short *items = (unsigned short *)(smem + FS_FUSED_OFF_ITEMS);
    int *wave_tot = (int *)(smem + FS_FUSED_OFF_SCAN);

    const int e = ids[blockIdx.x];
    const FsEnvDev &E = envs[e];
    const FsShapesDev &sh = shapes[e];
    const FsParams &p = E.p;
    const int n = E.n;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;

    const float h = p.dt / (float)p.numSubsteps;
    const float inv_h = 1.0f / h;
    const float rad = p.radius + p.particleCollisionMargin;
    const float rad2 = rad * rad;
    const float inv_rad = 1.0f / rad;
    const float restd = p.solidRestDistance, restd2 = restd * restd;
    const float maxdv = p.maxAcceleration * h;
    const float thr2 = p.sleepThreshold * p.sleepThreshold;
    const int ncap = p.maxNeighbors < FS_MAX_NEIGHBORS ? p.maxNeighbors : FS_MAX_NEIGHBORS;
    const int max_deg = E.max_deg;

    // own particles: i = t + k * 1024
    FsVec4 pos[FS_FUSED_PPT], vel[FS_FUSED_PPT];
#pragma unroll
    for (int k = 0; k < FS_FUSED_PPT; ++k) {
        const int i = t + k * FS_FUSED_THREADS;
        if (i < n) {
            pos[k] = E.pos[i];
            vel[k] = E.vel[i];
        } else {
            pos[k] = FsVec4{0.0f, 0.0f, 0.0f, 0.0f};
            vel[k] = pos[k];
        }
    }

    for (int frame = 0; frame < n_steps; ++frame) {
        for (int sub = 0; sub < p.numSubsteps; ++sub) {
            // ---- predict (same arithmetic as fs_k_predict)
            FsVec4 xp[FS_FUSED_PPT];
            int ncnt[FS_FUSED_PPT];
            for (int q = t; q < FS_FUSED_BUCKETS; q += FS_FUSED_THREADS) cursor[q] = 0;
#pragma unroll
            for (int k = 0; k < FS_FUSED_PPT; ++k) {
                const int i = t + k * FS_FUSED_THREADS;
                xp[k] = pos[k];
                if (pos[k].w > 0.0f) {
                    float vx = vel[k].x + h * (p.gravity[0] - p.damping * vel[k].x);
                    float vy = vel[k].y + h * (p.gravity[1] - p.damping * vel[k].y);
                    float vz = vel[k].z + h * (p.gravity[2] - p.damping * vel[k].z);
                    xp[k].x = pos[k].x + h * vx;
                    xp[k].y = pos[k].y + h * vy;
                    xp[k].z = pos[k].z + h * vz;
                }
                if (i < n) {
                    X[i] = xp[k];
                    X0x[i] = pos[k].x; X0y[i] = pos[k].y; X0z[i] = pos[k].z;
                }
            }
            __syncthreads();
            // ---- spatial hash in LDS: histogram -> exclusive scan -> scatter
            int bucket[FS_FUSED_PPT];
#pragma unroll
            for (int k = 0; k < FS_FUSED_PPT; ++k) {
                const int i = t + k * FS_FUSED_THREADS;
                bucket[k] = fs_fused_bucket((int)floorf(xp[k].x * inv_rad), (int)floorf(xp[k].y * inv_rad),
                                            (int)floorf(xp[k].z * inv_rad));
                if (i < n) atomicAdd(&cursor[bucket[k]], 1);
            }
            __syncthreads();
            {
                constexpr int PER = FS_FUSED_BUCKETS / FS_FUSED_THREADS;
                int loc[PER], sum = 0;
#pragma unroll
                for (int k = 0; k < PER; ++k) { loc[k] = cursor[t * PER + k]; sum += loc[k]; }
                int inc = sum;
                for (int off = 1; off < 64; off <<= 1) {
                    int o = __shfl_up(inc, off, 64);
                    if (lane >= off) inc += o;
                }
                if (lane == 63) wave_tot[wave] = inc;
                __syncthreads();
                int run = inc - sum;
                for (int w = 0; w < wave; ++w) run += wave_tot[w];
#pragma unroll
                for (int k = 0; k < PER; ++k) { cursor[t * PER + k] = run; run += loc[k]; }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < FS_FUSED_PPT; ++k) {
                const int i = t + k * FS_FUSED_THREADS;
                if (i < n) {
                    int slot = atomicAdd(&cursor[bucket[k]], 1);
                    items[slot] = (unsigned short)i;
                }
            }
            __syncthreads();
            // ---- particle-contact candidates (ascending id, <= 96 smallest), lists live in global memory
#pragma unroll 1
            for (int k = 0; k < FS_FUSED_PPT; ++k) {
                const int i = t + k * FS_FUSED_THREADS;
                int cnt = 0;
                if (i < n) {
                    const FsVec4 xi = xp[k];
                    const int cx = (int)floorf(xi.x * inv_rad), cy = (int)floorf(xi.y * inv_rad),
                              cz = (int)floorf(xi.z * inv_rad);
                    int phi = 0;
                    FsVec4 ri = FsVec4{0.0f, 0.0f, 0.0f, 0.0f};
                    bool have_meta = false;
                    for (int dz = -1; dz <= 1; ++dz)
                        for (int dy = -1; dy <= 1; ++dy)
                            for (int dx = -1; dx <= 1; ++dx) {
                                const int b = fs_fused_bucket(cx + dx, cy + dy, cz + dz);
                                const int beg = (b == 0) ? 0 : cursor[b - 1];
                                const int end = cursor[b];
                                for (int q = beg; q < end; ++q) {
                                    const int j = items[q];
                                    if (j == i) continue;
                                    const FsVec4 xj = X[j];
                                    float ex = xi.x - xj.x, ey = xi.y - xj.y, ez = xi.z - xj.z;
                                    float d2 = ex * ex + ey * ey + ez * ez;
                                    if (!(d2 < rad2)) continue;
                                    if ((int)floorf(xj.x * inv_rad) != cx + dx || (int)floorf(xj.y * inv_rad) != cy + dy ||
                                        (int)floorf(xj.z * inv_rad) != cz + dz)
                                        continue;
                                    if (!have_meta) { phi = E.phase[i]; ri = E.rest[i]; have_meta = true; }
                                    if (!fs_pair_allowed(phi, E.phase[j], ri, E.rest[j], rad2)) continue;
                                    if (cnt == ncap) {
                                        if (j > E.nlist[(size_t)(ncap - 1) * n + i]) continue;
                                        cnt = ncap - 1;
                                    }
                                    int s = cnt;
                                    while (s > 0) {
                                        int prev = E.nlist[(size_t)(s - 1) * n + i];
                                        if (prev < j) break;
                                        E.nlist[(size_t)s * n + i] = prev;
                                        --s;
                                    }
                                    E.nlist[(size_t)s * n + i] = j;
                                    ++cnt;
                                }
                            }
                    E.ncount[i] = cnt;
                }
                ncnt[k] = cnt;
            }
            // (no barrier needed: the iterations only read X / X0, which are stable since the predict barrier)

            // ---- Jacobi iterations: gather from LDS, compute in registers, barrier, publish, barrier
            for (int it = 0; it < p.numIterations; ++it) {
#pragma unroll
                for (int k = 0; k < FS_FUSED_PPT; ++k) {
                    const int i = t + k * FS_FUSED_THREADS;
                    if (i < n && xp[k].w > 0.0f) {
                        FsAcc a = {0.0f, 0.0f, 0.0f, 0};
                        const float xi0 = xp[k].x, xi1 = xp[k].y, xi2 = xp[k].z, wi = xp[k].w;
                        for (int s = 0; s < max_deg; ++s) {
                            const int j = E.ell_j[(size_t)s * n + i];
                            if (j < 0) break;
                            fs_spring(a, xi0, xi1, xi2, wi, X[j], E.ell_len[(size_t)s * n + i], E.ell_k[(size_t)s * n + i]);
                        }
                        const float ri0 = xi0 - pos[k].x, ri1 = xi1 - pos[k].y, ri2 = xi2 - pos[k].z;
                        for (int s = 0; s < ncnt[k]; ++s) {
                            const int j = E.nlist[(size_t)s * n + i];
                            const FsVec4 xj = X[j];
                            fs_particle_contact(a, xi0, xi1, xi2, wi, ri0, ri1, ri2, xj, xj.x - X0x[j], xj.y - X0y[j],
                                                xj.z - X0z[j], restd, restd2, p.particleFriction);
                        }
                        fs_shape_contacts(a, xi0, xi1, xi2, ri0, ri1, ri2, p, sh, sub);
                        fs_apply(a, p.relaxationFactor, xp[k].x, xp[k].y, xp[k].z);
                    }
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < FS_FUSED_PPT; ++k) {
                    const int i = t + k * FS_FUSED_THREADS;
                    if (i < n) X[i] = xp[k];
                }
                __syncthreads();
            }

            // ---- finalize (same arithmetic as fs_k_finalize)
#pragma unroll
            for (int k = 0; k < FS_FUSED_PPT; ++k) {
                if (!(pos[k].w > 0.0f)) {
                    vel[k] = FsVec4{0.0f, 0.0f, 0.0f, 0.0f};
                    continue;
                }
                float vx = (xp[k].x - pos[k].x) * inv_h, vy = (xp[k].y - pos[k].y) * inv_h, vz = (xp[k].z - pos[k].z) * inv_h;
                float ax = vx - vel[k].x, ay = vy - vel[k].y, az = vz - vel[k].z;
                float dv2 = ax * ax + ay * ay + az * az;
                if (dv2 > maxdv * maxdv) {
                    float sc = maxdv / sqrtf(dv2);
                    vx = vel[k].x + ax * sc; vy = vel[k].y + ay * sc; vz = vel[k].z + az * sc;
                }
                float v2 = vx * vx + vy * vy + vz * vz;
                if (p.maxSpeed < 3.402823466e+38f && v2 > p.maxSpeed * p.maxSpeed) {
                    float sc = p.maxSpeed / sqrtf(v2);
                    vx = vx * sc; vy = vy * sc; vz = vz * sc;
                    v2 = vx * vx + vy * vy + vz * vz;
                }
                if (v2 < thr2) {
                    vel[k] = FsVec4{0.0f, 0.0f, 0.0f, 0.0f};
                } else {
                    vel[k] = FsVec4{vx, vy, vz, 0.0f};
                    pos[k].x = xp[k].x; pos[k].y = xp[k].y; pos[k].z = xp[k].z;
                }
            }
            // next substep's predict overwrites X / X0 only after every thread left the last iteration barrier
        }
    }

#pragma unroll
    for (int k = 0; k < FS_FUSED_PPT; ++k) {
        const int i = t + k * FS_FUSED_THREADS;
        if (i < n) {
            E.pos[i] = pos[k];
            E.vel[i] = vel[k];
        }
    }
}
