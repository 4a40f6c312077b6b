// fs_stream_kernels.h -- streaming solver: one kernel per solver stage over HBM-resident SoA particle state.
//
// Works for any particle count (cloths up to 120x120 and meshes do not fit one CU's LDS) and batches every episode of
// the context into each launch (blockIdx.y = episode slot).  Stage list mirrors the reference's own GPU stage timers
// (NvFlex.h:197-223): predict, createCellIndices/createGrid, collideParticles, 30 x {solveSprings, solveContacts,
// applyDeltas}, finalize.  Every thread owns one particle; every particle field is one 16-byte load per lane.
#pragma once
#include "fs_constraints.h"

#define FS_TILE 256

// ---- predict + cell histogram.  reference: gravity NvFlex.h:99, damping :117, invMass == 0 -> kinematic :545
__global__ __launch_bounds__(FS_TILE) void fs_k_predict(const FsEnvDev *envs, const int *ids) {
    if (ids[blockIdx.y] < 0) return;  // retired slot
    const FsEnvDev &E = envs[ids[blockIdx.y]];
    const int i = blockIdx.x * FS_TILE + threadIdx.x;
    if (i >= E.n) return;
    const FsParams &p = E.p;
    const float h = p.dt / (float)p.numSubsteps;
    FsVec4 x = E.pos[i];
    FsVec4 v = E.vel[i];
    E.x0[i] = x;
    E.v0[i] = v;
    FsVec4 xp = x;
    if (x.w > 0.0f) {
        float vx = v.x + h * (p.gravity[0] - p.damping * v.x);
        float vy = v.y + h * (p.gravity[1] - p.damping * v.y);
        float vz = v.z + h * (p.gravity[2] - p.damping * v.z);
        xp.x = x.x + h * vx;
        xp.y = x.y + h * vy;
        xp.z = x.z + h * vz;
    }
    E.xa[i] = xp;
    const float inv = 1.0f / (p.radius + p.particleCollisionMargin);
    int b = fs_bucket((int)floorf(xp.x * inv), (int)floorf(xp.y * inv), (int)floorf(xp.z * inv));
    atomicAdd(&E.cell_count[b], 1);
}

// ---- exclusive scan of the bucket histogram (one workgroup per episode); leaves count/fill zeroed for the next use
__global__ __launch_bounds__(1024) void fs_k_grid_scan(const FsEnvDev *envs, const int *ids) {
    if (ids[blockIdx.x] < 0) return;  // retired slot
    const FsEnvDev &E = envs[ids[blockIdx.x]];
    __shared__ int wave_tot[16];
    constexpr int PER = FS_GRID_BUCKETS / 1024;
    const int t = threadIdx.x;
    int loc[PER];
    int sum = 0;
    for (int k = 0; k < PER; ++k) {
        loc[k] = E.cell_count[t * PER + k];
        sum += loc[k];
    }
    // inclusive scan of `sum` across the 64 lanes of the wave, then across the 16 waves
    int lane = t & 63, wave = t >> 6;
    int inc = sum;
    for (int off = 1; off < 64; off <<= 1) {
        int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    int run = base + inc - sum;
    for (int k = 0; k < PER; ++k) {
        E.cell_fill[t * PER + k] = run;  // running write cursor, starts at the bucket's first slot
        run += loc[k];
    }
    __syncthreads();
    for (int k = 0; k < PER; ++k) E.cell_count[t * PER + k] = 0;
}

// ---- scatter particle ids into buckets.  After this kernel cell_fill[b] == end of bucket b.
__global__ __launch_bounds__(FS_TILE) void fs_k_grid_scatter(const FsEnvDev *envs, const int *ids) {
    if (ids[blockIdx.y] < 0) return;  // retired slot
    const FsEnvDev &E = envs[ids[blockIdx.y]];
    const int i = blockIdx.x * FS_TILE + threadIdx.x;
    if (i >= E.n) return;
    const FsParams &p = E.p;
    const float inv = 1.0f / (p.radius + p.particleCollisionMargin);
    FsVec4 xp = E.xa[i];
    int b = fs_bucket((int)floorf(xp.x * inv), (int)floorf(xp.y * inv), (int)floorf(xp.z * inv));
    int slot = atomicAdd(&E.cell_fill[b], 1);
    E.cell_items[slot] = i;
}

// ---- particle-contact candidates: ascending neighbour id, the (up to) 96 smallest ids.
// A bucket spans [end[b] - size, end[b]); sizes are recovered from consecutive ends (end[b-1] == start[b]).
__global__ __launch_bounds__(FS_TILE) void fs_k_find_neighbors(const FsEnvDev *envs, const int *ids) {
    if (ids[blockIdx.y] < 0) return;  // retired slot
    const FsEnvDev &E = envs[ids[blockIdx.y]];
    const int i = blockIdx.x * FS_TILE + threadIdx.x;
    if (i >= E.n) return;
    const FsParams &p = E.p;
    const float r = p.radius + p.particleCollisionMargin;
    const float r2 = r * r;
    const float inv = 1.0f / r;
    const int n = E.n;
    const int cap = p.maxNeighbors < FS_MAX_NEIGHBORS ? p.maxNeighbors : FS_MAX_NEIGHBORS;
    const FsVec4 xi = E.xa[i];
    const int phi = E.phase[i];
    const FsVec4 ri = E.rest[i];
    const int cx = (int)floorf(xi.x * inv), cy = (int)floorf(xi.y * inv), cz = (int)floorf(xi.z * inv);
    int cnt = 0;
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int b = fs_bucket(cx + dx, cy + dy, cz + dz);
                const int beg = (b == 0) ? 0 : E.cell_fill[b - 1];
                const int end = E.cell_fill[b];
                for (int q = beg; q < end; ++q) {
                    const int j = E.cell_items[q];
                    if (j == i) continue;
                    const FsVec4 xj = E.xa[j];
                    // the bucket may alias a far-away cell: require the true cell to be the visited one
                    if ((int)floorf(xj.x * inv) != cx + dx || (int)floorf(xj.y * inv) != cy + dy ||
                        (int)floorf(xj.z * inv) != cz + dz)
                        continue;
                    float ex = xi.x - xj.x, ey = xi.y - xj.y, ez = xi.z - xj.z;
                    float d2 = ex * ex + ey * ey + ez * ez;
                    if (!(d2 < r2)) continue;
                    if (!fs_pair_allowed(phi, E.phase[j], ri, E.rest[j], r2)) continue;
                    // sorted insert, keep the `cap` smallest ids
                    if (cnt == cap) {
                        if (j > E.nlist[(size_t)(cap - 1) * n + i]) continue;
                        cnt = cap - 1;
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

// ---- one Jacobi iteration: solveSprings + solveContacts + applyDeltas for particle i
__global__ __launch_bounds__(FS_TILE) void fs_k_iterate(const FsEnvDev *envs, const FsShapesDev *shapes, const int *ids,
                                                        int sub, int flip) {
    const int e = ids[blockIdx.y];
    if (e < 0) return;  // retired slot
    const FsEnvDev &E = envs[e];
    const int i = blockIdx.x * FS_TILE + threadIdx.x;
    if (i >= E.n) return;
    const FsParams &p = E.p;
    const FsVec4 *__restrict__ src = flip ? E.xb : E.xa;
    FsVec4 *__restrict__ dst = flip ? E.xa : E.xb;
    FsVec4 xi = src[i];
    if (!(xi.w > 0.0f)) {
        dst[i] = xi;
        return;
    }
    FsAcc a = {0.0f, 0.0f, 0.0f, 0};
    // slot-major (ELL) adjacency: the wave's loads of slot s are contiguous (the CSR rows of neighbouring particles are 12
    // entries apart, i.e. one cache line per lane); slots ascend with the spring id, like the CSR rows.  Four slots per
    // trip: their index / length / stiffness loads and then their four position gathers are in flight together.
    const unsigned un = (unsigned)E.n;
    const int max_deg = E.max_deg;
    for (int s0 = 0; s0 < max_deg; s0 += 4) {
        int jj[4];
        float ll[4], kk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = s0 + q < max_deg;
            const unsigned at = (unsigned)(s0 + q) * un + (unsigned)i;
            jj[q] = in ? E.ell_j[at] : -1;
            ll[q] = in ? E.ell_len[at] : 0.0f;
            kk[q] = in ? E.ell_k[at] : 0.0f;
        }
        FsVec4 xj[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xj[q] = src[jj[q] < 0 ? i : jj[q]];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (jj[q] >= 0) fs_spring(a, xi.x, xi.y, xi.z, xi.w, xj[q], ll[q], kk[q]);
        if (jj[3] < 0) break;  // the padding (-1) is at the tail of every row
    }
    const FsVec4 x0i = E.x0[i];
    const float ri0 = xi.x - x0i.x, ri1 = xi.y - x0i.y, ri2 = xi.z - x0i.z;
    const int nc = E.ncount[i];
    const float restd = p.solidRestDistance, restd2 = restd * restd;
    for (int q = 0; q < nc; ++q) {
        const int j = E.nlist[(size_t)q * E.n + i];
        const FsVec4 xj = src[j];
        const FsVec4 x0j = E.x0[j];
        fs_particle_contact(a, xi.x, xi.y, xi.z, xi.w, ri0, ri1, ri2, xj, xj.x - x0j.x, xj.y - x0j.y, xj.z - x0j.z, restd,
                            restd2, p.particleFriction);
    }
    fs_shape_contacts(a, xi.x, xi.y, xi.z, ri0, ri1, ri2, p, shapes[e], sub);
    fs_apply(a, p.relaxationFactor, xi.x, xi.y, xi.z);
    dst[i] = xi;
}

// ---- finalize: velocity from displacement, maxAcceleration / maxSpeed clamps (NvFlex.h:112-113), sleeping (:110)
__global__ __launch_bounds__(FS_TILE) void fs_k_finalize(const FsEnvDev *envs, const int *ids, int flip) {
    if (ids[blockIdx.y] < 0) return;  // retired slot
    const FsEnvDev &E = envs[ids[blockIdx.y]];
    const int i = blockIdx.x * FS_TILE + threadIdx.x;
    if (i >= E.n) return;
    const FsParams &p = E.p;
    const float h = p.dt / (float)p.numSubsteps;
    const float inv_h = 1.0f / h;
    const FsVec4 x0 = E.x0[i];
    if (!(x0.w > 0.0f)) {
        E.vel[i] = FsVec4{0.0f, 0.0f, 0.0f, 0.0f};
        return;
    }
    const FsVec4 xp = (flip ? E.xb : E.xa)[i];
    const FsVec4 v0 = E.v0[i];
    float vx = (xp.x - x0.x) * inv_h, vy = (xp.y - x0.y) * inv_h, vz = (xp.z - x0.z) * inv_h;
    float ax = vx - v0.x, ay = vy - v0.y, az = vz - v0.z;
    float dv2 = ax * ax + ay * ay + az * az;
    const float maxdv = p.maxAcceleration * h;
    if (dv2 > maxdv * maxdv) {
        float sc = maxdv / sqrtf(dv2);
        vx = v0.x + ax * sc; vy = v0.y + ay * sc; vz = v0.z + az * sc;
    }
    float v2 = vx * vx + vy * vy + vz * vz;
    if (p.maxSpeed < 3.402823466e+38f && v2 > p.maxSpeed * p.maxSpeed) {
        float sc = p.maxSpeed / sqrtf(v2);
        vx = vx * sc; vy = vy * sc; vz = vz * sc;
        v2 = vx * vx + vy * vy + vz * vz;
    }
    const float thr2 = p.sleepThreshold * p.sleepThreshold;
    if (v2 < thr2) {
        E.vel[i] = FsVec4{0.0f, 0.0f, 0.0f, 0.0f};
    } else {
        E.vel[i] = FsVec4{vx, vy, vz, 0.0f};
        E.pos[i] = FsVec4{xp.x, xp.y, xp.z, x0.w};
    }
}
