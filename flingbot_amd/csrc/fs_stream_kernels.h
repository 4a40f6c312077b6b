// fs_stream_kernels.h -- streaming solver: one kernel per solver stage over HBM-resident SoA particle state.
//
// Works for any particle count (cloths up to 120x120 and meshes do not fit one CU's LDS) and batches every episode of
// the context into each launch (blockIdx.y = episode slot).  Stage list mirrors the reference's own GPU stage timers
// (NvFlex.h:197-223): predict, createCellIndices/createGrid, collideParticles, 30 x {solveSprings, solveContacts,
// applyDeltas}, finalize.  Every thread owns one particle; every particle field is one 16-byte load per lane.
#pragma once
#include "fs_constraints.h"

#define FS_TILE 256

// Cell -> bucket of the global hash, like the fused kernel's: the row (cy, cz) is hashed and x added on top, so the three
// cells cx-1..cx+1 a search visits per row are adjacent buckets = ONE contiguous run of the bucket-ordered arrays.
__device__ __forceinline__ int fs_stream_bucket(int cx, int cy, int cz) {
    const unsigned h = (unsigned)cy * 0x85EBCA77u + (unsigned)cz * 0xC2B2AE3Du;
    return (int)(((h ^ (h >> 15)) * 0x2C1B3C6Du >> 18) + (unsigned)cx) & (FS_GRID_BUCKETS - 1);
}
#ifndef FS_STREAM_CHUNK
#define FS_STREAM_CHUNK 6  // springs whose loads are issued together in fs_k_iterate
#endif

// ---- predict + cell histogram.  reference: gravity NvFlex.h:99, damping :117, invMass == 0 -> kinematic :545
// XCD-affine tile mapping.  Every per-particle kernel of the step is launched 1-D with ceil8(ne) * gx workgroups; the
// dispatcher places workgroup n on XCD n % 8, so episode slot `by` runs entirely on XCD by % 8 -- in every kernel of the
// step (fs_k_grid_scan's one workgroup per episode lands there too).  The per-XCD L2 keeps what the previous kernel wrote,
// so an episode's positions and lists are L2 hits in the next kernel instead of round trips to the memory side (measured:
// 104x104 x 16 episodes 1.18 -> 0.97 ms/step, crumpled 64x64 x 64 episodes 1.89 -> 1.64).
__device__ __forceinline__ bool fs_stream_tile(int gx, int ne, int &bx, int &by) {
    const int n = blockIdx.x, k = n >> 3;
    by = (k / gx) * 8 + (n & 7);
    bx = k - (k / gx) * gx;
    return by < ne;
}

// ---- launch table.  The kernels of a step reach their episode through the launch list: ids[slot] -> envs[episode] -> arrays,
// two DEPENDENT scalar loads in front of every kernel's first vector load -- and a step is 129 dependent kernels of ~9 us.
// Once per fs_step_stream call this kernel copies the descriptor of every listed episode into the slot-indexed table the
// other kernels then receive as `envs` (slot_env = the episode id, -1 for a slot a device-side loop has retired), so each
// of them starts with ONE scalar load.  One workgroup per slot.
// It also works out the slot's sphere sweeps for every substep of the frame (FsSlotSweeps), with fs_shape_sweep's own
// expressions: what every particle of the episode used to recompute in each of the 120 iterations of a frame.
__global__ __launch_bounds__(64) void fs_k_slot_table(const FsEnvDev *envs, const int *ids, FsEnvDev *table,
                                                      const FsShapesDev *shapes, FsSlotSweeps *sweeps) {
    static_assert(sizeof(FsEnvDev) % 4 == 0, "copied as dwords");
    const int slot = blockIdx.x, e = ids[slot];
    uint32_t *dst = (uint32_t *)(table + slot);
    if (e >= 0) {
        const uint32_t *src = (const uint32_t *)(envs + e);
        for (unsigned k = threadIdx.x; k < sizeof(FsEnvDev) / 4; k += 64) dst[k] = src[k];
        const FsShapesDev &sh = shapes[e];
        FsSlotSweeps &W = sweeps[slot];
        const int S = envs[e].p.numSubsteps, count = sh.count < FS_MAX_SHAPES ? sh.count : FS_MAX_SHAPES;
        const bool fits = S >= 1 && S <= FS_SWEEP_MAX_SUBSTEPS;
        if (threadIdx.x == 0) { W.count = count; W.substeps = fits ? S : 0; }
        if (fits)
            for (int k = threadIdx.x; k < S * count; k += 64) {
                const int sub = k / count, q = k - sub * count;
                float c0, c1, c2, s0, s1, s2;
                fs_shape_sweep(sh, q, sub, (float)S, c0, c1, c2, s0, s1, s2);
                W.c[sub][q] = FsVec4{c0, c1, c2, sh.pos[q].w};
                W.s[sub][q] = FsVec4{s0, s1, s2, 0.0f};
            }
    }
    __syncthreads();
    if (threadIdx.x == 0) table[slot].slot_env = e;
}

__global__ __launch_bounds__(FS_TILE) void fs_k_predict(const FsEnvDev *envs, const int *ids, int gx, int ne) {
    int bx, by;
    if (!fs_stream_tile(gx, ne, bx, by)) return;
    const FsEnvDev &E = envs[by];  // the slot's own copy of its episode's descriptor (fs_k_slot_table)
    if (E.slot_env < 0) return;    // retired slot
    const int i = bx * FS_TILE + threadIdx.x;
    if (i >= E.n) return;
    const FsParams &p = E.p;
    const float h = p.dt / (float)p.numSubsteps;
    FsVec4 x = fs_ld4o(E.pos, (unsigned)i);  // scalar base + 32-bit offset (global_load saddr form), like every access below
    FsVec4 v = fs_ld4o(E.vel, (unsigned)i);
    fs_st4o(E.x0, (unsigned)i, x);
    fs_st4o(E.v0, (unsigned)i, v);
    FsVec4 xp = x;
    if (x.w > 0.0f) {
        float vx = v.x + h * (p.gravity[0] - p.damping * v.x);
        float vy = v.y + h * (p.gravity[1] - p.damping * v.y);
        float vz = v.z + h * (p.gravity[2] - p.damping * v.z);
        xp.x = x.x + h * vx;
        xp.y = x.y + h * vy;
        xp.z = x.z + h * vz;
    }
    fs_st4o(E.xa, (unsigned)i, xp);
    const float inv = 1.0f / (p.radius + p.particleCollisionMargin);
    int b = fs_stream_bucket((int)floorf(xp.x * inv), (int)floorf(xp.y * inv), (int)floorf(xp.z * inv));
    atomicAdd(&E.cell_count[b], 1);
}

// ---- exclusive scan of the bucket histogram (one workgroup per episode); leaves count/fill zeroed for the next use
__global__ __launch_bounds__(1024) void fs_k_grid_scan(const FsEnvDev *envs, const int *ids) {
    const FsEnvDev &E = envs[blockIdx.x];
    if (E.slot_env < 0) return;  // retired slot
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
__global__ __launch_bounds__(FS_TILE) void fs_k_grid_scatter(const FsEnvDev *envs, const int *ids, int gx, int ne) {
    int bx, by;
    if (!fs_stream_tile(gx, ne, bx, by)) return;
    const FsEnvDev &E = envs[by];  // the slot's own copy of its episode's descriptor (fs_k_slot_table)
    if (E.slot_env < 0) return;    // retired slot
    const int i = bx * FS_TILE + threadIdx.x;
    if (i >= E.n) return;
    const FsParams &p = E.p;
    const float inv = 1.0f / (p.radius + p.particleCollisionMargin);
    FsVec4 xp = fs_ld4o(E.xa, (unsigned)i);
    int b = fs_stream_bucket((int)floorf(xp.x * inv), (int)floorf(xp.y * inv), (int)floorf(xp.z * inv));
    int slot = atomicAdd(&E.cell_fill[b], 1);
    // bucket-ordered copy of the predicted positions with the particle id in w: the search reads candidates sequentially
    // (xb is free until the first Jacobi iteration writes it)
    fs_st4o(E.xb, (unsigned)slot, FsVec4{xp.x, xp.y, xp.z, __int_as_float(i)});
}

// ---- the substep boundary in ONE launch (cloths up to FS_BOUND_MAX particles; round 2): finalize of the substep that just
// ended (FIN), predict of the one that starts (PRE), the bucket histogram, its exclusive scan and the bucket-ordered copy of
// the predicted positions -- what fs_k_finalize, fs_k_predict, fs_k_grid_scan and fs_k_grid_scatter do in four dependent
// launches.  One 1024-thread workgroup per episode: the histogram lives in LDS (64 KiB), an LDS atomicAdd returns the
// particle's rank inside its bucket, so after the scan every particle knows its slot without a second pass of atomics; the
// thread keeps its (up to 16) predicted positions in registers from the first pass to the last.  The arithmetic is that of
// the four kernels, statement for statement; the order of the particles INSIDE a bucket differs (it was the order of the
// global atomics before), which the search cannot see: its lists are sorted sets.  A frame is 129 dependent launches instead
// of 140; measured in EXPERIMENTS.md (round-2 notes, 4.2 "The launch floor").  (Round 4 built the same boundary as tiles of
// 4096 particles per workgroup with a last-ticket workgroup doing scan + scatter, for launches whose episodes cannot fill the
// chip: slower at every size -- global histogram atomics and the fences cost more than the idle CUs; EXPERIMENTS R4.2.)
#define FS_BOUND_THREADS 1024
#define FS_BOUND_PPT 16
#define FS_BOUND_MAX (FS_BOUND_THREADS * FS_BOUND_PPT)
// histogram entry of bucket b: one pad word per 32 buckets, so that the scan's 16 consecutive buckets per thread (lane
// stride 16 words = 2 banks for a whole wave) spread over all banks (measured: fs_k_boundary 24.5 -> 17.3 us, EXPERIMENTS.md round-2 notes 4.2)
#define FS_BOUND_IDX(b) ((b) + ((b) >> 5))
#define FS_BOUND_HIST (FS_GRID_BUCKETS + FS_GRID_BUCKETS / 32)
#define FS_BOUND_LDS_BYTES (FS_BOUND_HIST * 4 + 64)
// Round 4: the kernel was one dependent round trip after the other -- per particle `load, s_waitcnt vmcnt(0), load, ...`, the
// second and third loads behind the `invMass > 0` branch, and behind every store a reload of the next array pointer from the
// descriptor (a store through E.pos may alias *E as far as the compiler can tell): 32 us for a 104 x 104 cloth at 0.07
// VALU-active.  Now the descriptor's fields are read once into registers, a particle's three inputs are requested
// unconditionally, FS_BOUND_CHUNK particles' loads go out before the first of them is used, and the bucket-ordered copy
// re-reads the predicted positions (L2 hits, batched the same way) instead of keeping 16 float4 per thread alive across the
// scan.  The arithmetic per particle is unchanged.
#define FS_BOUND_CHUNK 4
template <bool FIN, bool PRE>
__global__ __launch_bounds__(FS_BOUND_THREADS) void fs_k_boundary(const FsEnvDev *__restrict__ envs, const int *ids, int flip) {
    static_assert(FS_GRID_BUCKETS == FS_BOUND_THREADS * 16 && FS_BOUND_MAX <= (1 << 14), "16 buckets per thread, 14-bit ranks");
    static_assert(FS_BOUND_PPT % FS_BOUND_CHUNK == 0, "whole chunks");
    extern __shared__ __attribute__((aligned(16))) int bound_smem[];
    int *hist = bound_smem, *wave_tot = bound_smem + FS_BOUND_HIST;
    const FsEnvDev &E = envs[blockIdx.x];
    if (E.slot_env < 0) return;  // retired slot
    const int n = E.n, t = threadIdx.x;
    // the descriptor's fields, once
    FsVec4 *const g_pos = E.pos, *const g_vel = E.vel, *const g_x0 = E.x0, *const g_v0 = E.v0, *const g_xa = E.xa, *const g_xb = E.xb;
    int *const g_fill = E.cell_fill;
    const FsVec4 *const g_last = flip ? g_xb : g_xa;  // the iterate the substep that just ended left its result in
    const float h = E.p.dt / (float)E.p.numSubsteps;
    const float gr0 = E.p.gravity[0], gr1 = E.p.gravity[1], gr2 = E.p.gravity[2], damping = E.p.damping;
    const float max_acc = E.p.maxAcceleration, max_speed = E.p.maxSpeed, sleep_thr = E.p.sleepThreshold;
    const float inv_cell = 1.0f / (E.p.radius + E.p.particleCollisionMargin);
    if (PRE) {
#pragma unroll
        for (int k = 0; k < 17; ++k)
            if (t + k * FS_BOUND_THREADS < FS_BOUND_HIST) hist[t + k * FS_BOUND_THREADS] = 0;
        __syncthreads();
    }
    int code[FS_BOUND_PPT];  // bucket | rank inside the bucket << 14
#pragma unroll
    for (int k0 = 0; k0 < FS_BOUND_PPT; k0 += FS_BOUND_CHUNK) {
        if (k0 * FS_BOUND_THREADS >= n) {  // (uniform: nothing of the cloth is left for this chunk)
#pragma unroll
            for (int k = 0; k < FS_BOUND_CHUNK; ++k) code[k0 + k] = -1;
            continue;
        }
        FsVec4 in_a[FS_BOUND_CHUNK], in_b[FS_BOUND_CHUNK], in_c[FS_BOUND_CHUNK];
#pragma unroll
        for (int k = 0; k < FS_BOUND_CHUNK; ++k) {  // every load of the chunk first (lanes past the end read particle n - 1)
            const int ir = t + (k0 + k) * FS_BOUND_THREADS;
            const unsigned i = (unsigned)(ir < n ? ir : n - 1);
            if (FIN) {
                in_a[k] = fs_ld4o(g_x0, i);
                in_b[k] = fs_ld4o(g_last, i);
                in_c[k] = fs_ld4o(g_v0, i);
            } else {
                in_a[k] = fs_ld4o(g_pos, i);
                in_b[k] = fs_ld4o(g_vel, i);
            }
        }
        int bucket[FS_BOUND_CHUNK];
#pragma unroll
        for (int k = 0; k < FS_BOUND_CHUNK; ++k) {
            const unsigned i = (unsigned)(t + (k0 + k) * FS_BOUND_THREADS);
            bucket[k] = -1;
            if ((int)i < n) {
                FsVec4 x, v;
                if (FIN) {  // fs_k_finalize: velocity from displacement, maxAcceleration / maxSpeed clamps, sleeping
                    const float inv_h = 1.0f / h;
                    const FsVec4 x0 = in_a[k];
                    x = x0;  // kinematic or asleep: the position stays (pos == x0 since the predict that copied it)
                    v = FsVec4{0.0f, 0.0f, 0.0f, 0.0f};
                    if (x0.w > 0.0f) {
                        const FsVec4 xp = in_b[k];
                        const FsVec4 v0 = in_c[k];
                        float vx = (xp.x - x0.x) * inv_h, vy = (xp.y - x0.y) * inv_h, vz = (xp.z - x0.z) * inv_h;
                        float ax = vx - v0.x, ay = vy - v0.y, az = vz - v0.z;
                        float dv2 = ax * ax + ay * ay + az * az;
                        const float maxdv = max_acc * h;
                        if (dv2 > maxdv * maxdv) {
                            float sc = maxdv / sqrtf(dv2);
                            vx = v0.x + ax * sc; vy = v0.y + ay * sc; vz = v0.z + az * sc;
                        }
                        float v2 = vx * vx + vy * vy + vz * vz;
                        if (max_speed < 3.402823466e+38f && v2 > max_speed * max_speed) {
                            float sc = max_speed / sqrtf(v2);
                            vx = vx * sc; vy = vy * sc; vz = vz * sc;
                            v2 = vx * vx + vy * vy + vz * vz;
                        }
                        const float thr2 = sleep_thr * sleep_thr;
                        if (!(v2 < thr2)) {
                            v = FsVec4{vx, vy, vz, 0.0f};
                            x = FsVec4{xp.x, xp.y, xp.z, x0.w};
                            fs_st4o(g_pos, i, x);
                        }
                    }
                    fs_st4o(g_vel, i, v);
                } else {
                    x = in_a[k];
                    v = in_b[k];
                }
                if (PRE) {  // fs_k_predict
                    fs_st4o(g_x0, i, x);
                    fs_st4o(g_v0, i, v);
                    FsVec4 xp = x;
                    if (x.w > 0.0f) {
                        float vx = v.x + h * (gr0 - damping * v.x);
                        float vy = v.y + h * (gr1 - damping * v.y);
                        float vz = v.z + h * (gr2 - damping * v.z);
                        xp.x = x.x + h * vx;
                        xp.y = x.y + h * vy;
                        xp.z = x.z + h * vz;
                    }
                    fs_st4o(g_xa, i, xp);
                    bucket[k] = fs_stream_bucket((int)floorf(xp.x * inv_cell), (int)floorf(xp.y * inv_cell), (int)floorf(xp.z * inv_cell));
                }
            }
        }
        if (PRE) {  // the chunk's LDS atomics together: the return value is the particle's rank inside its bucket
            int rank[FS_BOUND_CHUNK];
#pragma unroll
            for (int k = 0; k < FS_BOUND_CHUNK; ++k) rank[k] = bucket[k] >= 0 ? atomicAdd(&hist[FS_BOUND_IDX(bucket[k])], 1) : 0;
#pragma unroll
            for (int k = 0; k < FS_BOUND_CHUNK; ++k) code[k0 + k] = bucket[k] >= 0 ? (bucket[k] | (rank[k] << 14)) : -1;
        }
    }
    if (!PRE) return;
    __syncthreads();
    // exclusive scan of the histogram: 16 consecutive buckets per thread, lanes, waves (fs_k_grid_scan)
    int loc[16], sum = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        loc[k] = hist[FS_BOUND_IDX(t * 16 + k)];
        sum += loc[k];
    }
    const int lane = t & 63, wave = t >> 6;
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int run = inc - sum;
    for (int w2 = 0; w2 < wave; ++w2) run += wave_tot[w2];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        hist[FS_BOUND_IDX(t * 16 + k)] = run;  // first slot of the bucket
        run += loc[k];
    }
    __syncthreads();
    // what the search reads: the END of bucket b = the first slot of bucket b + 1; written coalesced
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int b = t + k * FS_BOUND_THREADS;
        g_fill[b] = b + 1 < FS_GRID_BUCKETS ? hist[FS_BOUND_IDX(b + 1)] : n;
    }
    // bucket-ordered copy of the predicted positions with the particle id in w (fs_k_grid_scatter); this workgroup wrote
    // them above, the re-read is an L2 hit
#pragma unroll
    for (int k0 = 0; k0 < FS_BOUND_PPT; k0 += FS_BOUND_CHUNK) {
        if (k0 * FS_BOUND_THREADS >= n) continue;
        FsVec4 xp[FS_BOUND_CHUNK];
#pragma unroll
        for (int k = 0; k < FS_BOUND_CHUNK; ++k) {
            const int ir = t + (k0 + k) * FS_BOUND_THREADS;
            xp[k] = fs_ld4o(g_xa, (unsigned)(ir < n ? ir : n - 1));
        }
#pragma unroll
        for (int k = 0; k < FS_BOUND_CHUNK; ++k) {
            const int c = code[k0 + k];
            if (c >= 0) {
                const int slot = hist[FS_BOUND_IDX(c & (FS_GRID_BUCKETS - 1))] + (c >> 14);
                fs_st4o(g_xb, (unsigned)slot, FsVec4{xp[k].x, xp[k].y, xp[k].z, __int_as_float(t + (k0 + k) * FS_BOUND_THREADS)});
            }
        }
    }
}

// ---- particle-contact candidates: ascending neighbour id, the (up to) 96 smallest ids.
// Same two-phase scheme as the fused kernel (fs_fused_kernel.h), on global arrays: threads take the particles in BUCKET
// order (thread <-> slot of the sorted copy in xb); phase A scans the 9 runs (3 adjacent buckets each) four candidates per
// trip and parks a packed entry (slot | 4-bit hit mask) per trip with hits in a per-thread LDS queue; phase B walks the
// queue one survivor per trip: id, self test, phase / rest-pose filter (set membership on the packed rest-near ids when
// the whole cloth is one phase), sorted duplicate-free insertion with the smallest ids (FS_NB_STAGED_STREAM = 8) staged in registers.
// A bucket spans [end[b-1], end[b]) of the sorted arrays (cell_fill holds the ends after the scatter).
#define FS_STREAM_FINDQ 32
// STENCIL: every episode of the launch is a grid cloth in find mode 4 (the host checks the launch list): the packed rest-near
// ids are never loaded and the kernel fits 5 waves per SIMD instead of 4.
template <bool STENCIL>
__global__ __launch_bounds__(FS_TILE) void fs_k_find_neighbors(const FsEnvDev *envs, const FsSlotSweeps *shapes, const int *ids,
                                                               int sub, int gx, int ne) {
    int bx, by;
    if (!fs_stream_tile(gx, ne, bx, by)) return;
    const FsEnvDev &E = envs[by];  // the slot's own copy of its episode's descriptor (fs_k_slot_table)
    if (E.slot_env < 0) return;    // retired slot
    __shared__ uint32_t queue_s[FS_STREAM_FINDQ][FS_TILE];
    const int qs = bx * FS_TILE + threadIdx.x;
    const int n = E.n;
    if (qs >= n) return;
    const FsParams &p = E.p;
    const float r = p.radius + p.particleCollisionMargin;
    const int cap = p.maxNeighbors < FS_MAX_NEIGHBORS ? p.maxNeighbors : FS_MAX_NEIGHBORS;
    const FsFindConsts c = {n, cap, r * r, 1.0f / r, E.find_mode, E.gp_dimx, E.gp_magic};
    const fs_gcv4 xs = (fs_gcv4)E.xb;
    const fs_gci fill = (fs_gci)E.cell_fill, phase = (fs_gci)E.phase;
    const fs_gi nlist = (fs_gi)E.nlist;
    const FsVec4 xsi = fs_ld4(E.xb, qs);
    const FsVec4 xi = FsVec4{xsi.x, xsi.y, xsi.z, 0.0f};
    const int i = __float_as_int(xsi.w);
    // collideShapes rides along (this kernel is the once-per-substep pass over the predicted positions): the particle's shape
    // candidates go into the upper bits of its candidate-count word
    const int shape_bits = (int)(fs_swept_shape_candidates(p, shapes[by], sub, xi.x, xi.y, xi.z) << FS_SHAPE_MASK_SHIFT);
    if (c.mode == 3) {  // one phase without the SelfCollide flag: no pairs at all
        E.ncount[i] = shape_bits;
        return;
    }
    FsNearWords near;
#pragma unroll
    for (int q = 0; q < 8; ++q) near.w[q] = (!STENCIL && c.mode == 1) ? E.restnear_w[(size_t)q * n + i] : 0xffffffffu;
    const int cx = (int)floorf(xi.x * c.inv_rad), cy = (int)floorf(xi.y * c.inv_rad), cz = (int)floorf(xi.z * c.inv_rad);
    int phi = 0, qn = 0;
    FsNbListT<FS_NB_STAGED_STREAM> L = fs_nb_empty<FS_NB_STAGED_STREAM>();
    FsVec4 ri = FsVec4{0.0f, 0.0f, 0.0f, 0.0f};
    bool have_meta = false;
    uint32_t *queue = &queue_s[0][threadIdx.x];

    auto drain = [&]() {  // phase B over the queued entries
        int e = 0, qb = 0;
        unsigned m = 0u;
        for (;;) {
            if (!m) {
                if (e == qn) break;
                const uint32_t u = queue[e * FS_TILE];
                ++e;
                m = u & 15u;
                qb = (int)(u >> 4);
            }
            const int at = qb + __builtin_ctz(m);
            m &= m - 1u;
            const int j = __float_as_int(xs[at < n ? at : n - 1].w);
            if (j == i) continue;
            fs_fused_accept<STENCIL, FS_NB_STAGED_STREAM>(c, i, j, L, phi, ri, have_meta, phase, E.rest, nlist, near);
        }
        qn = 0;
    };

    // the bounds of all nine runs are requested together: nine dependent round trips (bounds -> positions, run after run)
    // become one
    int run_b0[9], run_beg[9], run_end[9];
#pragma unroll
    for (int r9 = 0; r9 < 9; ++r9) {
        const int b0 = fs_stream_bucket(cx - 1, cy + (r9 % 3 - 1), cz + (r9 / 3 - 1));
        run_b0[r9] = b0;
        run_beg[r9] = (b0 == 0) ? 0 : fill[b0 - 1];
        run_end[r9] = fill[b0 + 2 - (FS_GRID_BUCKETS - 1) > 0 ? FS_GRID_BUCKETS - 1 : b0 + 2];
    }
#pragma unroll
    for (int r9 = 0; r9 < 9; ++r9) {
            const int b0 = run_b0[r9];
            const int wrap = b0 + 2 - (FS_GRID_BUCKETS - 1);  // > 0: that many buckets continue at bucket 0
            for (int seg = 0; seg < 2; ++seg) {
                int beg, end;
                if (seg == 0) {
                    beg = run_beg[r9];
                    end = run_end[r9];
                } else {
                    if (wrap <= 0) break;
                    beg = 0;
                    end = fill[wrap - 1];
                }
                for (int q = beg & ~3; q < end; q += 4) {
                    unsigned m = 0u;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int at = q + k < n ? q + k : n - 1;  // (clamped slots are masked out below)
                        const FsVec4 xj = fs_ld4(E.xb, at);
                        const float ex = xi.x - xj.x, ey = xi.y - xj.y, ez = xi.z - xj.z;
                        const float d2 = ex * ex + ey * ey + ez * ez;
                        m |= (unsigned)(d2 < c.rad2) << k;
                    }
                    const int lo = beg - q > 0 ? beg - q : 0, hi = end - q < 4 ? end - q : 4;
                    m &= ((1u << (hi - lo)) - 1u) << lo;
                    if (m) {
                        if (qn == FS_STREAM_FINDQ) drain();  // queue full (dense crumple): work it off first
                        queue[qn * FS_TILE] = ((uint32_t)q << 4) | m;
                        ++qn;
                    }
                }
            }
        }
    drain();
    E.ncount[i] = fs_fused_nb_finish(c, i, L, nlist) | shape_bits;
}

// ---- one Jacobi iteration: solveSprings + solveContacts + applyDeltas for particle i
// CODED: the spring slots come from the one-byte codes + the (offset, length, stiffness) dictionary the workgroup holds
// in LDS (fs_scene.h build_stream_codes) instead of the three ELL arrays: 16 bytes of adjacency per particle and
// iteration instead of 144, which is what keeps the adjacency of a launch in the L2s.
template <int CHUNK, bool EAGER, bool CODED>
__device__ __forceinline__ void fs_iterate_particle(const FsEnvDev &E, const FsSlotSweeps &shape_set, int i, int sub, int flip,
                                                    const FsVec4 *sdict) {
    const FsParams &p = E.p;
    const FsVec4 *__restrict__ src = flip ? E.xb : E.xa;
    FsVec4 *__restrict__ dst = flip ? E.xa : E.xb;
    // EAGER (small launches, latency-bound: a lone episode's step is ~140 dependent kernels of a few round trips each):
    // everything whose address does not depend on loaded data is requested before the first use -- own position, substep-start
    // position, candidate count, the first four candidate ids -- so the kernel is three dependent round trips long
    // (addresses -> ids -> neighbour positions) instead of seven.
    FsVec4 xi = src[i];
    FsU32x4 cw = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    if (CODED) cw = E.scode[i];
    FsVec4 x0i;
    int nc = 0, cj0[4] = {-1, -1, -1, -1};
    {   // both forms: the loads whose addresses depend on nothing but i go out first
        x0i = E.x0[i];
        nc = E.ncount[i];  // (count | shape candidates << 8: split below)
#pragma unroll
        for (int k = 0; k < 4; ++k) cj0[k] = E.nlist[(size_t)k * E.n + i];  // slots beyond the count hold stale ids: masked below
    }
    if (!(xi.w > 0.0f)) {
        dst[i] = xi;
        return;
    }
    const unsigned shape_mask = (unsigned)nc >> FS_SHAPE_MASK_SHIFT;
    nc &= FS_NCOUNT_MASK;
    FsAcc a = {0.0f, 0.0f, 0.0f, 0};
    // slot-major (ELL) adjacency: the wave's loads of slot s are contiguous (the CSR rows of neighbouring particles are 12
    // entries apart, i.e. one cache line per lane); slots ascend with the spring id, like the CSR rows.  CHUNK slots per
    // trip: their index / length / stiffness loads and then their position gathers are in flight together.
    const unsigned un = (unsigned)E.n;
    const int max_deg = E.max_deg;
    FsVec4 cx0[4], c00[4];
    for (int s0 = 0; s0 < max_deg; s0 += CHUNK) {
        int jj[CHUNK];
        float ll[CHUNK], kk[CHUNK];
        if (CODED) {
#pragma unroll
            for (int q = 0; q < CHUNK; ++q) {
                const int s = s0 + q;
                const unsigned word = s < 8 ? (s < 4 ? cw.x : cw.y) : (s < 12 ? cw.z : cw.w);
                const unsigned code = s < 16 ? (word >> (8 * (s & 3))) & 255u : 255u;
                const FsVec4 d = sdict[code];  // entry 255 is all zero: the slot gathers the particle itself, length 0
                jj[q] = code == 255u ? -1 : i + __float_as_int(d.x);
                ll[q] = d.y;
                kk[q] = d.z;
            }
        } else {
#pragma unroll
            for (int q = 0; q < CHUNK; ++q) {
                const bool in = s0 + q < max_deg;
                const unsigned at = (unsigned)(s0 + q) * un + (unsigned)i;
                jj[q] = in ? E.ell_j[at] : -1;
                ll[q] = in ? E.ell_len[at] : 0.0f;
                kk[q] = in ? E.ell_k[at] : 0.0f;
            }
        }
        FsVec4 xj[CHUNK];
#pragma unroll
        for (int q = 0; q < CHUNK; ++q) xj[q] = src[jj[q] < 0 ? i : jj[q]];
        if (EAGER && s0 == 0) {  // the first candidates' positions travel with the first springs' gathers
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = (k < nc && cj0[k] >= 0 && cj0[k] < E.n) ? cj0[k] : i;
                cx0[k] = src[j];
                c00[k] = E.x0[j];
            }
        }
#pragma unroll
        for (int q = 0; q < CHUNK; ++q)
            fs_spring_bf(a, xi.x, xi.y, xi.z, xi.w, xj[q], ll[q], kk[q]);  // a padded slot gathers the particle itself: length 0, inactive
        if (jj[CHUNK - 1] < 0) break;  // the padding (-1) is at the tail of every row
    }
    const float ri0 = xi.x - x0i.x, ri1 = xi.y - x0i.y, ri2 = xi.z - x0i.z;
    const float restd = p.solidRestDistance, restd2 = restd * restd;
    // four candidates per trip.  The ids of trip t+1 are requested together with the positions / substep-start positions of
    // trip t (the first trip's ids came with the particle's own loads), so a trip costs one round trip, not two;
    // evaluation order is unchanged
    int cj[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cj[k] = (k < nc && cj0[k] >= 0 && cj0[k] < E.n) ? cj0[k] : -1;
    for (int q0 = 0; q0 < nc; q0 += 4) {
        int cjn[4] = {-1, -1, -1, -1};
        if (q0 + 4 < nc) {
#pragma unroll
            for (int k = 0; k < 4; ++k) cjn[k] = q0 + 4 + k < nc ? E.nlist[(size_t)(q0 + 4 + k) * E.n + i] : -1;
        }
        FsVec4 cx[4], c0[4];
        if (EAGER && q0 == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { cx[k] = cx0[k]; c0[k] = c00[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = cj[k] < 0 ? i : cj[k];
                cx[k] = src[j];
                c0[k] = E.x0[j];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (cj[k] >= 0)
                fs_particle_contact(a, xi.x, xi.y, xi.z, xi.w, ri0, ri1, ri2, cx[k], cx[k].x - c0[k].x, cx[k].y - c0[k].y,
                                    cx[k].z - c0[k].z, restd, restd2, p.particleFriction);
#pragma unroll
        for (int k = 0; k < 4; ++k) cj[k] = cjn[k];
    }
    fs_swept_shape_contacts(a, xi.x, xi.y, xi.z, ri0, ri1, ri2, p, shape_set, sub, shape_mask);
    fs_apply(a, p.relaxationFactor, xi.x, xi.y, xi.z);
    dst[i] = xi;
}

// GRID form of the iteration (grid cloths, FsEnvDev::gp_count > 0): the neighbour ids of particle (ix, iz) follow from the
// canonical (dx, dz) offsets in the episode descriptor -- scalar data -- so the twelve spring gathers are requested together
// with the particle's own loads instead of one round trip later (behind the adjacency); lengths and stiffnesses come from
// the one-byte codes + dictionary as in the CODED form.  The dictionary entry of the thread is requested first and staged
// while those loads are in flight.  In-bounds canonical entries, taken in canonical order, ARE the particle's springs in
// spring-id order (verified per particle by build_grid_pattern), so the accumulation order is unchanged.  Used for the
// big launches only (>= 96 x 4096 particles: 104x104 x 64 episodes 2.63 -> 2.48 ms/step, crumpled 64x64 x 128 3.15 -> 3.01);
// below that the barrier of the dictionary staging costs more than the saved round trip (measured: slower than the
// latency form for 1..32 episodes, also with the lengths taken from the ELL arrays instead of the dictionary).
#define FS_GRID_SLOTS 12
__device__ __forceinline__ void fs_iterate_particle_grid(const FsEnvDev &E, const FsSlotSweeps &shape_set, int i_raw, bool valid,
                                                         int sub, int flip, FsVec4 *sdict) {
    const FsParams &p = E.p;
    const FsVec4 *__restrict__ src = flip ? E.xb : E.xa;
    FsVec4 *__restrict__ dst = flip ? E.xa : E.xb;
    const int t = threadIdx.x;
    const FsVec4 dent = t < E.sdict_size ? E.sdict[t] : FsVec4{0.0f, 0.0f, 0.0f, 0.0f};  // oldest load: staged below
    const int i = valid ? i_raw : 0;
    FsVec4 xi = src[i];
    const FsU32x4 cw = E.scode[i];
    const FsVec4 x0i = E.x0[i];
    const int ncw = valid ? E.ncount[i] : 0;
    const int nc = ncw & FS_NCOUNT_MASK;
    const unsigned shape_mask = (unsigned)ncw >> FS_SHAPE_MASK_SHIFT;
    int cj0[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cj0[k] = E.nlist[(size_t)k * E.n + i];  // slots beyond the count hold stale ids: masked below
    const int dimx = E.gp_dimx, dimz = E.gp_dimz, cnt = E.gp_count;
    const int iz = i / dimx, ix = i - iz * dimx;
    FsVec4 xj[FS_GRID_SLOTS];
    unsigned inb = 0u;
#pragma unroll
    for (int q = 0; q < FS_GRID_SLOTS; ++q) {
        const int jx = ix + E.gp_dx[q], jz = iz + E.gp_dz[q];
        const bool in = q < cnt && (unsigned)jx < (unsigned)dimx && (unsigned)jz < (unsigned)dimz;
        inb |= (unsigned)in << q;
        xj[q] = src[in ? jz * dimx + jx : i];
    }
    sdict[t] = dent;
    __syncthreads();
    if (!valid) return;
    if (!(xi.w > 0.0f)) {
        dst[i] = xi;
        return;
    }
    FsAcc a = {0.0f, 0.0f, 0.0f, 0};
    int slot = 0;
#pragma unroll
    for (int q = 0; q < FS_GRID_SLOTS; ++q) {
        if ((inb >> q) & 1u) {
            const unsigned word = slot < 8 ? (slot < 4 ? cw.x : cw.y) : (slot < 12 ? cw.z : cw.w);
            const FsVec4 d = sdict[(word >> (8 * (slot & 3))) & 255u];
            fs_spring_bf(a, xi.x, xi.y, xi.z, xi.w, xj[q], d.y, d.z);
            ++slot;
        }
    }
    const float ri0 = xi.x - x0i.x, ri1 = xi.y - x0i.y, ri2 = xi.z - x0i.z;
    const float restd = p.solidRestDistance, restd2 = restd * restd;
    // candidates: four per trip, the ids of the next trip requested with the positions of the current one
    int cj[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cj[k] = (k < nc && cj0[k] >= 0 && cj0[k] < E.n) ? cj0[k] : -1;
    for (int q0 = 0; q0 < nc; q0 += 4) {
        int cjn[4] = {-1, -1, -1, -1};
        if (q0 + 4 < nc) {
#pragma unroll
            for (int k = 0; k < 4; ++k) cjn[k] = q0 + 4 + k < nc ? E.nlist[(size_t)(q0 + 4 + k) * E.n + i] : -1;
        }
        FsVec4 cx[4], c0[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = cj[k] < 0 ? i : cj[k];
            cx[k] = src[j];
            c0[k] = E.x0[j];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (cj[k] >= 0)
                fs_particle_contact(a, xi.x, xi.y, xi.z, xi.w, ri0, ri1, ri2, cx[k], cx[k].x - c0[k].x, cx[k].y - c0[k].y,
                                    cx[k].z - c0[k].z, restd, restd2, p.particleFriction);
#pragma unroll
        for (int k = 0; k < 4; ++k) cj[k] = cjn[k];
    }
    fs_swept_shape_contacts(a, xi.x, xi.y, xi.z, ri0, ri1, ri2, p, shape_set, sub, shape_mask);
    fs_apply(a, p.relaxationFactor, xi.x, xi.y, xi.z);
    dst[i] = xi;
}

__global__ __launch_bounds__(FS_TILE) void fs_k_iterate_grid(const FsEnvDev *envs, const FsSlotSweeps *shapes, const int *ids,
                                                             int sub, int flip, int gx, int ne) {
    __shared__ FsVec4 sdict[256];
    int bx, by;
    if (!fs_stream_tile(gx, ne, bx, by)) return;
    const FsEnvDev &E = envs[by];
    const int e = E.slot_env;
    if (e < 0) return;  // retired slot
    if (bx * FS_TILE >= E.n) return;  // whole workgroup beyond this episode's particles
    const int i = bx * FS_TILE + threadIdx.x;
    fs_iterate_particle_grid(E, shapes[by], i, i < E.n, sub, flip, sdict);
}

// GRID-L form of the iteration (grid cloths with the canonical spring list, FsEnvDev::gp_L_ok): like the GRID form the
// neighbour ids follow from (column, row) + the canonical (dx, dz) list -- compile-time constants here, the host has
// verified the cloth against FS_G64_DX_LIST / FS_G64_DZ_LIST -- and the rest lengths come from the per-particle table
// gp_L[12][n] in canonical slot order (coalesced), the stiffness per slot from the descriptor (scalar).  Nothing has to be
// decoded or staged: no dictionary, no workgroup barrier, and ALL loads of the spring phase -- own position, twelve
// neighbour positions, twelve rest lengths, substep-start position, candidate count and the first candidate ids -- leave
// in one round trip.  Per-particle accumulation order = canonical order = spring-id order (build_grid_pattern).
template <bool POSK>
__device__ __forceinline__ void fs_iterate_particle_gridl(const FsEnvDev &E, const FsSlotSweeps &shape_set, int i, int sub, int flip) {
    const FsParams &p = E.p;
    // global address space + 32-bit indices: global_load with a scalar base instead of flat loads behind 64-bit VALU adds
    const FsVec4 *src = flip ? E.xb : E.xa;
    FsVec4 *dst = flip ? E.xa : E.xb;
    constexpr int cdx[FS_G64_SLOTS] = FS_G64_DX_LIST, cdz[FS_G64_SLOTS] = FS_G64_DZ_LIST;
    const unsigned un = (unsigned)E.n, ui = (unsigned)i;
    FsVec4 xi = fs_ld4o(src, ui);
    const FsVec4 x0i = fs_ld4o(E.x0, ui);
    const int ncw = fs_ldo(E.ncount, ui);
    const int nc = ncw & FS_NCOUNT_MASK;
    const unsigned shape_mask = (unsigned)ncw >> FS_SHAPE_MASK_SHIFT;
    int cj0[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cj0[k] = fs_ldo(E.nlist, (unsigned)k * un + ui);  // slots beyond the count hold stale ids: masked below
    const int dimx = E.gp_dimx, dimz = E.gp_dimz;
    const int iz = (int)__umulhi(ui, E.gp_magic), ix = i - iz * dimx;
    FsVec4 xj[FS_G64_SLOTS];
    float L[FS_G64_SLOTS];
    unsigned inb = 0u;
#pragma unroll
    for (int q = 0; q < FS_G64_SLOTS; ++q) {
        const bool in = (unsigned)(ix + cdx[q]) < (unsigned)dimx && (unsigned)(iz + cdz[q]) < (unsigned)dimz;
        inb |= (unsigned)in << q;
        xj[q] = fs_ld4o(src, in ? (unsigned)(i + cdz[q] * dimx + cdx[q]) : ui);
        L[q] = fs_ldo(E.g64_L, (unsigned)q * un + ui);
    }
    if (!(xi.w > 0.0f)) {
        fs_st4o(dst, ui, xi);
        return;
    }
    FsAcc a = {0.0f, 0.0f, 0.0f, 0};
    // (An equal-mass fast form like the fused grid-64 kernel's -- 26 instead of 37 VALU instructions per spring, wave-uniform
    // fallback -- was measured here and is SLOWER at every launch size (64 episodes: 1.42 -> 1.51 ms per step): this kernel
    // runs one round of 4-5 waves per SIMD and is bound by its load latency and the launch, not by VALU issue, and the second
    // code path costs registers.)
#pragma unroll
    for (int q = 0; q < FS_G64_SLOTS; ++q)
        fs_spring_bfm<POSK>(a, xi.x, xi.y, xi.z, xi.w, xj[q], L[q], E.gp_k[q], (inb >> q) & 1u);
    const float ri0 = xi.x - x0i.x, ri1 = xi.y - x0i.y, ri2 = xi.z - x0i.z;
    const float restd = p.solidRestDistance, restd2 = restd * restd;
    // candidates: four per trip, the ids of the next trip requested with the positions of the current one
    int cj[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cj[k] = (k < nc && cj0[k] >= 0 && cj0[k] < E.n) ? cj0[k] : -1;
    for (int q0 = 0; q0 < nc; q0 += 4) {
        int cjn[4] = {-1, -1, -1, -1};
        if (q0 + 4 < nc) {
#pragma unroll
            for (int k = 0; k < 4; ++k) cjn[k] = q0 + 4 + k < nc ? fs_ldo(E.nlist, (unsigned)(q0 + 4 + k) * un + ui) : -1;
        }
        FsVec4 cx[4], c0[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned j = cj[k] < 0 ? ui : (unsigned)cj[k];
            cx[k] = fs_ld4o(src, j);
            c0[k] = fs_ld4o(E.x0, j);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (cj[k] >= 0)
                fs_particle_contact(a, xi.x, xi.y, xi.z, xi.w, ri0, ri1, ri2, cx[k], cx[k].x - c0[k].x, cx[k].y - c0[k].y,
                                    cx[k].z - c0[k].z, restd, restd2, p.particleFriction);
#pragma unroll
        for (int k = 0; k < 4; ++k) cj[k] = cjn[k];
    }
    fs_swept_shape_contacts(a, xi.x, xi.y, xi.z, ri0, ri1, ri2, p, shape_set, sub, shape_mask);
    fs_apply(a, p.relaxationFactor, xi.x, xi.y, xi.z);
    fs_st4o(dst, ui, xi);
}

// POSK: every episode of the launch has positive stiffnesses only (no tethers; FsEnvDev::gp_halvable, decided by the host).
// 10 % fewer VALU instructions per spring but 106 instead of 96 VGPRs (4 instead of 5 waves per SIMD; forcing 5 spills), so
// the launcher takes it only for launches of at most one round of 4 waves per SIMD (<= 262144 particles), where occupancy
// beyond 4 buys nothing: 64x64 cloths x 1 / 8 / 32 / 64 episodes 0.913 / 1.054 / 1.133 / 1.484 -> 0.880 / 1.026 / 1.099 / 1.454 ms
// per step; larger launches measured slower with it and keep the general form.
template <bool POSK>
__global__ __launch_bounds__(FS_TILE) void fs_k_iterate_gridl(const FsEnvDev *envs, const FsSlotSweeps *shapes, const int *ids,
                                                              int sub, int flip, int gx, int ne) {
    int bx, by;
    if (!fs_stream_tile(gx, ne, bx, by)) return;
    const FsEnvDev &E = envs[by];
    const int e = E.slot_env;
    if (e < 0) return;  // retired slot
    const int i = bx * FS_TILE + threadIdx.x;
#ifdef FS_BLOCK_CLOCKS  // developer build: the constant 100 MHz clock at this workgroup's entry and exit, in rows 94 / 95 (by
                        // `flip`) of the neighbour table (scripts/stream_clocks.py); wave 0 of the workgroup only
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (i < E.n) fs_iterate_particle_gridl<POSK>(E, shapes[by], i, sub, flip);
#ifdef FS_BLOCK_CLOCKS
    if (threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long tr1 = __builtin_amdgcn_s_memrealtime();
        E.nlist[(size_t)(94 + flip) * E.n + i] = (int)(unsigned)tr0;
        E.nlist[(size_t)(94 + flip) * E.n + i + 1] = (int)(unsigned)tr1;
    }
#endif
}

// THROUGHPUT form of the grid-L iteration (round 6, EXPERIMENTS R6.4; launches above 262 144 particles of tether-free cloths).
// Such a launch is VALU-bound (0.19 VALU-active per wave-cycle of a ceiling of 0.2), not one wave's critical path, so the
// trade-offs of fs_iterate_particle_gridl turn around: (1) the springs are gathered and evaluated in two halves of six -- half
// the live registers, more waves per SIMD to hide the second round trip; (2) equal masses and positive stiffnesses
// (wave-uniform check per half) leave a spring nothing to decide but `length > 0`: no out-of-grid test (a slot that leaves the
// grid gathers the particle itself: length 0, inactive by itself), the mass ratio folded into g64_kh = k / 2 -- fs_spring_fast;
// a wavefront with a pinned particle among a half's neighbours takes fs_spring_bfm for that half.  An active spring performs
// the same operations in the same order in either form, the accumulation order is the canonical slot order: same bits.
__device__ __forceinline__ void fs_iterate_particle_gridl_tp(const FsEnvDev &E, const FsSlotSweeps &shape_set, int i, int sub, int flip) {
    const FsParams &p = E.p;
    const FsVec4 *src = flip ? E.xb : E.xa;
    FsVec4 *dst = flip ? E.xa : E.xb;
    constexpr int cdx[FS_G64_SLOTS] = FS_G64_DX_LIST, cdz[FS_G64_SLOTS] = FS_G64_DZ_LIST;
    const unsigned un = (unsigned)E.n, ui = (unsigned)i;
    FsVec4 xi = fs_ld4o(src, ui);
    if (!(xi.w > 0.0f)) {
        fs_st4o(dst, ui, xi);
        return;
    }
    const int dimx = E.gp_dimx, dimz = E.gp_dimz;
    const int iz = (int)__umulhi(ui, E.gp_magic), ix = i - iz * dimx;
    FsAcc a = {0.0f, 0.0f, 0.0f, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        FsVec4 xj[6];
        float L[6];
        unsigned inb = 0u;
        bool same = true;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int q = 6 * h + r;
            const bool in = (unsigned)(ix + cdx[q]) < (unsigned)dimx && (unsigned)(iz + cdz[q]) < (unsigned)dimz;
            inb |= (unsigned)in << r;
            xj[r] = fs_ld4o(src, in ? (unsigned)(i + cdz[q] * dimx + cdx[q]) : ui);
            L[r] = fs_ldo(E.g64_L + (size_t)q * un, ui);   // (the slot's row as a scalar base: one shared vector offset for all twelve)
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) same = same & (xj[r].w == xi.w);
        if (__builtin_amdgcn_ballot_w64(!same) == 0ull) {
#pragma unroll
            for (int r = 0; r < 6; ++r) fs_spring_fast(a, xi.x, xi.y, xi.z, xj[r], L[r], E.g64_kh[6 * h + r]);
        } else {
#pragma unroll
            for (int r = 0; r < 6; ++r)
                fs_spring_bfm<true>(a, xi.x, xi.y, xi.z, xi.w, xj[r], L[r], E.gp_k[6 * h + r], (inb >> r) & 1u);
        }
        __builtin_amdgcn_sched_barrier(0);   // the second half's gathers stay behind the first half's arithmetic
    }
    const FsVec4 x0i = fs_ld4o(E.x0, ui);
    const int ncw = fs_ldo(E.ncount, ui);
    const int nc = ncw & FS_NCOUNT_MASK;
    const unsigned shape_mask = (unsigned)ncw >> FS_SHAPE_MASK_SHIFT;
    const float ri0 = xi.x - x0i.x, ri1 = xi.y - x0i.y, ri2 = xi.z - x0i.z;
    const float restd = p.solidRestDistance, restd2 = restd * restd;
    int cj[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = k < nc ? fs_ldo(E.nlist, (unsigned)k * un + ui) : -1;
        cj[k] = (c >= 0 && c < E.n) ? c : -1;
    }
    for (int q0 = 0; q0 < nc; q0 += 4) {
        int cjn[4] = {-1, -1, -1, -1};
        if (q0 + 4 < nc) {
#pragma unroll
            for (int k = 0; k < 4; ++k) cjn[k] = q0 + 4 + k < nc ? fs_ldo(E.nlist, (unsigned)(q0 + 4 + k) * un + ui) : -1;
        }
        FsVec4 cx[4], c0[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned j = cj[k] < 0 ? ui : (unsigned)cj[k];
            cx[k] = fs_ld4o(src, j);
            c0[k] = fs_ld4o(E.x0, j);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (cj[k] >= 0)
                fs_particle_contact(a, xi.x, xi.y, xi.z, xi.w, ri0, ri1, ri2, cx[k], cx[k].x - c0[k].x, cx[k].y - c0[k].y,
                                    cx[k].z - c0[k].z, restd, restd2, p.particleFriction);
#pragma unroll
        for (int k = 0; k < 4; ++k) cj[k] = cjn[k];
    }
    fs_swept_shape_contacts(a, xi.x, xi.y, xi.z, ri0, ri1, ri2, p, shape_set, sub, shape_mask);
    fs_apply(a, p.relaxationFactor, xi.x, xi.y, xi.z);
    fs_st4o(dst, ui, xi);
}
__global__ __launch_bounds__(FS_TILE) void fs_k_iterate_gridl_tp(const FsEnvDev *envs, const FsSlotSweeps *shapes, const int *ids,
                                                                 int sub, int flip, int gx, int ne) {
    int bx, by;
    if (!fs_stream_tile(gx, ne, bx, by)) return;
    const FsEnvDev &E = envs[by];
    if (E.slot_env < 0) return;  // retired slot
    const int i = bx * FS_TILE + threadIdx.x;
    if (i < E.n) fs_iterate_particle_gridl_tp(E, shapes[by], i, sub, flip);
}

// (A CONTACT-SORTED form was built and measured in round 2 as well: springs in id order, then accumulator, particle and first
// candidate ids through LDS, the workgroup's 256 particles ordered by candidate-count class (ballots + a 4 x 4 table), thread t
// finishing the t-th particle of that order, so that the lanes of a wave carry equal numbers of contacts -- on oracle states of
// the bench workload the sum over a workgroup's waves of the largest candidate count drops from 17.2 to 7.4.  Bit-identical,
// and SLOWER: 64 episodes 1.129 -> 1.280 ms per step, one episode 0.856 -> 0.955, 256 episodes 3.425 -> 3.385.  Most candidate
// evaluations end at the wave-uniform `distance < solidRestDistance` test after ~20 instructions, so the candidate loop is a
// much smaller share of the 886 VALU instructions per wave than count x 74 suggests, and two more barriers plus an LDS hop in
// a kernel that lives on its critical path cost more than the divergence they remove.  Removed.)
// (A GRID-T form -- this kernel with the 256 + 4 dimx positions a workgroup's particles and their spring neighbours occupy
// staged in LDS by coalesced loads, 2.6 dwordx4 loads per thread instead of 13 gathers -- was built and measured in round 2:
// bit-identical and SLOWER at every launch size (64x64 x 64 episodes 1.480 -> 1.504 ms per step, x 8: 1.05 -> 1.17, x 256:
// 4.19 -> 4.38): the gathers hit the L1 / the XCD's L2 and are not what a launch waits for, while the staging adds a barrier
// and an LDS hop to the chain of dependent latencies that is.  Removed; EXPERIMENTS.md, round-2 notes 4.2.)

// The spring dictionary of the workgroup's episode -> LDS (one entry per thread; FS_TILE = 256 = dictionary size).
template <bool CODED>
__device__ __forceinline__ void fs_stage_sdict(const FsEnvDev &E, FsVec4 *sdict) {
    if (CODED) {
        const int t = threadIdx.x;
        sdict[t] = t < E.sdict_size ? E.sdict[t] : FsVec4{0.0f, 0.0f, 0.0f, 0.0f};
        __syncthreads();
    }
}

// throughput form (big launches): six springs in flight, later loads issued when needed (fewer live registers)
template <bool CODED>
__global__ __launch_bounds__(FS_TILE) void fs_k_iterate(const FsEnvDev *envs, const FsSlotSweeps *shapes, const int *ids,
                                                        int sub, int flip, int gx, int ne) {
    __shared__ FsVec4 sdict[CODED ? 256 : 1];
    int bx, by;
    if (!fs_stream_tile(gx, ne, bx, by)) return;
    const FsEnvDev &E = envs[by];
    const int e = E.slot_env;
    if (e < 0) return;  // retired slot
    const int i = bx * FS_TILE + threadIdx.x;
    if (bx * FS_TILE >= E.n) return;  // whole workgroup beyond this episode's particles
    fs_stage_sdict<CODED>(E, sdict);
    if (i < E.n) fs_iterate_particle<FS_STREAM_CHUNK, false, CODED>(E, shapes[by], i, sub, flip, sdict);
}
// latency form (small launches)
template <bool CODED>
__global__ __launch_bounds__(FS_TILE) void fs_k_iterate_eager(const FsEnvDev *envs, const FsSlotSweeps *shapes, const int *ids,
                                                              int sub, int flip, int gx, int ne) {
    __shared__ FsVec4 sdict[CODED ? 256 : 1];
    int bx, by;
    if (!fs_stream_tile(gx, ne, bx, by)) return;
    const FsEnvDev &E = envs[by];
    const int e = E.slot_env;
    if (e < 0) return;  // retired slot
    const int i = bx * FS_TILE + threadIdx.x;
    if (bx * FS_TILE >= E.n) return;
    fs_stage_sdict<CODED>(E, sdict);
    if (i < E.n) fs_iterate_particle<12, true, CODED>(E, shapes[by], i, sub, flip, sdict);
}

// ---- finalize: velocity from displacement, maxAcceleration / maxSpeed clamps (NvFlex.h:112-113), sleeping (:110)
__global__ __launch_bounds__(FS_TILE) void fs_k_finalize(const FsEnvDev *envs, const int *ids, int flip, int gx, int ne) {
    int bx, by;
    if (!fs_stream_tile(gx, ne, bx, by)) return;
    const FsEnvDev &E = envs[by];  // the slot's own copy of its episode's descriptor (fs_k_slot_table)
    if (E.slot_env < 0) return;    // retired slot
    const int i = bx * FS_TILE + threadIdx.x;
    if (i >= E.n) return;
    const FsParams &p = E.p;
    const float h = p.dt / (float)p.numSubsteps;
    const float inv_h = 1.0f / h;
    const FsVec4 x0 = fs_ld4o(E.x0, (unsigned)i);
    if (!(x0.w > 0.0f)) {
        fs_st4o(E.vel, (unsigned)i, FsVec4{0.0f, 0.0f, 0.0f, 0.0f});
        return;
    }
    const FsVec4 xp = fs_ld4o(flip ? E.xb : E.xa, (unsigned)i);
    const FsVec4 v0 = fs_ld4o(E.v0, (unsigned)i);
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
        fs_st4o(E.vel, (unsigned)i, FsVec4{0.0f, 0.0f, 0.0f, 0.0f});
    } else {
        fs_st4o(E.vel, (unsigned)i, FsVec4{vx, vy, vz, 0.0f});
        fs_st4o(E.pos, (unsigned)i, FsVec4{xp.x, xp.y, xp.z, x0.w});
    }
}
