// fs_fused_kernel.h -- fused LDS-resident solver: ONE workgroup advances ONE cloth episode by whole frames.
//
// MI355X mapping: a 64x64 cloth is 4096 particles; its current iterate (float4, 64 KiB), substep-start positions
// (3 x 16 KiB) and the spatial-hash bins (24 KiB) all fit the 160 KiB LDS of one CU, so the 4 substeps x 30 Jacobi
// iterations of a frame (reference softgym_cloth.h:154-155) run without touching HBM for particle positions:
// 1024 threads (16 waves, 4 per SIMD), 4 particles per thread walked in a ROLLED loop, neighbour positions gathered from
// LDS with ds_read_b128.  Grid = #episodes: 256 CUs advance 256 episodes concurrently.
//
// Two variants share all arithmetic (fs_constraints.h), so they and the CPU oracle agree bit for bit:
//   * compact  -- dictionary-coded adjacency: per particle 12 or 16 slots of {16-bit LDS offset of the neighbour, 16-bit
//                 LDS offset of a dictionary entry}, the dictionary (<= 256 distinct (rest length, stiffness) pairs; a
//                 grid cloth has nine) in LDS; the 12 packed words of a particle are streamed from L2 one particle
//                 ahead of their use (shared by every episode of the same cloth).
//   * generic  -- adjacency streamed from an L2-resident ELL table (any degree, any number of distinct springs).
// DESIGN.md 4.1 describes the stages (predict, hash, two-phase neighbour search, contact set, iterations); what was
// measured, tried and rejected is in EXPERIMENTS.md (round-2 notes, 4.1 / 4.15).
#pragma once
#include "fs_constraints.h"

#ifndef FS_FUSED_THREADS
#define FS_FUSED_THREADS 1024
#endif
#define FS_FUSED_PPT (4096 / FS_FUSED_THREADS)
#define FS_FUSED_MAX_PARTICLES (FS_FUSED_THREADS * FS_FUSED_PPT)
#define FS_FUSED_MAX_DEG 64
#ifndef FS_STENCIL_FILTER
#define FS_STENCIL_FILTER 1  // 0 (developer A/B): grid cloths keep the packed-id rest-near test (find mode 1 instead of 4)
#endif
#ifndef FS_FUSED_BUCKET_BITS
#define FS_FUSED_BUCKET_BITS 13
#endif
#define FS_FUSED_BUCKETS (1 << FS_FUSED_BUCKET_BITS)  // hashed cells (fs_fused_bucket: row hash + x).  The bucket table holds
                               // 16-BIT entries, two per LDS word: a count or a prefix sum never exceeds the 4096 particles of
                               // an episode, so 8192 buckets fit the 16 KiB that 4096 32-bit entries took -- a cell then
                               // shares its bucket with another cell half as often, and every aliased particle is a candidate
                               // the search has to test and reject (EXPERIMENTS.md R3.7)
#define FS_FUSED_CUR_BYTES (FS_FUSED_BUCKETS * 2)
#define FS_FUSED_SLOTS 16      // compact adjacency slots per particle

// LDS carve (bytes):
//   DICT float2[256]    distinct (rest length, stiffness) pairs of the cloth                    2 KiB
//   X    float4[4096]   current Jacobi iterate (xyz + invMass); search queues while the search runs   64 KiB
//   X0   float[3][4096] substep-start position (own displacement + neighbours' for friction);
//                       bucket-ordered predicted positions XS while the search runs                  48 KiB
//   HASH cursor u16[8192] | items u16[4096] | scan int[16]   (neighbour search only)           24 KiB
//   contact set: cset u16[CAP] | cacc float4[CAP] | chist int[128], CAP = min(1024, threads)   18.5 KiB
// The next iterate needs no LDS: each thread carries its four new positions in a rotating set of registers.
// (DICT and X sit below 64 KiB so their bases fold into the 16-bit offset field of the ds_read instructions; the gather
// addresses are then just the 16-bit halves of the packed adjacency words)
#define FS_FUSED_OFF_DICT 0
#define FS_FUSED_OFF_X (FS_FUSED_OFF_DICT + 256 * 8)
#define FS_FUSED_OFF_X0 (FS_FUSED_OFF_X + FS_FUSED_MAX_PARTICLES * 16)
#define FS_FUSED_OFF_CUR (FS_FUSED_OFF_X0 + FS_FUSED_MAX_PARTICLES * 12)
#define FS_FUSED_OFF_ITEMS (FS_FUSED_OFF_CUR + FS_FUSED_CUR_BYTES)
#define FS_FUSED_OFF_SCAN (FS_FUSED_OFF_ITEMS + FS_FUSED_MAX_PARTICLES * 2)
// contact set (rebuilt every substep, used by the iterations): ids of up to 1024 particles that have contact candidates,
// ordered by descending candidate count | their spring accumulators float4[1024] | count histogram / cursors int[128]
#define FS_FUSED_CSET_CAP (FS_FUSED_THREADS < 1024 ? FS_FUSED_THREADS : 1024)  // pass 2: thread e finishes cset[e]
#define FS_FUSED_OFF_CSET (FS_FUSED_OFF_SCAN + 64)
#define FS_FUSED_OFF_CACC (FS_FUSED_OFF_CSET + FS_FUSED_CSET_CAP * 2)
#define FS_FUSED_OFF_CHIST (FS_FUSED_OFF_CACC + FS_FUSED_CSET_CAP * 16)
#define FS_FUSED_LDS_BYTES (FS_FUSED_OFF_CHIST + 512)
#ifndef FS_FUSED_PREFETCH_CAND
#define FS_FUSED_PREFETCH_CAND 4  // contact candidates fetched ahead of the spring block
#endif
// The overflow queue (fs_k_fused_step; the grid-64 kernel has its own copy, see fs_fused_grid_kernel.h for the reasoning): a
// particle with up to 8 contact candidates that found no room in the contact set parks {spring sums, particle | first
// candidate << 12 | spring count << 24 | (candidates - 1) << 29} in the LDS the hash tables leave idle during the iterations,
// and the lanes of every wave but the first finish it in pass 2 instead of the main loop evaluating it inline.
#ifndef FS_FUSED_QUEUE
#define FS_FUSED_QUEUE 1
#endif
#define FS_FUSED_QUEUE_LANES (FS_FUSED_THREADS - 64)
#define FS_FUSED_QUEUE_ROOM ((FS_FUSED_CUR_BYTES + FS_FUSED_MAX_PARTICLES * 2) / 16)
#define FS_FUSED_QUEUE_CAP (FS_FUSED_QUEUE_ROOM < 2 * FS_FUSED_QUEUE_LANES ? FS_FUSED_QUEUE_ROOM : 2 * FS_FUSED_QUEUE_LANES)
static_assert(FS_FUSED_MAX_PARTICLES <= 4096, "queue words carry 12-bit particle ids");

#define FS_GLOBAL __attribute__((address_space(1)))
typedef FS_GLOBAL const int *fs_gci;
typedef FS_GLOBAL int *fs_gi;
typedef FS_GLOBAL const float *fs_gcf;
typedef FS_GLOBAL const uint32_t *fs_gcu;
typedef FS_GLOBAL const FsVec4 *fs_gcv4;
typedef FS_GLOBAL FsVec4 *fs_gv4;

// 16-byte global load/store through builtin vector types (address-space qualified structs cannot be copied in C++)
typedef float fs_f4 __attribute__((ext_vector_type(4)));
typedef float fs_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ FsVec4 fs_ld4(const FsVec4 *p, size_t i) {
    const fs_f4 v = ((FS_GLOBAL const fs_f4 *)p)[i];
    return FsVec4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void fs_st4(FsVec4 *p, size_t i, const FsVec4 v) {
    ((FS_GLOBAL fs_f4 *)p)[i] = fs_f4{v.x, v.y, v.z, v.w};
}

// the same with a 32-bit BYTE offset from a (uniform) base: global_load / global_store with a scalar base register and a
// 32-bit vector offset instead of a 64-bit vector address built with VALU adds
__device__ __forceinline__ FsVec4 fs_ld4o(const FsVec4 *base, unsigned idx) {
    const fs_f4 v = *(FS_GLOBAL const fs_f4 *)((FS_GLOBAL const char *)base + (idx << 4));
    return FsVec4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void fs_st4o(FsVec4 *base, unsigned idx, const FsVec4 v) {
    *(FS_GLOBAL fs_f4 *)((FS_GLOBAL char *)base + (idx << 4)) = fs_f4{v.x, v.y, v.z, v.w};
}
template <typename T>
__device__ __forceinline__ T fs_ldo(const T *base, unsigned idx) {
    return *(FS_GLOBAL const T *)((FS_GLOBAL const char *)base + (idx << 2));
}

// Cell -> bucket of the LDS hash.  The row (cy, cz) is hashed, x is added on top: the three cells cx-1..cx+1 a search
// visits per row are ADJACENT buckets, i.e. one contiguous run of the bucket-ordered arrays (9 runs per particle instead
// of 27 cells; longer runs also even out the per-lane trip counts of a wave).
__device__ __forceinline__ int fs_fused_row(int cy, int cz) {
    const unsigned h = (unsigned)cy * 0x85EBCA77u + (unsigned)cz * 0xC2B2AE3Du;
    return (int)((h ^ (h >> 15)) * 0x2C1B3C6Du >> (32 - FS_FUSED_BUCKET_BITS));
}
__device__ __forceinline__ int fs_fused_bucket(int cx, int cy, int cz) {
    return (fs_fused_row(cy, cz) + cx) & (FS_FUSED_BUCKETS - 1);
}

// Scalars of one episode, read once (uniform => SGPRs) so the hot loops never reload them through the descriptor.
struct FsFusedConsts {
    int n, substeps, iters, ncap, n_shapes, n_planes;
    float h, inv_h, g0, g1, g2, damping;
    float rad2, inv_rad, restd, restd2, mu_p, mu_s, mu_k, cd, relax, maxdv2, maxdv, thr2, max_speed;
    float pl0, pl1, pl2, pl3;
};

__device__ __forceinline__ FsFusedConsts fs_fused_consts(const FsEnvDev &E, const FsShapesDev &sh) {
    const FsParams &p = E.p;
    FsFusedConsts c;
    c.n = E.n; c.substeps = p.numSubsteps; c.iters = p.numIterations;
    c.ncap = p.maxNeighbors < FS_MAX_NEIGHBORS ? p.maxNeighbors : FS_MAX_NEIGHBORS;
    c.n_shapes = sh.count; c.n_planes = p.numPlanes;
    c.h = p.dt / (float)p.numSubsteps;
    c.inv_h = 1.0f / c.h;
    c.g0 = p.gravity[0]; c.g1 = p.gravity[1]; c.g2 = p.gravity[2]; c.damping = p.damping;
    const float rad = p.radius + p.particleCollisionMargin;
    c.rad2 = rad * rad; c.inv_rad = 1.0f / rad;
    c.restd = p.solidRestDistance; c.restd2 = c.restd * c.restd;
    c.mu_p = p.particleFriction; c.mu_s = p.staticFriction; c.mu_k = p.dynamicFriction;
    c.cd = p.collisionDistance; c.relax = p.relaxationFactor;
    c.maxdv = p.maxAcceleration * c.h; c.maxdv2 = c.maxdv * c.maxdv;
    c.thr2 = p.sleepThreshold * p.sleepThreshold; c.max_speed = p.maxSpeed;
    c.pl0 = p.planes[0][0]; c.pl1 = p.planes[0][1]; c.pl2 = p.planes[0][2]; c.pl3 = p.planes[0][3];
    return c;
}

// predict one particle (same arithmetic as fs_k_predict)
__device__ __forceinline__ FsVec4 fs_fused_predict(const FsFusedConsts &c, const FsVec4 pos, const FsVec4 vel) {
    FsVec4 xp = pos;
    if (pos.w > 0.0f) {
        float vx = vel.x + c.h * (c.g0 - c.damping * vel.x);
        float vy = vel.y + c.h * (c.g1 - c.damping * vel.y);
        float vz = vel.z + c.h * (c.g2 - c.damping * vel.z);
        xp.x = pos.x + c.h * vx;
        xp.y = pos.y + c.h * vy;
        xp.z = pos.z + c.h * vz;
    }
    return xp;
}

// finalize one particle (same arithmetic as fs_k_finalize)
__device__ __forceinline__ void fs_fused_finalize(const FsFusedConsts &c, FsVec4 &pos, FsVec4 &vel, const FsVec4 xp) {
    if (!(pos.w > 0.0f)) {
        vel = FsVec4{0.0f, 0.0f, 0.0f, 0.0f};
        return;
    }
    float vx = (xp.x - pos.x) * c.inv_h, vy = (xp.y - pos.y) * c.inv_h, vz = (xp.z - pos.z) * c.inv_h;
    float ax = vx - vel.x, ay = vy - vel.y, az = vz - vel.z;
    float dv2 = ax * ax + ay * ay + az * az;
    if (dv2 > c.maxdv2) {
        float sc = c.maxdv / sqrtf(dv2);
        vx = vel.x + ax * sc; vy = vel.y + ay * sc; vz = vel.z + az * sc;
    }
    float v2 = vx * vx + vy * vy + vz * vz;
    if (c.max_speed < 3.402823466e+38f && v2 > c.max_speed * c.max_speed) {
        float sc = c.max_speed / sqrtf(v2);
        vx = vx * sc; vy = vy * sc; vz = vz * sc;
        v2 = vx * vx + vy * vy + vz * vz;
    }
    if (v2 < c.thr2) {
        vel = FsVec4{0.0f, 0.0f, 0.0f, 0.0f};
    } else {
        vel = FsVec4{vx, vy, vz, 0.0f};
        pos.x = xp.x; pos.y = xp.y; pos.z = xp.z;
    }
}

// spatial hash of the predicted positions in LDS: histogram -> exclusive scan -> scatter.
// On return cursor[b] == end of bucket b (start == cursor[b-1]) and items[] holds particle ids grouped by bucket.
// Also writes XS: the predicted positions in bucket order (SoA, aliasing the X0 region, whose content is parked in
// global memory during the search), so the candidate scan reads positions sequentially instead of chasing ids.
__device__ __forceinline__ void fs_fused_build_grid(const FsFusedConsts &c, const FsVec4 (&xp)[FS_FUSED_PPT],
                                                    unsigned short *cursor, unsigned short *items, int *wave_tot, float *XSx,
                                                    float *XSy, float *XSz) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned *cw = (unsigned *)cursor;  // two buckets per word; an LDS atomic adds 1 to the bucket's half (no carry: <= 4096)
    int bucket[FS_FUSED_PPT];
#pragma unroll
    for (int k = 0; k < FS_FUSED_PPT; ++k) {
        const int i = t + k * FS_FUSED_THREADS;
        bucket[k] = fs_fused_bucket((int)floorf(xp[k].x * c.inv_rad), (int)floorf(xp[k].y * c.inv_rad),
                                    (int)floorf(xp[k].z * c.inv_rad));
        if (i < c.n) atomicAdd(&cw[bucket[k] >> 1], 1u << ((bucket[k] & 1) << 4));
    }
    __syncthreads();
    {
        constexpr int PW = FS_FUSED_BUCKETS / FS_FUSED_THREADS / 2;  // words per thread
        static_assert(PW >= 1 && PW * 2 * FS_FUSED_THREADS == FS_FUSED_BUCKETS, "bucket table: whole words per thread");
        unsigned wv[PW];
        int sum = 0;
#pragma unroll
        for (int k = 0; k < PW; ++k) { wv[k] = cw[t * PW + k]; sum += (int)(wv[k] & 0xffffu) + (int)(wv[k] >> 16); }
        int inc = sum;
        for (int off = 1; off < 64; off <<= 1) {
            int o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        int run = inc - sum;
        for (int w = 0; w < wave; ++w) run += wave_tot[w];  // <= 16 waves
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            const unsigned lo = (unsigned)run;
            run += (int)(wv[k] & 0xffffu);
            const unsigned hi = (unsigned)run;
            run += (int)(wv[k] >> 16);
            cw[t * PW + k] = lo | (hi << 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < FS_FUSED_PPT; ++k) {
        const int i = t + k * FS_FUSED_THREADS;
        if (i < c.n) {
            const unsigned sh = (unsigned)(bucket[k] & 1) << 4;
            const int slot = (int)((atomicAdd(&cw[bucket[k] >> 1], 1u << sh) >> sh) & 0xffffu);
            items[slot] = (unsigned short)i;
            XSx[slot] = xp[k].x; XSy[slot] = xp[k].y; XSz[slot] = xp[k].z;
        }
    }
    __syncthreads();
}

// particle-contact candidates of particle i (ascending id, the <= ncap smallest); list written slot-major to global
struct FsFindConsts {  // by value: a reference would force the caller's constants onto the stack
    int n, ncap;
    float rad2, inv_rad;
    int mode;  // 0: test phases and rest positions per pair (global loads); 1: uniform phase with SelfCollideFilter ->
               // membership test against the particle's packed rest-near ids; 2: uniform phase, no filter; 3: no pairs;
               // 4: like 1 on a canonical grid cloth whose rest-near sets are the 8 grid neighbours: two index differences
    int dimx;        // mode 4: grid width and ceil(2^32 / width) (row = umulhi(i, magic), verified by the host)
    unsigned magic;
};
struct FsNearWords {  // rest-near ids of one particle, by value (registers)
    uint32_t w[8];
};
 #ifdef FS_TIMING
__device__ unsigned long long fs_dbg_cnt[8];
#define FS_DBG_COUNT(k, v)                                                                                   \
    if (blockIdx.x == 0) {                                                                                   \
        const int l_ = threadIdx.x & 63;                                                                     \
        if (__builtin_amdgcn_readfirstlane(l_) == l_) atomicAdd(&fs_dbg_cnt[k], (unsigned long long)(v));   \
    }
#define FS_DBG_LANE(k, v) \
    if (blockIdx.x == 0) atomicAdd(&fs_dbg_cnt[k], (unsigned long long)(v));
#else
#define FS_DBG_COUNT(k, v)
#define FS_DBG_LANE(k, v)
#endif
#ifdef FS_TIMING_COUNTS  // trip / candidate counters inside the search loops (their atomics distort the timers)
#define FS_CNT_WAVE(k, v) FS_DBG_COUNT(k, v)
#define FS_CNT_LANE(k, v) FS_DBG_LANE(k, v)
#else
#define FS_CNT_WAVE(k, v)
#define FS_CNT_LANE(k, v)
#endif
#define FS_FUSED_FINDQ 32  // per-thread queue depth (u16 entries) of the two-phase neighbour search; the queue aliases X,
                           // which is dead during the search (the predicted positions sit in XS and in their owners' registers)

// The neighbour list of the particle being searched, while it is built: the FOUR smallest accepted ids live in
// registers (ascending, FS_NB_EMPTY padded), only what does not fit there goes to the global list (from slot FS_NB_STAGED
// on, every element larger than the staged ones).  How many are staged is the search's choice (the template parameter ST):
// the fused kernels stage 4 (measured in round 3, after the grid cloths' searches stopped carrying the packed rest-near ids:
// 4 -> 2.471, 6 -> 2.476, 8 -> 2.513 ms per launch of the bench), the streaming search 8.  A crumpling sheet has < 1 real contact per particle on average, so the
// dependent global read-modify-write chains of an in-memory insertion sort are gone from the common path; the staged
// ids are stored once at the end.
#define FS_NB_EMPTY 0x7fffffff
#ifndef FS_NB_STAGED_FUSED
#define FS_NB_STAGED_FUSED 4
#endif
#ifndef FS_NB_STAGED_STREAM
#define FS_NB_STAGED_STREAM 8
#endif
template <int ST>
struct FsNbListT {
    int a[ST];  // ascending; FS_NB_EMPTY = free (statically indexed only: stays in registers)
    int gcnt;   // elements in the global part (slots ST .. ST + gcnt - 1)
};
template <int ST>
__device__ __forceinline__ FsNbListT<ST> fs_nb_empty() {
    FsNbListT<ST> L;
#pragma unroll
    for (int q = 0; q < ST; ++q) L.a[q] = FS_NB_EMPTY;
    L.gcnt = 0;
    return L;
}

// sorted insertion into the global part: ascending ids, at most `gcap` kept; an id already present is not inserted again
template <int ST>
__device__ __forceinline__ void fs_fused_global_insert(int n, int i, int j, int gcap, int &gcnt, fs_gi nlist) {
    if (gcap <= 0) return;
    fs_gi list = nlist + (size_t)ST * n + i;
    int s = gcnt, prev = -1;
    while (s > 0) {
        prev = list[(size_t)(s - 1) * n];
        if (prev <= j) break;
        --s;
    }
    if (s > 0 && prev == j) return;
    if (gcnt == gcap) {
        if (s == gcap) return;
        gcnt = gcap - 1;
    }
    for (int u = gcnt; u > s; --u) list[(size_t)u * n] = list[(size_t)(u - 1) * n];
    list[(size_t)s * n] = j;
    ++gcnt;
}

// Second half of the search for one candidate j that passed the distance test: phase / rest-pose filter, then sorted,
// duplicate-free insertion (a particle can be met twice: its bucket may lie in two of the visited runs when rows alias).
// STENCIL: the caller knows c.mode == 4 (the packed ids in `near` are then never read and cost no registers).
template <bool STENCIL = false, int ST = FS_NB_STAGED_STREAM>
__device__ __forceinline__ void fs_fused_accept(const FsFindConsts &c, int i, int j, FsNbListT<ST> &L, int &phi, FsVec4 &ri,
                                                bool &have_meta, fs_gci phase, const FsVec4 *rest, fs_gi nlist,
                                                const FsNearWords &near) {
    if (STENCIL) {
        const int rj = (int)__umulhi((unsigned)j, c.magic), ri_ = (int)__umulhi((unsigned)i, c.magic);
        const int cj = j - rj * c.dimx, ci = i - ri_ * c.dimx;
        if ((unsigned)(rj - ri_ + 1) <= 2u && (unsigned)(cj - ci + 1) <= 2u) return;
    } else if (c.mode == 0) {  // general: phases and rest positions from global memory, per pair
        if (!have_meta) {
            phi = phase[i];
            ri = fs_ld4(rest, i);
            have_meta = true;
        }
        const FsVec4 rj = fs_ld4(rest, j);
        if (!fs_pair_allowed(phi, phase[j], ri, rj, c.rad2)) return;
    } else if (c.mode == 4) {  // grid cloth, one phase: "closer than the radius in the rest pose" = one of the 8 grid neighbours
        const int rj = (int)__umulhi((unsigned)j, c.magic), ri_ = (int)__umulhi((unsigned)i, c.magic);
        const int cj = j - rj * c.dimx, ci = i - ri_ * c.dimx;
        if ((unsigned)(rj - ri_ + 1) <= 2u && (unsigned)(cj - ci + 1) <= 2u) return;  // (j != i: the caller skipped it)
    } else if (c.mode == 1) {  // one phase for the whole cloth: the filter is a set-membership test on packed ids
        const uint32_t jj = (uint32_t)j | ((uint32_t)j << 16);
        uint32_t hit = 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t x = near.w[q] ^ jj;               // a zero 16-bit half <=> that slot holds j
            hit |= (x - 0x00010001u) & ~x & 0x80008000u;     // classic "has a zero half-word" test
        }
        if (hit) return;
    }
    bool dup = false;
#pragma unroll
    for (int q = 0; q < ST; ++q) dup |= (j == L.a[q]);
    if (dup) return;
    const int gcap = c.ncap - ST;
    if (j > L.a[ST - 1]) {  // staged part full and j beyond it (the last slot is FS_NB_EMPTY otherwise)
        fs_fused_global_insert<ST>(c.n, i, j, gcap, L.gcnt, nlist);
        return;
    }
    int t = j;
#pragma unroll
    for (int q = 0; q < ST; ++q) {
        const int lo = min(L.a[q], t);
        t = max(L.a[q], t);
        L.a[q] = lo;
    }
    if (t != FS_NB_EMPTY) fs_fused_global_insert<ST>(c.n, i, t, gcap, L.gcnt, nlist);  // displaced: larger than all staged
}

// stores the staged ids; returns the list length
template <int ST>
__device__ __forceinline__ int fs_fused_nb_finish(const FsFindConsts &c, int i, const FsNbListT<ST> &L, fs_gi nlist) {
    int st = 0;
    const size_t n = (size_t)c.n;
#pragma unroll
    for (int q = 0; q < ST; ++q) {
        if (L.a[q] != FS_NB_EMPTY) { nlist[(size_t)q * n + i] = L.a[q]; ++st; }  // ascending: the used slots are a prefix
    }
    const int total = st + L.gcnt;
    return total < c.ncap ? total : c.ncap;
}

// Two-phase search.  Lanes hit their few real neighbours at different trips of the candidate loop, so with the accept
// code inside that loop nearly every trip drags the whole wave through it (a wave of 64 lanes has a distance hit in
// almost every trip: the 8 mesh neighbours of every cloth particle are always in range).
//   Phase A scans the runs FOUR candidates per trip from the bucket-ordered position copy XS (three aligned 16-byte
//   LDS reads) and does nothing but the distance test; a trip with hits parks ONE packed entry (run slot | 4-bit hit
//   mask) in a per-thread LDS queue.
//   Phase B walks the queue, where all lanes have work at the same time: id lookup, self / rest-pose filter, sorted
//   insertion.
typedef __attribute__((address_space(3))) const int *fs_lci;
typedef __attribute__((address_space(3))) const float *fs_lcf;
typedef __attribute__((address_space(3))) const fs_f4 *fs_lcf4;
typedef __attribute__((address_space(3))) const unsigned short *fs_lcus;
typedef __attribute__((address_space(3))) unsigned short *fs_lus;

template <bool STENCIL = false>
__device__ __forceinline__ void fs_fused_drain(const FsFindConsts &c, int i, int &qn, FsNbListT<FS_NB_STAGED_FUSED> &L, int &phi, FsVec4 &ri,
                                               bool &have_meta, fs_lcus items, fs_lus queue, fs_gci phase,
                                               const FsVec4 *rest, fs_gi nlist, const FsNearWords &near) {
    // one survivor per lane and trip: the next packed entry is fetched in the trip that finds the current one used up,
    // so the trip count is the largest SURVIVOR count of the wave, not (most entries) x (most hits per entry)
    int e = 0, qb = 0;
    unsigned m = 0u;
    for (;;) {
        if (!m) {
            if (e == qn) break;
            const unsigned u = queue[e * FS_FUSED_THREADS];
            ++e;
            m = u & 15u;
            qb = (int)(u >> 4) << 2;
        }
        FS_CNT_WAVE(5, 1)  // wave-level phase-B candidate trips
        const int j = items[qb + __builtin_ctz(m)];
        m &= m - 1u;
        if (j == i) continue;
        FS_CNT_LANE(3, 1)  // lane-level survivors
        fs_fused_accept<STENCIL, FS_NB_STAGED_FUSED>(c, i, j, L, phi, ri, have_meta, phase, rest, nlist, near);
    }
    qn = 0;
}

template <bool STENCIL = false>
__device__ __noinline__ int fs_fused_find_neighbors(const FsFindConsts c, int i, const FsVec4 xi, fs_lcus cursor,
                                                       fs_lcus items, fs_gci phase, const FsVec4 *rest, fs_gi nlist,
                                                       const FsNearWords near,
                                                       fs_lus queue /* [FINDQ][blockDim] + threadIdx */, fs_lcf XSx) {
    fs_lcf XSy = XSx + FS_FUSED_MAX_PARTICLES, XSz = XSy + FS_FUSED_MAX_PARTICLES;
    const int cx = (int)floorf(xi.x * c.inv_rad), cy = (int)floorf(xi.y * c.inv_rad), cz = (int)floorf(xi.z * c.inv_rad);
    int phi = 0, qn = 0;
    FsNbListT<FS_NB_STAGED_FUSED> L = fs_nb_empty<FS_NB_STAGED_FUSED>();
    FsVec4 ri = FsVec4{0.0f, 0.0f, 0.0f, 0.0f};
    bool have_meta = false;
#ifdef FS_TIMING
    const unsigned long long tf0 = __builtin_amdgcn_s_memtime();
#endif
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy) {
            // buckets b0, b0+1, b0+2 (mod BUCKETS) hold the cells (cx-1..cx+1, cy+dy, cz+dz): one run, or two when the
            // triple wraps around the end of the table.  Whatever else hashes into the run is rejected by distance.
            const int b0 = (fs_fused_row(cy + dy, cz + dz) + cx - 1) & (FS_FUSED_BUCKETS - 1);
            const int wrap = b0 + 2 - (FS_FUSED_BUCKETS - 1);  // > 0: that many buckets continue at bucket 0
            for (int seg = 0; seg < 2; ++seg) {
                int beg, end;
                if (seg == 0) {
                    beg = (b0 == 0) ? 0 : (int)cursor[b0 - 1];
                    end = (int)cursor[wrap > 0 ? FS_FUSED_BUCKETS - 1 : b0 + 2];
                } else {
                    if (wrap <= 0) break;
                    beg = 0;
                    end = (int)cursor[wrap - 1];
                }
                FS_CNT_WAVE(0, 1)          // wave-level run visits
                FS_CNT_LANE(2, end - beg)  // lane-level candidates
                for (int q = beg & ~3; q < end; q += 4) {
                    FS_CNT_WAVE(1, 1)      // wave-level candidate trips
                    const fs_f4 ax = *(fs_lcf4)(XSx + q), ay = *(fs_lcf4)(XSy + q), az = *(fs_lcf4)(XSz + q);
                    const fs_f4 ex = xi.x - ax, ey = xi.y - ay, ez = xi.z - az;
                    const fs_f4 d2 = ex * ex + ey * ey + ez * ez;
                    unsigned m = (unsigned)(d2.x < c.rad2) | ((unsigned)(d2.y < c.rad2) << 1) |
                                 ((unsigned)(d2.z < c.rad2) << 2) | ((unsigned)(d2.w < c.rad2) << 3);
                    // slots of this group that belong to the run: [beg - q, end - q) clipped to [0, 4)
                    const int lo = beg - q > 0 ? beg - q : 0, hi = end - q < 4 ? end - q : 4;
                    m &= ((1u << (hi - lo)) - 1u) << lo;
                    if (m) {
                        if (qn == FS_FUSED_FINDQ)  // queue full (dense crumple): work it off first
                            fs_fused_drain<STENCIL>(c, i, qn, L, phi, ri, have_meta, items, queue, phase, rest, nlist, near);
                        queue[qn * FS_FUSED_THREADS] = (unsigned short)(((unsigned)q << 2) | m);
                        ++qn;
                    }
                }
            }
        }
#ifdef FS_TIMING
    const unsigned long long tf1 = __builtin_amdgcn_s_memtime();
    FS_DBG_COUNT(6, tf1 - tf0)
#endif
    fs_fused_drain<STENCIL>(c, i, qn, L, phi, ri, have_meta, items, queue, phase, rest, nlist, near);
#ifdef FS_TIMING
    FS_DBG_COUNT(7, __builtin_amdgcn_s_memtime() - tf1)
#endif
    return fs_fused_nb_finish(c, i, L, nlist);
}


// Exclusive offsets of the contact-count histogram in DESCENDING count order: chist[q] <- number of particles with more than q
// candidates (q = 1..127), chist[0] <- number of particles with any.  One wavefront, two bins per lane, a suffix scan by
// shuffles (it was a 96-trip loop of one thread behind a workgroup barrier, every substep).  chist[0] must be 0 on entry.
__device__ __forceinline__ void fs_fused_count_offsets(int *chist, int t) {
    if (t >= 64) return;
    const int h0 = chist[2 * t], h1 = chist[2 * t + 1];
    const int pair = h0 + h1;
    int inc = pair;  // inclusive suffix sum over the lanes
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_down(inc, off, 64);
        if (t + off < 64) inc += o;
    }
    const int above = inc - pair;  // particles in bins > 2 t + 1
    chist[2 * t + 1] = above;
    chist[2 * t] = above + h1;     // (lane 0: bin 0 holds no particles, so this is the total)
}

// planes + kinematic spheres for one particle
// planes + spheres for one particle of the fused kernels.  `mask`: the particle's collideShapes candidates of this substep
// (fs_shape_candidates).  The plane test is per lane (a sheet on the ground lists it for every particle: a wave-level skip would
// only add instructions there); the sphere block is skipped by every wavefront none of whose lanes lists a sphere -- all of
// them while the pickers are parked -- and a sphere's sweep and test run only where a lane lists it.
// (Measured and dropped, EXPERIMENTS R5.1: the substep's sweeps staged in an LDS table instead of recomputed per candidate, and
// per-row wave-uniform hints -- "every particle of this row lists the plane / none lists a sphere" -- worked out once per
// substep so that the common cases cost scalar branches only: bit-identical, and 1.5 % / 3.8 % SLOWER on the bench; both lengthen
// live ranges in a kernel that already spills.)
__device__ __forceinline__ void fs_fused_shape_contacts(FsAcc &a, const FsFusedConsts &c, const FsParams &p,
                                                        const FsShapesDev &sh, int sub, unsigned mask, float xi0, float xi1,
                                                        float xi2, float ri0, float ri1, float ri2) {
    if (c.n_planes == 1) {
        if (mask & 1u) fs_plane_contact(a, xi0, xi1, xi2, ri0, ri1, ri2, c.pl0, c.pl1, c.pl2, c.pl3, c.cd, c.mu_s, c.mu_k);
    } else {
        for (int q = 0; q < c.n_planes; ++q)
            if ((mask >> q) & 1u)
                fs_plane_contact(a, xi0, xi1, xi2, ri0, ri1, ri2, p.planes[q][0], p.planes[q][1], p.planes[q][2],
                                 p.planes[q][3], c.cd, c.mu_s, c.mu_k);
    }
    if (c.n_shapes == 0 || __builtin_amdgcn_ballot_w64((mask >> FS_SHAPE_SPHERE_BIT) != 0u) == 0ull) return;
    const float S = (float)c.substeps;
    for (int q = 0; q < c.n_shapes; ++q)
        if ((mask >> (FS_SHAPE_SPHERE_BIT + q)) & 1u) {
            float c0, c1, c2, s0, s1, s2;
            fs_shape_sweep(sh, q, sub, S, c0, c1, c2, s0, s1, s2);
            fs_sphere_contact(a, xi0, xi1, xi2, ri0, ri1, ri2, c0, c1, c2, sh.pos[q].w, s0, s1, s2, c.cd, c.mu_s, c.mu_k);
        }
}

// The spring sweep of one particle over the packed adjacency: one spring per scheduling region; the LDS gather and
// dictionary fetch of spring s+1 are issued in front of the arithmetic of spring s.
//   FAST: the wave's neighbours all carry their particle's own mass and the cloth has no tethers -> fs_spring_fast on
//         the pre-halved stiffness; otherwise fs_spring_bf on the stiffness times `kscale` (2 when the LDS dictionary
//         holds halves, else 1 -- both exact).
template <int SLOTS, bool FAST>
__device__ __forceinline__ void fs_fused_spring_block(FsAcc &a, const char *smem, const uint32_t (&jw)[SLOTS / 2 > 0 ? SLOTS / 2 : 1],
                                                      const uint32_t (&cw)[SLOTS / 2 > 0 ? SLOTS / 2 : 1], float xi0, float xi1,
                                                      float xi2, float wi, float kscale) {
    FsVec4 xj = *(const FsVec4 *)(smem + FS_FUSED_OFF_X + (jw[0] & 0xffffu));
    float2 lk = *(const float2 *)(smem + FS_FUSED_OFF_DICT + (cw[0] & 0xffffu));
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        FsVec4 xj_next = xj;
        float2 lk_next = lk;
        if (s + 1 < SLOTS) {
            const uint32_t joff = (jw[(s + 1) >> 1] >> (16 * ((s + 1) & 1))) & 0xffffu;
            const uint32_t coff = (cw[(s + 1) >> 1] >> (16 * ((s + 1) & 1))) & 0xffffu;
            xj_next = *(const FsVec4 *)(smem + FS_FUSED_OFF_X + joff);
            lk_next = *(const float2 *)(smem + FS_FUSED_OFF_DICT + coff);
        }
        if (FAST) fs_spring_fast(a, xi0, xi1, xi2, xj, lk.x, lk.y);
        else fs_spring_bf(a, xi0, xi1, xi2, wi, xj, lk.x, lk.y * kscale);
        __builtin_amdgcn_sched_barrier(0);
        xj = xj_next;
        lk = lk_next;
    }
}

// word of the 16-bit slot pack that holds particle k of the thread (k is a run-time value in the rolled particle loop)
#if FS_FUSED_PPT <= 4
#define FS_SLOTWORD(sp, k) ((sp)[0])
#else
#define FS_SLOTWORD(sp, k) (((k) >> 2) ? (sp)[1] : (sp)[0])
#endif

// SLOTS > 0: packed (dictionary-coded) adjacency with that many slots (12 or 16); SLOTS == 0: plain ELL adjacency.
//
// Register discipline (1024 threads => 128 VGPRs, and every attempt to keep per-particle state of the thread's four
// particles in registers made hipcc interleave four inlined copies of the body and spill): the particle loop is a
// ROLLED loop and nothing per-particle survives it in registers except the four new positions, which rotate through a
// statically indexed register file (r[q] <- r[q-1]) and are published to X after the barrier; the current iterate is X
// (LDS); the velocity lives in global memory (touched twice per substep); the
// packed adjacency (2 x SLOTS/2 dwords per particle, shared by all episodes of the same cloth, L2/L1 resident) and the
// head of the contact-candidate list are fetched one particle ahead of their use.
template <int SLOTS>
__global__ __launch_bounds__(FS_FUSED_THREADS) void fs_k_fused_step(const FsEnvDev *__restrict__ envs, const FsShapesDev *__restrict__ shapes,
                                                                    const int *ids, int n_steps) {
    constexpr bool COMPACT = SLOTS > 0;
    constexpr int JW = COMPACT ? SLOTS / 2 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    FsVec4 *X = (FsVec4 *)(smem + FS_FUSED_OFF_X);
    float *X0x = (float *)(smem + FS_FUSED_OFF_X0);
    float *X0y = X0x + FS_FUSED_MAX_PARTICLES;
    float *X0z = X0y + FS_FUSED_MAX_PARTICLES;
    unsigned short *cursor = (unsigned short *)(smem + FS_FUSED_OFF_CUR);
    unsigned short *items = (unsigned short *)(smem + FS_FUSED_OFF_ITEMS);
    int *wave_tot = (int *)(smem + FS_FUSED_OFF_SCAN);

#ifdef FS_TIMING  // developer build: per-section shader-clock totals of block 0's waves, printed at the end
    unsigned long long ts_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ts_last = __builtin_amdgcn_s_memtime();
#define FS_TS(k)                                                     \
    {                                                                \
        const unsigned long long now = __builtin_amdgcn_s_memtime(); \
        ts_acc[k] += now - ts_last;                                  \
        ts_last = now;                                               \
    }
#else
#define FS_TS(k)
#endif
    const int e = ids[blockIdx.x];
    if (e < 0) return;  // slot retired by a device-side loop (fs_wait_until_stable)
    const FsEnvDev &E = envs[e];
    const FsShapesDev &sh = shapes[e];
    const FsFusedConsts c = fs_fused_consts(E, sh);
    const int n = c.n;
    const int kmax = (n + FS_FUSED_THREADS - 1) / FS_FUSED_THREADS;  // particle slots per thread that hold a particle for some thread
    const unsigned un = (unsigned)n;
    const int t = threadIdx.x;

    FsVec4 *const g_pos = E.pos, *const g_vel = E.vel;
    const FsVec4 *const g_rest = E.rest;
    const fs_gci g_phase = (fs_gci)E.phase;
    const fs_gi g_nlist = (fs_gi)E.nlist, g_ncount = (fs_gi)E.ncount;
    const fs_gci g_ell_j = (fs_gci)E.ell_j;
    const fs_gcf g_ell_len = (fs_gcf)E.ell_len, g_ell_k = (fs_gcf)E.ell_k;
    const fs_gcu g_nbr = (fs_gcu)E.nbr_w, g_code = (fs_gcu)E.code_w;
    const int max_deg = E.max_deg;

    // Spring dictionary -> LDS.  When no entry is a tether (k < 0) and every stiffness halves exactly, LDS holds
    // (L, k / 2) and the waves whose particles see only neighbours of their own mass take the short spring form.
    float2 dict_e = make_float2(0.0f, 0.0f);
    bool dict_bad = false;
    if (COMPACT && t < 256) {
        const fs_gcf g_dict = (fs_gcf)E.dict;
        dict_e = make_float2(g_dict[2 * t], g_dict[2 * t + 1]);
        dict_bad = t < E.dict_size && (dict_e.y < 0.0f || (dict_e.y * 0.5f) * 2.0f != dict_e.y);
    }
    // Is the whole cloth one phase?  Then the per-pair phase / rest-position loads of the neighbour search (dependent
    // global gathers) collapse into a register test against the packed rest-near ids.
    const fs_gcu g_near = (fs_gcu)E.restnear_w;
    int find_mode = 0;
    bool dict_halved = false;
    {
        const int ph0 = g_phase[0];
        int differs = 0;
        for (int i = t; i < n; i += FS_FUSED_THREADS) differs |= (g_phase[i] != ph0);
        if (t == 0) { wave_tot[0] = 0; wave_tot[1] = 0; }
        __syncthreads();
        if (differs) atomicOr(&wave_tot[0], 1);
        if (dict_bad) atomicOr(&wave_tot[1], 1);
        __syncthreads();
        const int mixed = wave_tot[0];
        dict_halved = COMPACT && wave_tot[1] == 0;
        __syncthreads();
        if (!mixed) {
            if (!(ph0 & FS_PHASE_SELF_COLLIDE)) find_mode = 3;  // same group, no self-collision flag: no pairs at all
            else if (!(ph0 & FS_PHASE_SELF_COLLIDE_FILTER)) find_mode = 2;
            else if (E.restnear_ok == 2 && FS_STENCIL_FILTER) find_mode = 4;
            else if (E.restnear_ok) find_mode = 1;
        }
    }
    if (COMPACT && t < 256)
        *(float2 *)(smem + FS_FUSED_OFF_DICT + t * 8) = make_float2(dict_e.x, dict_halved ? dict_e.y * 0.5f : dict_e.y);
    const float kscale = dict_halved ? 2.0f : 1.0f;
    // own particles: i = t + k * 1024.  Load positions into X (w = invMass) and X0.
    for (int i = t; i < n; i += FS_FUSED_THREADS) {
        const FsVec4 p = fs_ld4(g_pos, i);
        X[i] = p;
        X0x[i] = p.x; X0y[i] = p.y; X0z[i] = p.z;
    }
    // bit k: every lane's particle k of this wave has only spring neighbours of its own inverse mass (constant for the
    // launch: masses change between launches only) -> short spring form
    unsigned fastmask = 0u;
    if (COMPACT && dict_halved) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < FS_FUSED_PPT; ++k) {
            const int i = t + k * FS_FUSED_THREADS;
            bool differs = false;
            if (i < n) {
                const float wi = X[i].w;
                if (wi > 0.0f) {
#pragma unroll
                    for (int q = 0; q < JW; ++q) {
                        const uint32_t w = g_nbr[(unsigned)q * un + (unsigned)i];
                        differs |= *(const float *)(smem + FS_FUSED_OFF_X + (w & 0xffffu) + 12) != wi;
                        differs |= *(const float *)(smem + FS_FUSED_OFF_X + (w >> 16) + 12) != wi;
                    }
                }
            }
            if (__builtin_amdgcn_ballot_w64(differs) == 0ull) fastmask |= 1u << k;
        }
    }

    FS_TS(0)
#pragma unroll 1
    for (int frame = 0; frame < n_steps; ++frame) {
#pragma unroll 1
        for (int sub = 0; sub < c.substeps; ++sub) {
            // ---- predict from (X0, vel) into X; build the spatial hash
            for (int q = t; q < FS_FUSED_BUCKETS / 2; q += FS_FUSED_THREADS) ((unsigned *)cursor)[q] = 0u;
            FsVec4 xp[FS_FUSED_PPT];
#pragma unroll
            for (int k = 0; k < FS_FUSED_PPT; ++k) {
                const int i = t + k * FS_FUSED_THREADS;
                xp[k] = FsVec4{0.0f, 0.0f, 0.0f, 0.0f};
                if (i < n) {
                    const FsVec4 p0 = FsVec4{X0x[i], X0y[i], X0z[i], X[i].w};
                    xp[k] = fs_fused_predict(c, p0, fs_ld4(g_vel, i));
                    fs_st4(E.x0, i, p0);  // X0 is parked in global memory: its LDS holds the bucket-ordered positions
                }                         // (XS) until the search is over
            }
            __syncthreads();
            fs_fused_build_grid(c, xp, cursor, items, wave_tot, X0x, X0y, X0z);
            FS_TS(1)
            const FsFindConsts fc = {n, c.ncap, c.rad2, c.inv_rad, find_mode, E.gp_dimx, E.gp_magic};
            // Particles are searched in BUCKET order (lane <-> slot of the sorted copy): the lanes of one cell walk the
            // same 27 buckets with the same trip counts and read the same LDS words (broadcast, no bank conflicts);
            // in id order every lane of a wave walked different buckets and the scan was LDS-conflict bound.
#pragma unroll 1
            for (int qs = t; qs < n; qs += FS_FUSED_THREADS) {
                const int i = items[qs];
                const FsVec4 xi = FsVec4{X0x[qs], X0y[qs], X0z[qs], 0.0f};  // = XS[qs], the predicted position of i
                // collideShapes rides along: the particle's shape candidates go into the upper bits of its count word
                const int shape_bits = (int)(fs_shape_candidates(E.p, sh, sub, xi.x, xi.y, xi.z) << FS_SHAPE_MASK_SHIFT);
                if (find_mode == 4) {  // grid cloth: no packed rest-near ids to carry through the search
                    FsNearWords none;
#pragma unroll
                    for (int q = 0; q < 8; ++q) none.w[q] = 0xffffffffu;
                    g_ncount[i] = fs_fused_find_neighbors<true>(fc, i, xi, (fs_lcus)cursor, (fs_lcus)items, g_phase, g_rest, g_nlist,
                                                                none, (fs_lus)(smem + FS_FUSED_OFF_X) + t, (fs_lcf)X0x) | shape_bits;
                    continue;
                }
                FsNearWords near;
#pragma unroll
                for (int q = 0; q < 8; ++q) near.w[q] = find_mode == 1 ? g_near[(unsigned)q * un + (unsigned)i] : 0xffffffffu;
                g_ncount[i] = (find_mode == 3 ? 0
                                              : fs_fused_find_neighbors<false>(fc, i, xi, (fs_lcus)cursor, (fs_lcus)items, g_phase,
                                                                               g_rest, g_nlist, near,
                                                                               (fs_lus)(smem + FS_FUSED_OFF_X) + t, (fs_lcf)X0x)) |
                              shape_bits;
            }
            FS_TS(2)
            __syncthreads();  // every wave is done with XS and the hash
            FS_TS(3)
#pragma unroll
            for (int k = 0; k < FS_FUSED_PPT; ++k) {  // the predicted positions become the first iterate; bring X0 back
                const int i = t + k * FS_FUSED_THREADS;
                if (i < n) {
                    X[i] = xp[k];
                    const FsVec4 p0 = fs_ld4(E.x0, i);
                    X0x[i] = p0.x; X0y[i] = p0.y; X0z[i] = p0.z;
                }
            }

            // ---- contact set.  Candidate counts are very uneven (crumpling sheet: mean 0.8, max ~10 per particle), and a
            // wave walking its lanes' own lists stays in the ~110-instruction contact body for max(count) rounds with a
            // few live lanes.  So the particles that have candidates (typically a quarter) are collected, ordered by
            // descending count, into a compact set; in every iteration their owners only evaluate the springs and park
            // the accumulator in LDS (pass 1), then thread e finishes particle cset[e] -- contacts, plane, spheres,
            // applyDeltas -- next to 63 lanes with the same amount of work (pass 2).  The per-particle accumulation
            // order (springs, contacts ascending, planes, spheres) is unchanged, so results stay bit-identical.
            unsigned short *cset = (unsigned short *)(smem + FS_FUSED_OFF_CSET);
            FsVec4 *cacc = (FsVec4 *)(smem + FS_FUSED_OFF_CACC);
            int *chist = (int *)(smem + FS_FUSED_OFF_CHIST);
            if (t < 128) chist[t] = 0;
            int *nqueued = (int *)(smem + FS_FUSED_OFF_SCAN);  // (the hash build's scratch: idle)
            if (t == 0) *nqueued = 0;
            __syncthreads();
            int ccls[FS_FUSED_PPT];
#pragma unroll
            for (int k = 0; k < FS_FUSED_PPT; ++k) {
                const int i = t + k * FS_FUSED_THREADS;
                int cc = 0;
                if (i < n && X[i].w > 0.0f) cc = g_ncount[i] & FS_NCOUNT_MASK;
                ccls[k] = cc > 96 ? 96 : cc;
                if (ccls[k] > 0) atomicAdd(&chist[ccls[k]], 1);
            }
            __syncthreads();
            fs_fused_count_offsets(chist, t);  // exclusive offsets in descending count order; chist[0] <- total
            __syncthreads();
            // 16-bit set slot of each of the thread's particles (four per word), 0xffff = not in the set
            unsigned long long slotpack[(FS_FUSED_PPT + 3) / 4];
#pragma unroll
            for (int q = 0; q < (FS_FUSED_PPT + 3) / 4; ++q) slotpack[q] = ~0ull;
#pragma unroll
            for (int k = 0; k < FS_FUSED_PPT; ++k) {
                if (ccls[k] > 0) {
                    const int pos = atomicAdd(&chist[ccls[k]], 1);
                    if (pos < FS_FUSED_CSET_CAP) {
                        cset[pos] = (unsigned short)(t + k * FS_FUSED_THREADS);
                        slotpack[k >> 2] = (slotpack[k >> 2] & ~(0xffffull << (16 * (k & 3)))) |
                                           ((unsigned long long)pos << (16 * (k & 3)));
                    }
#if FS_FUSED_QUEUE
                    else if (ccls[k] <= 8) {  // slot value CAP + place in the overflow queue
                        const int sp = atomicAdd(nqueued, 1);
                        if (sp < FS_FUSED_QUEUE_CAP)
                            slotpack[k >> 2] = (slotpack[k >> 2] & ~(0xffffull << (16 * (k & 3)))) |
                                               ((unsigned long long)(FS_FUSED_CSET_CAP + sp) << (16 * (k & 3)));
                    }
#endif
                }
            }
            __syncthreads();
            const int csize = chist[0] < FS_FUSED_CSET_CAP ? chist[0] : FS_FUSED_CSET_CAP;
            const int n_queued = *nqueued < FS_FUSED_QUEUE_CAP ? *nqueued : FS_FUSED_QUEUE_CAP;
            FsVec4 *squeue = (FsVec4 *)(smem + FS_FUSED_OFF_CUR);
            // the set particle this thread finishes in pass 2, with its candidate count and list head (constant over
            // the substep's iterations)
            const int i2 = t < csize ? (int)cset[t] : -1;
            int cnt2 = 0, cj2[FS_FUSED_PREFETCH_CAND];
            unsigned smask2 = 0u;
#pragma unroll
            for (int q = 0; q < FS_FUSED_PREFETCH_CAND; ++q) cj2[q] = 0;
            if (i2 >= 0) {
                const int word = g_ncount[i2];
                cnt2 = word & FS_NCOUNT_MASK;
                smask2 = (unsigned)word >> FS_SHAPE_MASK_SHIFT;
#pragma unroll
                for (int q = 0; q < FS_FUSED_PREFETCH_CAND; ++q) cj2[q] = g_nlist[(unsigned)q * un + (unsigned)i2];
            }

            FS_TS(4)
            // ---- Jacobi iterations: gather from X, new positions in rotating registers, barrier, publish, barrier
#pragma unroll 1
            for (int it = 0; it < c.iters; ++it) {
                // software pipeline over the thread's particles: adjacency words / candidate head of particle k+1 are
                // requested before particle k is computed
                uint32_t jw[JW], cw[JW];
                int cntw = 0, cj[FS_FUSED_PREFETCH_CAND];  // cntw: candidate count | shape candidates << 8
                {
                    unsigned i0 = t < n ? (unsigned)t : 0u;
                    asm volatile("" : "+v"(i0));  // keep these iteration-invariant loads inside the loop
                    if (COMPACT) {
#pragma unroll
                        for (int q = 0; q < JW; ++q) { jw[q] = g_nbr[(unsigned)q * un + i0]; cw[q] = g_code[(unsigned)q * un + i0]; }
                    }
                    cntw = g_ncount[i0];
#pragma unroll
                    for (int q = 0; q < FS_FUSED_PREFETCH_CAND; ++q) cj[q] = g_nlist[(unsigned)q * un + i0];
                }
                // new positions of the thread's particles: a rotating register file (r3 <- r2 <- r1 <- r0 <- new) so the
                // ROLLED particle loop needs no dynamically indexed registers and no LDS staging
                float rx[FS_FUSED_PPT], ry[FS_FUSED_PPT], rz[FS_FUSED_PPT];  // statically indexed only
#pragma unroll
                for (int q = 0; q < FS_FUSED_PPT; ++q) { rx[q] = 0.0f; ry[q] = 0.0f; rz[q] = 0.0f; }
#pragma unroll 1
                for (int k = 0; k < FS_FUSED_PPT; ++k) {
                    if (k >= kmax) {  // a particle slot no thread has (cloths of <= 3072 particles): rotate, compute nothing
#pragma unroll
                        for (int q = FS_FUSED_PPT - 1; q > 0; --q) { rx[q] = rx[q - 1]; ry[q] = ry[q - 1]; rz[q] = rz[q - 1]; }
                        continue;
                    }
                    const int i_raw = t + k * FS_FUSED_THREADS;
                    const int i = i_raw < n ? i_raw : 0;  // lanes past the end recompute particle 0 and discard it
                    uint32_t jw_n[JW], cw_n[JW];
                    int cntw_n, cj_n[FS_FUSED_PREFETCH_CAND];
                    const int cnt = cntw & FS_NCOUNT_MASK;
                    const unsigned smask = (unsigned)cntw >> FS_SHAPE_MASK_SHIFT;
                    const FsVec4 xi = X[i];
                    float nx = xi.x, ny = xi.y, nz = xi.z;
                    FsAcc a = {0.0f, 0.0f, 0.0f, 0};
                    const float xi0 = xi.x, xi1 = xi.y, xi2 = xi.z, wi = xi.w;
                    if (xi.w > 0.0f) {
                        if (COMPACT) {
                            if ((fastmask >> k) & 1u)
                                fs_fused_spring_block<SLOTS, true>(a, smem, jw, cw, xi0, xi1, xi2, wi, 1.0f);
                            else
                                fs_fused_spring_block<SLOTS, false>(a, smem, jw, cw, xi0, xi1, xi2, wi, kscale);
                        } else {
                            for (int s = 0; s < max_deg; ++s) {
                                const int j = g_ell_j[(unsigned)s * un + (unsigned)i];
                                if (j < 0) break;
                                fs_spring(a, xi0, xi1, xi2, wi, X[j], g_ell_len[(unsigned)s * un + (unsigned)i],
                                          g_ell_k[(unsigned)s * un + (unsigned)i]);
                            }
                        }
                    }
                    FS_TS(5)
                    // Prefetch for the thread's NEXT particle, issued here -- after the spring block, whose working set is
                    // dead by now -- and consumed at the top of the next trip: the contact / shape / apply section plus
                    // the other three waves of the SIMD cover the L2 latency.  (Issued at the top of the body the 17
                    // in-flight registers overlapped the spring block's peak pressure and were spilled to scratch.)
                    {
                        unsigned in = i_raw + FS_FUSED_THREADS < n ? (unsigned)(i_raw + FS_FUSED_THREADS) : 0u;
                        if (COMPACT) {
#pragma unroll
                            for (int q = 0; q < JW; ++q) { jw_n[q] = g_nbr[(unsigned)q * un + in]; cw_n[q] = g_code[(unsigned)q * un + in]; }
                        }
                        cntw_n = g_ncount[in];
#pragma unroll
                        for (int q = 0; q < FS_FUSED_PREFETCH_CAND; ++q) cj_n[q] = g_nlist[(unsigned)q * un + in];
                    }
                    const unsigned myslot = (unsigned)(FS_SLOTWORD(slotpack, k) >> (16 * (k & 3))) & 0xffffu;
                    if (xi.w > 0.0f && myslot < (unsigned)FS_FUSED_CSET_CAP && i_raw < n) {
                        cacc[myslot] = FsVec4{a.d0, a.d1, a.d2, __int_as_float(a.cnt)};  // pass 2 finishes this particle
                    } else if (xi.w > 0.0f && myslot != 0xffffu && i_raw < n) {  // queued: pass 2 finishes it as well
                        squeue[myslot - FS_FUSED_CSET_CAP] =
                            FsVec4{a.d0, a.d1, a.d2, __int_as_float(i | (cj[0] << 12) | (a.cnt << 24) | ((cnt - 1) << 29))};
                    } else if (xi.w > 0.0f) {
                        const float ri0 = xi0 - X0x[i], ri1 = xi1 - X0y[i], ri2 = xi2 - X0z[i];
                        // A particle handled here has few candidates (the set takes the heavy ones first): the first
                        // FS_FUSED_PREFETCH_CAND ids are already in registers.  No load may sit on this path -- the
                        // compiler guards a conditionally loaded id with s_waitcnt vmcnt(0), which would also wait for the
                        // next-particle prefetch issued just above, every trip -- so the tail beyond the registers has
                        // its own, rarely entered loop.
#pragma unroll 1
                        for (int q = 0; q < FS_FUSED_PREFETCH_CAND && q < cnt; ++q) {
                            const int j = cj[0];
#pragma unroll
                            for (int r = 0; r + 1 < FS_FUSED_PREFETCH_CAND; ++r) cj[r] = cj[r + 1];
                            const FsVec4 xj = X[j];
                            fs_particle_contact(a, xi0, xi1, xi2, wi, ri0, ri1, ri2, xj, xj.x - X0x[j], xj.y - X0y[j],
                                                xj.z - X0z[j], c.restd, c.restd2, c.mu_p);
                        }
                        if (cnt > FS_FUSED_PREFETCH_CAND) {
#pragma unroll 1
                            for (int sq = FS_FUSED_PREFETCH_CAND; sq < cnt; ++sq) {
                                const int j = g_nlist[(unsigned)sq * un + (unsigned)i];
                                const FsVec4 xj = X[j];
                                fs_particle_contact(a, xi0, xi1, xi2, wi, ri0, ri1, ri2, xj, xj.x - X0x[j], xj.y - X0y[j],
                                                    xj.z - X0z[j], c.restd, c.restd2, c.mu_p);
                            }
                        }
                        fs_fused_shape_contacts(a, c, E.p, sh, sub, smask, xi0, xi1, xi2, ri0, ri1, ri2);
                        fs_apply(a, c.relax, nx, ny, nz);
                    }
#pragma unroll
                    for (int q = FS_FUSED_PPT - 1; q > 0; --q) { rx[q] = rx[q - 1]; ry[q] = ry[q - 1]; rz[q] = rz[q - 1]; }
                    rx[0] = nx; ry[0] = ny; rz[0] = nz;
                    if (COMPACT) {
#pragma unroll
                        for (int q = 0; q < JW; ++q) { jw[q] = jw_n[q]; cw[q] = cw_n[q]; }
                    }
                    cntw = cntw_n;
#pragma unroll
                    for (int q = 0; q < FS_FUSED_PREFETCH_CAND; ++q) cj[q] = cj_n[q];
                    FS_TS(6)
                }
                __syncthreads();
                FS_TS(7)
                // ---- pass 2: finish the contact-set particle of this thread
                float n2x = 0.0f, n2y = 0.0f, n2z = 0.0f;
                if (i2 >= 0) {
                    const FsVec4 xi = X[i2];
                    const FsVec4 pa = cacc[t];
                    FsAcc a = {pa.x, pa.y, pa.z, __float_as_int(pa.w)};
                    const float xi0 = xi.x, xi1 = xi.y, xi2 = xi.z, wi = xi.w;
                    const float ri0 = xi0 - X0x[i2], ri1 = xi1 - X0y[i2], ri2 = xi2 - X0z[i2];
                    int cjt[FS_FUSED_PREFETCH_CAND];
#pragma unroll
                    for (int q = 0; q < FS_FUSED_PREFETCH_CAND; ++q) cjt[q] = cj2[q];
                    for (int s0 = 0; s0 < cnt2; s0 += FS_FUSED_PREFETCH_CAND) {
                        int cn[FS_FUSED_PREFETCH_CAND];
#pragma unroll
                        for (int q = 0; q < FS_FUSED_PREFETCH_CAND; ++q) {
                            const int sn = s0 + FS_FUSED_PREFETCH_CAND + q;
                            cn[q] = sn < cnt2 ? g_nlist[(unsigned)sn * un + (unsigned)i2] : 0;
                        }
#pragma unroll 1
                        for (int q = 0; q < FS_FUSED_PREFETCH_CAND && s0 + q < cnt2; ++q) {
                            const int j = cjt[0];
#pragma unroll
                            for (int r = 0; r + 1 < FS_FUSED_PREFETCH_CAND; ++r) cjt[r] = cjt[r + 1];
                            const FsVec4 xj = X[j];
                            fs_particle_contact(a, xi0, xi1, xi2, wi, ri0, ri1, ri2, xj, xj.x - X0x[j], xj.y - X0y[j],
                                                xj.z - X0z[j], c.restd, c.restd2, c.mu_p);
                        }
#pragma unroll
                        for (int q = 0; q < FS_FUSED_PREFETCH_CAND; ++q) cjt[q] = cn[q];
                    }
                    fs_fused_shape_contacts(a, c, E.p, sh, sub, smask2, xi0, xi1, xi2, ri0, ri1, ri2);
                    n2x = xi0; n2y = xi1; n2z = xi2;
                    fs_apply(a, c.relax, n2x, n2y, n2z);
                }
#if FS_FUSED_QUEUE
                // ---- the overflow queue: entry q of the waves behind the first (which holds the set's longest lists), the new
                //      position back into the entry until the publish below
#pragma unroll 1
                for (int q = t - 64; q >= 0 && q < n_queued; q += FS_FUSED_QUEUE_LANES) {
                    const FsVec4 pa = squeue[q];
                    const unsigned word = (unsigned)__float_as_int(pa.w);
                    const int i = (int)(word & 0xfffu), more = (int)(word >> 29);
                    int j = (int)((word >> 12) & 0xfffu);
                    int jn = more > 0 ? g_nlist[un + (unsigned)i] : 0;  // (the second candidate travels while the first is evaluated)
                    const unsigned smask = (unsigned)g_ncount[i] >> FS_SHAPE_MASK_SHIFT;  // (the entry's word has no room for it)
                    FsAcc a = {pa.x, pa.y, pa.z, (int)((word >> 24) & 31u)};
                    const FsVec4 xi = X[i];
                    float xi0 = xi.x, xi1 = xi.y, xi2 = xi.z;
                    const float wi = xi.w;
                    const float ri0 = xi0 - X0x[i], ri1 = xi1 - X0y[i], ri2 = xi2 - X0z[i];
#pragma unroll 1
                    for (int sq = 0; sq <= more; ++sq) {
                        const FsVec4 xj = X[j];
                        fs_particle_contact(a, xi0, xi1, xi2, wi, ri0, ri1, ri2, xj, xj.x - X0x[j], xj.y - X0y[j], xj.z - X0z[j], c.restd,
                                            c.restd2, c.mu_p);
                        j = jn;
                        if (sq + 2 <= more) jn = g_nlist[(unsigned)(sq + 2) * un + (unsigned)i];
                    }
                    fs_fused_shape_contacts(a, c, E.p, sh, sub, smask, xi0, xi1, xi2, ri0, ri1, ri2);
                    fs_apply(a, c.relax, xi0, xi1, xi2);
                    squeue[q] = FsVec4{xi0, xi1, xi2, pa.w};
                }
#endif
                FS_TS(8)
                __syncthreads();  // every read of the old iterate is done
                FS_TS(9)
#pragma unroll
                for (int q = 0; q < FS_FUSED_PPT; ++q) {  // particle q of the thread sits in slot PPT-1-q
                    const int i = t + q * FS_FUSED_THREADS;
                    const bool in_set = ((unsigned)(slotpack[q >> 2] >> (16 * (q & 3))) & 0xffffu) != 0xffffu;
                    if (i < n && !in_set) { FsVec4 &d = X[i]; d.x = rx[FS_FUSED_PPT - 1 - q]; d.y = ry[FS_FUSED_PPT - 1 - q]; d.z = rz[FS_FUSED_PPT - 1 - q]; }
                }
                if (i2 >= 0) { FsVec4 &d = X[i2]; d.x = n2x; d.y = n2y; d.z = n2z; }
#if FS_FUSED_QUEUE
#pragma unroll 1
                for (int q = t - 64; q >= 0 && q < n_queued; q += FS_FUSED_QUEUE_LANES) {
                    const FsVec4 pn = squeue[q];
                    FsVec4 &d = X[__float_as_int(pn.w) & 0xfff];
                    d.x = pn.x; d.y = pn.y; d.z = pn.z;
                }
#endif
                __syncthreads();
                FS_TS(10)
            }
            // ---- finalize: new velocity to global, new substep-start position into X0 (own entries only; every other
            //      thread is past the last iteration barrier and reads X0 again only after the next predict barrier)
            for (int i = t; i < n; i += FS_FUSED_THREADS) {
                const FsVec4 xf = X[i];
                FsVec4 p0 = FsVec4{X0x[i], X0y[i], X0z[i], xf.w};
                FsVec4 v = fs_ld4(g_vel, i);
                fs_fused_finalize(c, p0, v, xf);
                fs_st4(g_vel, i, v);
                X0x[i] = p0.x; X0y[i] = p0.y; X0z[i] = p0.z;
            }
        }
    }
    for (int i = t; i < n; i += FS_FUSED_THREADS) fs_st4(g_pos, i, FsVec4{X0x[i], X0y[i], X0z[i], X[i].w});
#ifdef FS_TIMING
    FS_TS(11)
    if (blockIdx.x == 0 && t == 0)
        printf("DBG cellvisits(wave) %llu trips(wave) %llu candidates(lane) %llu survivors(lane) %llu phaseB trips(wave) %llu phaseB candidate trips(wave) %llu cyclesA %llu cyclesB %llu\n",
               fs_dbg_cnt[0], fs_dbg_cnt[1], fs_dbg_cnt[2], fs_dbg_cnt[3], fs_dbg_cnt[4], fs_dbg_cnt[5], fs_dbg_cnt[6], fs_dbg_cnt[7]);
    if (blockIdx.x == 0 && (t & 63) == 0)
        printf("TS wave %2d: init %llu grid %llu find %llu findwait %llu cset %llu springs %llu rest %llu bar1 %llu pass2 %llu bar2 %llu publish %llu final %llu\n",
               t >> 6, ts_acc[0], ts_acc[1], ts_acc[2], ts_acc[3], ts_acc[4], ts_acc[5], ts_acc[6], ts_acc[7], ts_acc[8],
               ts_acc[9], ts_acc[10], ts_acc[11]);
#endif
}
