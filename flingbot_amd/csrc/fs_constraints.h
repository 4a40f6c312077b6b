// fs_constraints.h -- per-particle constraint projections shared by the streaming and the fused LDS solver kernels.
//
// Jacobi with local relaxation (reference NvFlex.h:86-90,152-153): every constraint touching particle i adds its
// position delta to an accumulator and bumps a counter; the particle then moves by relaxationFactor * delta / count.
// Accumulation order per particle is fixed (springs by ascending spring id, particle contacts by ascending
// neighbour id, planes, spheres) and the build uses -ffp-contract=off, so results do not depend on the launch
// geometry and are reproducible bit for bit.  Fused multiply-adds appear exactly where the specification (and the CPU
// oracle, with fmaf) spells them out: FS_FMA / fs_dot3 below -- the compiler never contracts on its own.
//
// Semantics: distance constraints NvFlex.h:656-667, solidRestDistance :101, particleFriction :107, inelastic
// particle contacts :108, collisionDistance :145, dynamic/static friction :105-106, planes :149, shapes :941-987.
// The friction model (positional Coulomb friction on the tangential displacement since the substep start) follows
// Macklin et al. 2014, section 6.1 -- the closed-source reference cannot be consulted.
#pragma once
#include <hip/hip_runtime.h>

#include "fs_types.h"

struct FsAcc {
    float d0, d1, d2;
    int cnt;
};

#define FS_FMA(a, b, c) __builtin_fmaf((a), (b), (c))  // one rounding, v_fma_f32; the oracle uses fmaf()
// a . b = fma(az, bz, fma(ay, by, ax * bx))
__device__ __forceinline__ float fs_dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return FS_FMA(az, bz, FS_FMA(ay, by, ax * bx));
}

// Reciprocal square root for every length inside the constraint sweeps: the hardware's v_rsq_f32 (1 ulp) on
// max(x, FLT_MIN) -- two instructions where rounds 1-3 spelled an integer seed + three Newton steps out in twelve (the
// reciprocal root was nearly half of the arithmetic of a spring; EXPERIMENTS R4.1: -7 ... -9 % on the fused kernels).
// The clamp keeps every result finite: a zero or denormal squared length gives 2^63 and the product l2 * rsqrt(l2) -- the
// length -- stays 0 or tiny, so the `length > 0` tests below work as they read.  The instruction is a pure function of
// its input bits, and the CPU oracle reproduces it from a table of this chip's 2^24 (exponent parity, mantissa) results
// (oracle/v_rsq_f32_gfx950.npz, dumped through fs_eval_rsqrt; the GPU suite re-reads the whole table on the box it runs
// on), so HIP and oracle still agree bit for bit.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "fs_rsqrt is gfx950's v_rsq_f32 AS DATA: the goldens and oracle/v_rsq_f32_gfx950.npz hold this chip's results. Another --offload-arch needs its own table (tests/golden/make_rsq_table.py) before it may be added to the build."
#endif
__device__ __forceinline__ float fs_rsqrt(float x) {
    return __builtin_amdgcn_rsqf(__builtin_fmaxf(x, 1.17549435e-38f));
}

// friction scale on a tangential displacement of length tl = tl2 * inv_tl under penetration pen
__device__ __forceinline__ float fs_friction_scale(float tl, float inv_tl, float pen, float mu_s, float mu_k) {
    if (tl < mu_s * pen) return 1.0f;
    float lim = mu_k * pen;
    return (tl > lim) ? lim * inv_tl : 1.0f;
}

// wi / (wi + wj), correctly rounded.  Shortcuts for the two cases that cover a cloth (equal masses: x / 2x == 0.5;
// pinned neighbour: x / x == 1) are exact, so the result equals the IEEE division bit for bit while the ~10
// instruction division sequence is skipped by whole waves.
__device__ __forceinline__ float fs_mass_ratio(float wi, float wj) {
    if (wj == wi) return 0.5f;
    if (wj == 0.0f) return 1.0f;
    return wi / (wi + wj);
}

// distance constraint between particle i (xi, wi) and j (xj.w = invMass), rest length L, stiffness k (<0: tether)
__device__ __forceinline__ void fs_spring(FsAcc &a, float xi0, float xi1, float xi2, float wi, const FsVec4 xj, float L,
                                          float k) {
    float ex = xi0 - xj.x, ey = xi1 - xj.y, ez = xi2 - xj.z;
    float l2 = fs_dot3(ex, ey, ez, ex, ey, ez);
    float inv_len = fs_rsqrt(l2);
    float len = l2 * inv_len;
    if (!(len > 0.0f)) return;
    float C = len - L;
    if (k < 0.0f) {
        if (!(C > 0.0f)) return;
        k = -k;
    }
    float ratio = fs_mass_ratio(wi, xj.w);
    float sc = (k * ratio) * (C * inv_len);
    a.d0 = FS_FMA(-ex, sc, a.d0);
    a.d1 = FS_FMA(-ey, sc, a.d1);
    a.d2 = FS_FMA(-ez, sc, a.d2);
    a.cnt++;
}

// Branch-free form of fs_spring for straight-line scheduling (the fused kernel batches its LDS gathers in front of
// it).  Every lane evaluates the same instruction stream; inactive constraints (zero length, slack tether) are
// masked with selects, so an active constraint performs exactly the operations of fs_spring, in the same order.
__device__ __forceinline__ void fs_spring_bf(FsAcc &a, float xi0, float xi1, float xi2, float wi, const FsVec4 xj, float L,
                                             float k) {
    float ex = xi0 - xj.x, ey = xi1 - xj.y, ez = xi2 - xj.z;
    float l2 = fs_dot3(ex, ey, ez, ex, ey, ez);
    float inv_len = fs_rsqrt(l2);
    float len = l2 * inv_len;
    float C = len - L;
    // (bitwise, not short-circuit, logic: short-circuit forms become exec-mask branches)
    const bool tether = k < 0.0f;
    const bool active = (len > 0.0f) & (!tether | (C > 0.0f));
    const float kk = tether ? -k : k;
    // wi / (wi + wj): exact shortcuts (see fs_mass_ratio).  The common case -- every lane's neighbour has the lane's own
    // mass -- is the fall-through path; anything else is moved out of line.
    const float wj = xj.w;
    float ratio = 0.5f;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(wj != wi) != 0ull, 0)) {
        ratio = (wj == wi) ? 0.5f : 1.0f;
        const bool odd = active & (wj != wi) & (wj != 0.0f);
        if (__builtin_amdgcn_ballot_w64(odd) != 0ull) ratio = odd ? wi / (wi + wj) : ratio;
    }
    // an inactive constraint contributes sc = +0: fma(-e, 0, d) == d for every finite e (the accumulators start at +0
    // and can never become -0), so one select on the scale replaces three on the accumulators
    float sc = active ? (kk * ratio) * (C * inv_len) : 0.0f;
    a.d0 = FS_FMA(-ex, sc, a.d0);
    a.d1 = FS_FMA(-ey, sc, a.d1);
    a.d2 = FS_FMA(-ez, sc, a.d2);
    a.cnt += active ? 1 : 0;
}

// fs_spring_bf with a per-lane "this slot exists" predicate (grid forms: a slot that leaves the grid gathers the particle
// itself and must neither move nor count it, whatever its length)
// POSK: the caller knows k > 0 (no tether): the slack test and the sign flip drop out, an active constraint performs exactly
// the same operations
template <bool POSK = false>
__device__ __forceinline__ void fs_spring_bfm(FsAcc &a, float xi0, float xi1, float xi2, float wi, const FsVec4 xj, float L,
                                              float k, bool in) {
    float ex = xi0 - xj.x, ey = xi1 - xj.y, ez = xi2 - xj.z;
    float l2 = fs_dot3(ex, ey, ez, ex, ey, ez);
    float inv_len = fs_rsqrt(l2);
    float len = l2 * inv_len;
    float C = len - L;
    const bool tether = !POSK && k < 0.0f;
    const bool active = POSK ? (in & (len > 0.0f)) : (in & (len > 0.0f) & (!tether | (C > 0.0f)));
    const float kk = tether ? -k : k;
    const float wj = xj.w;
    float ratio = 0.5f;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(wj != wi) != 0ull, 0)) {
        ratio = (wj == wi) ? 0.5f : 1.0f;
        const bool odd = active & (wj != wi) & (wj != 0.0f);
        if (__builtin_amdgcn_ballot_w64(odd) != 0ull) ratio = odd ? wi / (wi + wj) : ratio;
    }
    float sc = active ? (kk * ratio) * (C * inv_len) : 0.0f;
    a.d0 = FS_FMA(-ex, sc, a.d0);
    a.d1 = FS_FMA(-ey, sc, a.d1);
    a.d2 = FS_FMA(-ez, sc, a.d2);
    a.cnt += active ? 1 : 0;
}

// fs_spring_bf for the case that covers a free cloth: the neighbour has the particle's own mass (ratio == 0.5 exactly)
// and the cloth has no tethers (every k > 0); kh = 0.5 * k (exact: a power-of-two scaling).  Performs exactly the
// operations of fs_spring on an active constraint, in the same order -- (k * 0.5) * (C * inv_len) -- minus the tests
// whose outcome is known.
__device__ __forceinline__ void fs_spring_fast(FsAcc &a, float xi0, float xi1, float xi2, const FsVec4 xj, float L,
                                               float kh) {
    float ex = xi0 - xj.x, ey = xi1 - xj.y, ez = xi2 - xj.z;
    float l2 = fs_dot3(ex, ey, ez, ex, ey, ez);
    float inv_len = fs_rsqrt(l2);
    float len = l2 * inv_len;
    float C = len - L;
    const bool active = len > 0.0f;
    float sc = active ? kh * (C * inv_len) : 0.0f;
    a.d0 = FS_FMA(-ex, sc, a.d0);
    a.d1 = FS_FMA(-ey, sc, a.d1);
    a.d2 = FS_FMA(-ez, sc, a.d2);
    a.cnt += active ? 1 : 0;
}

// particle-particle contact; (ri*) = xi - x0_i, rj = xj - x0_j (displacement since substep start)
__device__ __forceinline__ void fs_particle_contact(FsAcc &a, float xi0, float xi1, float xi2, float wi, float ri0,
                                                    float ri1, float ri2, const FsVec4 xj, float rj0, float rj1,
                                                    float rj2, float restd, float restd2, float mu) {
    float ex = xi0 - xj.x, ey = xi1 - xj.y, ez = xi2 - xj.z;
    float l2 = fs_dot3(ex, ey, ez, ex, ey, ez);
    if (!(l2 < restd2)) return;
    float inv = fs_rsqrt(l2);
    float dist = l2 * inv;
    float nx, ny, nz;
    if (dist > 0.0f) {
        nx = ex * inv; ny = ey * inv; nz = ez * inv;
    } else {
        nx = 0.0f; ny = 1.0f; nz = 0.0f;
    }
    float pen = restd - dist;
    float ratio = fs_mass_ratio(wi, xj.w);
    float cn = pen * ratio;
    float c0 = nx * cn, c1 = ny * cn, c2 = nz * cn;
    if (mu > 0.0f) {
        float rx = ri0 - rj0, ry = ri1 - rj1, rz = ri2 - rj2;
        float rn = fs_dot3(rx, ry, rz, nx, ny, nz);
        float tx = FS_FMA(-nx, rn, rx), ty = FS_FMA(-ny, rn, ry), tz = FS_FMA(-nz, rn, rz);
        float tl2 = fs_dot3(tx, ty, tz, tx, ty, tz);
        if (tl2 > 0.0f) {
            float inv_tl = fs_rsqrt(tl2);
            float tl = tl2 * inv_tl;
            float fs = fs_friction_scale(tl, inv_tl, pen, mu, mu) * ratio;
            c0 = FS_FMA(-tx, fs, c0); c1 = FS_FMA(-ty, fs, c1); c2 = FS_FMA(-tz, fs, c2);
        }
    }
    a.d0 = a.d0 + c0; a.d1 = a.d1 + c1; a.d2 = a.d2 + c2;
    a.cnt++;
}

__device__ __forceinline__ void fs_plane_contact(FsAcc &a, float xi0, float xi1, float xi2, float ri0, float ri1,
                                                 float ri2, float p0, float p1, float p2, float p3, float cd, float mu_s,
                                                 float mu_k) {
    float sdist = fs_dot3(p0, p1, p2, xi0, xi1, xi2) + p3;
    if (!(sdist < cd)) return;
    float pen = cd - sdist;
    float c0 = p0 * pen, c1 = p1 * pen, c2 = p2 * pen;
    float rn = fs_dot3(ri0, ri1, ri2, p0, p1, p2);
    float tx = FS_FMA(-p0, rn, ri0), ty = FS_FMA(-p1, rn, ri1), tz = FS_FMA(-p2, rn, ri2);
    float tl2 = fs_dot3(tx, ty, tz, tx, ty, tz);
    if (tl2 > 0.0f) {
        float inv_tl = fs_rsqrt(tl2);
        float tl = tl2 * inv_tl;
        float fs = fs_friction_scale(tl, inv_tl, pen, mu_s, mu_k);
        c0 = FS_FMA(-tx, fs, c0); c1 = FS_FMA(-ty, fs, c1); c2 = FS_FMA(-tz, fs, c2);
    }
    a.d0 = a.d0 + c0; a.d1 = a.d1 + c1; a.d2 = a.d2 + c2;
    a.cnt++;
}

// kinematic sphere at centre (c*) with radius r, which moved by (s*) during this substep
__device__ __forceinline__ void fs_sphere_contact(FsAcc &a, float xi0, float xi1, float xi2, float ri0, float ri1,
                                                  float ri2, float c0_, float c1_, float c2_, float r, float s0,
                                                  float s1, float s2, float cd, float mu_s, float mu_k) {
    float ex = xi0 - c0_, ey = xi1 - c1_, ez = xi2 - c2_;
    float l2 = fs_dot3(ex, ey, ez, ex, ey, ez);
    float lim = r + cd;
    if (!(l2 < lim * lim)) return;
    float inv = fs_rsqrt(l2);
    float dist = l2 * inv;
    float nx, ny, nz;
    if (dist > 0.0f) {
        nx = ex * inv; ny = ey * inv; nz = ez * inv;
    } else {
        nx = 0.0f; ny = 1.0f; nz = 0.0f;
    }
    float pen = lim - dist;
    float c0 = nx * pen, c1 = ny * pen, c2 = nz * pen;
    float rx = ri0 - s0, ry = ri1 - s1, rz = ri2 - s2;
    float rn = fs_dot3(rx, ry, rz, nx, ny, nz);
    float tx = FS_FMA(-nx, rn, rx), ty = FS_FMA(-ny, rn, ry), tz = FS_FMA(-nz, rn, rz);
    float tl2 = fs_dot3(tx, ty, tz, tx, ty, tz);
    if (tl2 > 0.0f) {
        float inv_tl = fs_rsqrt(tl2);
        float tl = tl2 * inv_tl;
        float fs = fs_friction_scale(tl, inv_tl, pen, mu_s, mu_k);
        c0 = FS_FMA(-tx, fs, c0); c1 = FS_FMA(-ty, fs, c1); c2 = FS_FMA(-tz, fs, c2);
    }
    a.d0 = a.d0 + c0; a.d1 = a.d1 + c1; a.d2 = a.d2 + c2;
    a.cnt++;
}

// sphere centre at the end of substep `sub` (0-based) and its displacement during the substep: linear sweep from the
// previous to the current transform over the frame.
__device__ __forceinline__ void fs_shape_sweep(const FsShapesDev &sh, int q, int sub, float S, float &c0, float &c1,
                                               float &c2, float &s0, float &s1, float &s2) {
    float a1 = (float)(sub + 1) / S, a0 = (float)sub / S;
    float dx = sh.pos[q].x - sh.prev[q].x, dy = sh.pos[q].y - sh.prev[q].y, dz = sh.pos[q].z - sh.prev[q].z;
    c0 = sh.prev[q].x + dx * a1; c1 = sh.prev[q].y + dy * a1; c2 = sh.prev[q].z + dz * a1;
    float b0 = sh.prev[q].x + dx * a0, b1 = sh.prev[q].y + dy * a0, b2 = sh.prev[q].z + dz * a0;
    s0 = c0 - b0; s1 = c1 - b1; s2 = c2 - b2;
}

// ---- collideShapes (the stage of NvFlex.h:205: once per substep, before the iterations; "contact planes generated within
// NvFlexParams::shapeCollisionMargin", NvFlex.h:1074).  For one particle at its PREDICTED position: the planes and the
// kinematic spheres whose surface is closer than collisionDistance + shapeCollisionMargin (NvFlex.h:145,147), at most
// maxContactsPerParticle of them (NvFlex.h:361, main.cpp:828) -- planes in index order, then spheres in index order; a sphere
// stands where the iterations of this substep see it (end of its sweep).  Result: bit q = plane q, bit FS_SHAPE_SPHERE_BIT + q = sphere q.
// The iterations test only this set (oracle: collide_shapes).  The mask rides in the candidate-count word of the particle
// (FsEnvDev::ncount: bits 0-7 = particle-contact candidates, bits 8-31 = this mask), which every iteration loads anyway.
#define FS_SHAPE_MASK_SHIFT 8
#define FS_NCOUNT_MASK 0xff
#define FS_SHAPE_SPHERE_BIT 8   // within the mask: planes are bits 0 .. FS_SHAPE_SPHERE_BIT - 1, sphere q is bit FS_SHAPE_SPHERE_BIT + q
static_assert(FS_MAX_NEIGHBORS <= FS_NCOUNT_MASK, "the candidate count must fit the low bits of the ncount word");
static_assert(FS_MAX_PLANES <= FS_SHAPE_SPHERE_BIT, "plane bits must stay below the first sphere bit");
static_assert(FS_SHAPE_SPHERE_BIT + FS_MAX_SHAPES + FS_SHAPE_MASK_SHIFT <= 32, "plane + sphere candidate bits must fit the ncount word");
template <class SphereAt>  // SphereAt(q, c0, c1, c2, r): centre + radius of sphere q at the end of the substep
__device__ __forceinline__ unsigned fs_shape_candidates_core(const FsParams &p, int n_spheres, float x0, float x1, float x2,
                                                             SphereAt sphere_at) {
    const float reach = p.collisionDistance + p.shapeCollisionMargin;
    unsigned mask = 0u;
    int listed = 0;
    for (int q = 0; q < p.numPlanes && listed < p.maxContacts; ++q) {
        const float sdist = fs_dot3(p.planes[q][0], p.planes[q][1], p.planes[q][2], x0, x1, x2) + p.planes[q][3];
        if (sdist < reach) { mask |= 1u << q; ++listed; }
    }
    for (int q = 0; q < n_spheres && listed < p.maxContacts; ++q) {
        float c0, c1, c2, r;
        sphere_at(q, c0, c1, c2, r);
        const float ex = x0 - c0, ey = x1 - c1, ez = x2 - c2;
        const float l2 = fs_dot3(ex, ey, ez, ex, ey, ez);
        const float lim = r + reach;
        if (l2 < lim * lim) { mask |= 1u << (FS_SHAPE_SPHERE_BIT + q); ++listed; }
    }
    return mask;
}
__device__ __forceinline__ unsigned fs_shape_candidates(const FsParams &p, const FsShapesDev &sh, int sub, float x0, float x1,
                                                        float x2) {
    const float S = (float)p.numSubsteps;
    return fs_shape_candidates_core(p, sh.count, x0, x1, x2, [&](int q, float &c0, float &c1, float &c2, float &r) {
        float s0, s1, s2;
        fs_shape_sweep(sh, q, sub, S, c0, c1, c2, s0, s1, s2);
        r = sh.pos[q].w;
    });
}
__device__ __forceinline__ unsigned fs_swept_shape_candidates(const FsParams &p, const FsSlotSweeps &sw, int sub, float x0,
                                                              float x1, float x2) {
    return fs_shape_candidates_core(p, sw.count, x0, x1, x2, [&](int q, float &c0, float &c1, float &c2, float &r) {
        const FsVec4 c = sw.c[sub][q];
        c0 = c.x; c1 = c.y; c2 = c.z; r = c.w;
    });
}

// planes + spheres for one particle: the candidates of `mask` (fs_shape_candidates), planes ascending, then spheres ascending.
// A wavefront none of whose lanes has a candidate -- every row of a cloth that is far from the ground and the pickers --
// skips the block; within it a sphere's sweep + test run only where some lane lists that sphere.
__device__ __forceinline__ void fs_shape_contacts(FsAcc &a, float xi0, float xi1, float xi2, float ri0, float ri1,
                                                  float ri2, const FsParams &p, const FsShapesDev &sh, int sub, unsigned mask) {
    if (__builtin_amdgcn_ballot_w64(mask != 0u) == 0ull) return;
    for (int q = 0; q < p.numPlanes; ++q)
        if ((mask >> q) & 1u)
            fs_plane_contact(a, xi0, xi1, xi2, ri0, ri1, ri2, p.planes[q][0], p.planes[q][1], p.planes[q][2], p.planes[q][3],
                             p.collisionDistance, p.staticFriction, p.dynamicFriction);
    const float S = (float)p.numSubsteps;
    for (int q = 0; q < sh.count; ++q)
        if ((mask >> (FS_SHAPE_SPHERE_BIT + q)) & 1u) {
            float c0, c1, c2, s0, s1, s2;
            fs_shape_sweep(sh, q, sub, S, c0, c1, c2, s0, s1, s2);
            fs_sphere_contact(a, xi0, xi1, xi2, ri0, ri1, ri2, c0, c1, c2, sh.pos[q].w, s0, s1, s2, p.collisionDistance,
                              p.staticFriction, p.dynamicFriction);
        }
}

// the same with the spheres' sweeps of this substep taken from the launch slot's table (FsSlotSweeps, built once per launch
// sequence by fs_k_slot_table with fs_shape_sweep's expressions: identical numbers, scalar loads instead of ~28 vector
// instructions per sphere, particle and iteration)
__device__ __forceinline__ void fs_swept_shape_contacts(FsAcc &a, float xi0, float xi1, float xi2, float ri0, float ri1,
                                                        float ri2, const FsParams &p, const FsSlotSweeps &sw, int sub,
                                                        unsigned mask) {
    if (__builtin_amdgcn_ballot_w64(mask != 0u) == 0ull) return;
    for (int q = 0; q < p.numPlanes; ++q)
        if ((mask >> q) & 1u)
            fs_plane_contact(a, xi0, xi1, xi2, ri0, ri1, ri2, p.planes[q][0], p.planes[q][1], p.planes[q][2], p.planes[q][3],
                             p.collisionDistance, p.staticFriction, p.dynamicFriction);
    for (int q = 0; q < sw.count; ++q)
        if ((mask >> (FS_SHAPE_SPHERE_BIT + q)) & 1u) {
            const FsVec4 c = sw.c[sub][q], s = sw.s[sub][q];
            fs_sphere_contact(a, xi0, xi1, xi2, ri0, ri1, ri2, c.x, c.y, c.z, c.w, s.x, s.y, s.z, p.collisionDistance,
                              p.staticFriction, p.dynamicFriction);
        }
}

// applyDeltas with local relaxation
__device__ __forceinline__ void fs_apply(const FsAcc &a, float relax, float &x0, float &x1, float &x2) {
    if (a.cnt > 0) {
        float sc = relax / (float)a.cnt;
        x0 = FS_FMA(a.d0, sc, x0); x1 = FS_FMA(a.d1, sc, x1); x2 = FS_FMA(a.d2, sc, x2);
    }
}

// neighbour-search pair filter on phases and rest pose (NvFlex.h:165-166,564-565)
__device__ __forceinline__ bool fs_pair_allowed(int phi, int phj, const FsVec4 ri, const FsVec4 rj, float r2) {
    if ((phi & FS_PHASE_GROUP_MASK) == (phj & FS_PHASE_GROUP_MASK)) {
        if (!((phi & FS_PHASE_SELF_COLLIDE) && (phj & FS_PHASE_SELF_COLLIDE))) return false;
        if ((phi | phj) & FS_PHASE_SELF_COLLIDE_FILTER) {
            float ex = ri.x - rj.x, ey = ri.y - rj.y, ez = ri.z - rj.z;
            float e2 = ex * ex + ey * ey + ez * ez;
            if (e2 < r2) return false;
        }
    }
    return true;
}

// Spatial hash of an integer cell: multiplicative mix, top `bits` bits.  Any cell -> bucket map is correct here because
// the search re-derives the TRUE cell of every candidate and accepts it only from the visit of that exact cell, so
// aliasing (two of the 27 visited cells sharing a bucket, or far-away cells colliding) can never duplicate or invent a
// neighbour; the hash only has to spread cells evenly for ANY cloth orientation.  (An axis-wrapped grid degenerates:
// a sheet hanging vertically spans 36 cells along the axis that was given 4.)
__device__ __forceinline__ int fs_cell_hash(int cx, int cy, int cz, int bits) {
    const unsigned h = (unsigned)cx * 0x9E3779B1u + (unsigned)cy * 0x85EBCA77u + (unsigned)cz * 0xC2B2AE3Du;
    return (int)((h ^ (h >> 15)) * 0x2C1B3C6Du >> (32 - bits));
}
__device__ __forceinline__ int fs_bucket(int cx, int cy, int cz) { return fs_cell_hash(cx, cy, cz, 14); }
