// fs_types.h -- shared host/device plain-data types of libflingsim (MI355X cloth hot path).
#pragma once
#include <stdint.h>

#define FS_MAX_SHAPES 16
#define FS_MAX_NEIGHBORS 96   // g_maxNeighborsPerParticle, reference main.cpp:826
#define FS_MAX_PLANES 8       // NvFlexParams::planes, reference NvFlex.h:149

// NvFlex.h:159-192 phase bits
#define FS_PHASE_GROUP_MASK 0x000fffff
#define FS_PHASE_SELF_COLLIDE (1 << 20)
#define FS_PHASE_SELF_COLLIDE_FILTER (1 << 21)
#define FS_PHASE_CHANNEL_MASK 0x7f000000

// hashed uniform grid used for particle-neighbour search by the streaming path (fs_cell_hash, 14 bits)
#define FS_GRID_BUCKETS 16384

// Effective NvFlexParams subset that reaches the cloth step (reference NvFlex.h:95-154; values main.cpp:717-884,
// softgym_cloth.h:154-170).
struct FsParams {
    int numIterations, numSubsteps;
    float dt;
    float gravity[3];
    float radius, solidRestDistance, collisionDistance, shapeCollisionMargin, particleCollisionMargin;
    float dynamicFriction, staticFriction, particleFriction;
    float damping, sleepThreshold, relaxationFactor, maxAcceleration, maxSpeed;
    float restitution, adhesion, dissipation;
    int numPlanes;
    float planes[FS_MAX_PLANES][4];
    int maxNeighbors, maxContacts, relaxationMode;
};

// Canonical incident-spring list of an interior particle of CreateSpringGrid (helpers.h:838-924), in spring-id order:
// row-major pass [stretch x-1, bend x-2, shear (x+1,z-1), shear (x-1,z-1)] seen from both ends, then the column pass
// [stretch z-1, bend z-2].  The grid-64 fused kernel (fs_fused_grid_kernel.h) hard-wires this order (the dz of a slot
// is an immediate offset of its LDS gathers); the host verifies that a cloth follows it before selecting that kernel.
#define FS_G64_SLOTS 12
#define FS_G64_DX_LIST {-1, -2, +1, -1, +1, +2, -1, +1, 0, 0, 0, 0}
#define FS_G64_DZ_LIST {0, 0, -1, -1, 0, 0, +1, +1, -1, -2, +1, +2}

struct FsVec4 { float x, y, z, w; };
struct FsU32x4 { uint32_t x, y, z, w; };

// Kinematic collision shapes of one episode (spheres; reference helpers.h:484 AddSphere).
struct FsShapesDev {
    int count;
    int pad[3];
    FsVec4 pos[FS_MAX_SHAPES];   // xyz = current centre, w = radius
    FsVec4 prev[FS_MAX_SHAPES];  // xyz = previous centre (NvFlex.h:981-982)
};

// The kinematic spheres of one LAUNCH SLOT as the streaming iterations need them: centre at the end of every substep and
// displacement during it (fs_shape_sweep: linear sweep from the previous to the current transform over the frame).  The
// sweep is the same for every particle and every one of the 30 iterations of a substep, and uniform float arithmetic runs
// on the vector unit here (no scalar float ALU), so fs_k_slot_table works it out once per launch sequence and the iterate
// kernels read it through scalar loads.
#define FS_SWEEP_MAX_SUBSTEPS 8
struct FsSlotSweeps {
    int count;      // spheres
    int substeps;   // numSubsteps the table was built for (0: none -- the kernels then sweep per particle as before)
    int pad[2];
    FsVec4 c[FS_SWEEP_MAX_SUBSTEPS][FS_MAX_SHAPES];  // xyz centre at the end of the substep, w radius
    FsVec4 s[FS_SWEEP_MAX_SUBSTEPS][FS_MAX_SHAPES];  // xyz displacement during the substep
};

// Device-side descriptor of one episode.  Particle fields are separate arrays (SoA); each field is 16 B per particle
// so one lane moves one dwordx4.
struct FsEnvDev {
    int n, m, max_deg, has_scene;
    // dynamic state
    FsVec4 *pos;    // xyz + invMass (NvFlex.h:545)
    FsVec4 *vel;    // xyz, w unused
    int *phase;
    // per-substep scratch
    FsVec4 *x0;     // substep-start position
    FsVec4 *v0;     // substep-start velocity
    FsVec4 *xa;     // Jacobi ping
    FsVec4 *xb;     // Jacobi pong
    int *ncount;    // particle-contact candidates per particle
    int *nlist;     // [FS_MAX_NEIGHBORS][n] slot-major
    int *cell_count;  // [FS_GRID_BUCKETS + 1] -> exclusive starts after the scan
    int *cell_fill;   // [FS_GRID_BUCKETS]
    int *cell_items;  // [n] particle ids grouped by bucket
    // topology (shared between episodes with the same cloth)
    const FsVec4 *rest;     // rest pose (main.cpp:971-973)
    const int *adj_off;     // CSR: particle -> incident springs, ascending spring id
    const int *adj_j;       // other endpoint
    const float *adj_len;   // rest length
    const float *adj_k;     // stiffness
    // ELL copy of the same adjacency, slot-major [max_deg][n]; j < 0 marks an empty slot
    const int *ell_j;
    const float *ell_len;
    const float *ell_k;
    // compact adjacency (fused kernel): dictionary of distinct (len, k) pairs + packed codes / neighbour ids
    int dict_size;           // 0 = unavailable
    int slot_env;            // in a launch table (fs_k_slot_table): the episode this slot holds, -1 = retired; unused elsewhere
    const float *dict;       // [256][2]
    const uint32_t *code_w;  // [8][n]
    const uint32_t *nbr_w;   // [8][n]
    // compact adjacency (streaming kernels): (j - i, len, k) dictionary + one byte per spring slot (fs_scene.h)
    const FsVec4 *sdict;     // [256], x = bits(j - i)
    const FsU32x4 *scode;    // [n]
    int sdict_size;          // 0 = unavailable
    int pad1;
    // grid pattern (fs_scene.h): canonical (dx, dz) offsets of a particle's springs in spring-id order; gp_count = 0 = none
    int gp_count, gp_dimx, gp_dimz, gp_pad;
    int gp_dx[16], gp_dz[16];
    // grid-64 form (fused kernel for dimx == 64 grid cloths, fs_scene.h): rest lengths in canonical slot order,
    // slot-major [12][n] (0 where the slot leaves the grid), and the per-slot stiffness halved; g64_ok = 0 = unavailable
    const float *g64_L;
    float g64_kh[FS_G64_SLOTS];
    int g64_ok;
    // the table alone for any canonical grid cloth (streaming grid form): gp_L_ok, the full stiffness per slot, and
    // gp_magic = ceil(2^32 / gp_dimx) so that row = (i * gp_magic) >> 32
    int gp_L_ok;
    float gp_k[FS_G64_SLOTS];
    uint32_t gp_magic;
    int gp_halvable;  // every stiffness positive and exactly halvable: g64_kh = stiffness / 2 is usable (equal-mass spring form)
    // rest-pose neighbour ids for the SelfCollideFilter test, packed like nbr_w but holding plain particle ids
    const uint32_t *restnear_w;  // [8][n], 0xffff = empty
    int restnear_ok;
    int find_mode;  // host-maintained summary of the phases (fs_set_scene / fs_set_phases), used by the streaming search:
                    // 0 mixed phases: test every pair; 1 one phase with SelfCollide|SelfCollideFilter and restnear_ok: rest-near
                    // membership test; 2 one phase, SelfCollide without filter: every pair in range; 3 one phase, no
                    // SelfCollide: no pairs; 4 like 1 on a grid cloth whose rest-near sets are the 8 grid neighbours
                    // (restnear_ok == 2): index differences instead of packed ids
    FsParams p;
};
