/*
 * flex_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See flex_oracle.h for provenance.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  All arithmetic is IEEE fp32 with a fixed
 * operation order so the HIP kernels (also built with -ffp-contract=off, correctly rounded / and sqrt) can be
 * compared bit for bit.
 *
 * PARITY UNPINNED: the solver arithmetic of the reference lives in closed-source NVIDIA FleX 1.2.0 which is absent
 * from /root/reference; the reference has no tests or golden vectors.  Each decision is tagged
 *   [D] documented in a reference header (cited)    [I] inferred / chosen from the published FleX paper.
 */
#include "flex_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_SHAPES 16
#define ORC_MAX_NEIGHBORS 96 /* g_maxNeighborsPerParticle, main.cpp:826 */
#define ORC_MAX_PLANES 8
#define ORC_SPHERE_BIT 8     /* candidate mask of collideShapes: bit q = plane q, bit ORC_SPHERE_BIT + q = sphere q */
_Static_assert(ORC_MAX_PLANES <= ORC_SPHERE_BIT && ORC_SPHERE_BIT + ORC_MAX_SHAPES <= 24,
               "plane and sphere candidate bits must not alias (the product packs the mask above an 8-bit count)");

/* NvFlex.h:159-192 phase bits */
#define PH_GROUP_MASK 0x000fffff
#define PH_SELF_COLLIDE (1 << 20)
#define PH_SELF_COLLIDE_FILTER (1 << 21)
#define PH_CHANNEL_MASK 0x7f000000

typedef struct {
    int numIterations, numSubsteps;
    float dt;
    float gravity[3];
    float radius, solidRestDistance, collisionDistance, shapeCollisionMargin, particleCollisionMargin;
    float dynamicFriction, staticFriction, particleFriction;
    float damping, sleepThreshold, relaxationFactor, maxAcceleration, maxSpeed;
    float restitution, adhesion, dissipation;
    int numPlanes;
    float planes[8][4];
    int maxNeighbors, maxContacts, relaxationMode;
} orc_params;

struct orc_sim {
    int n, m, t;
    float *pos;   /* 4n: x y z invMass  (NvFlex.h:545) */
    float *vel;   /* 3n */
    int *phase;   /* n */
    float *rest;  /* 4n (main.cpp:971-973) */
    int *sidx;    /* 2m */
    float *slen;  /* m */
    float *sk;    /* m */
    int *tris;    /* 3t */
    float *tnrm;  /* 3t initial triangle normals */
    float *nrm;   /* 4n */
    /* CSR adjacency particle -> (spring id) in ascending spring id */
    int *adj_off; /* n+1 */
    int *adj_spr; /* 2m: spring id */
    /* shapes (spheres only are simulated; flingbot adds two, flex_utils.py:82-83) */
    int ns;
    float sh_radius[ORC_MAX_SHAPES];
    float sh_pos[ORC_MAX_SHAPES][3], sh_prev[ORC_MAX_SHAPES][3];
    float sh_rot[ORC_MAX_SHAPES][4], sh_prevrot[ORC_MAX_SHAPES][4];
    orc_params p;
    float scene_lower[3], scene_upper[3];
    /* scratch */
    float *xp, *xn, *x0, *v0;
    int *ncount, *nlist;
    int *cell_key, *cell_order;
    int max_list;  /* longest candidate list any particle has had since set_scene (white-box: PARITY.md) */
    unsigned *smask; /* n: shape-contact candidates of the substep (collideShapes): bit q = plane q, bit 8 + q = sphere q */
    float *splane;   /* ORC_ALT_CONTACT_PLANES only: n x ORC_MAX_SHAPES x 4, the frozen tangent plane of each sphere candidate */
    int missed_shape_contacts; /* white box: contacts that violated inside an iteration without being a candidate (since set_scene) */
    float *lam;    /* ORC_ALT_FRICTION_POST only */
    float *vframe; /* ORC_ALT_MAXACCEL_PER_FRAME only: 3n, the velocities at the start of the NvFlexUpdateSolver call */
    long accel_clamps;       /* white box: particle-substeps in which the maxAcceleration clamp changed a velocity (since set_scene) */
    long degenerate_normals; /* white box: contacts whose normal fell back to (0,1,0) because the pair was coincident (since set_scene) */
};

static void free_scene(orc_sim *s) {
    free(s->pos); free(s->vel); free(s->phase); free(s->rest); free(s->sidx); free(s->slen); free(s->sk);
    free(s->tris); free(s->tnrm); free(s->nrm); free(s->adj_off); free(s->adj_spr);
    free(s->xp); free(s->xn); free(s->x0); free(s->v0); free(s->ncount); free(s->nlist);
    free(s->cell_key); free(s->cell_order); free(s->lam); free(s->smask); free(s->splane); free(s->vframe);
    memset(s, 0, sizeof(*s));
}

orc_sim *orc_create(void) { return (orc_sim *)calloc(1, sizeof(orc_sim)); }
void orc_destroy(orc_sim *s) {
    if (!s) return;
    free_scene(s);
    free(s);
}

/* ---------------------------------------------------------------- scene build */

typedef struct { int n, cap; float *d; } fvec;
typedef struct { int n, cap; int *d; } ivec;
static void fpush(fvec *v, float x) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->d = (float *)realloc(v->d, sizeof(float) * v->cap); }
    v->d[v->n++] = x;
}
static void ipush(ivec *v, int x) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->d = (int *)realloc(v->d, sizeof(int) * v->cap); }
    v->d[v->n++] = x;
}

typedef struct { fvec pos, vel, slen, sk, tnrm; ivec phase, sidx, tris; } builder;

/* helpers.h:144-150 CreateSpring; Length() = maths.h:204-211 (sqrt of x*x+y*y+z*z, 0 if zero) */
static void create_spring(builder *b, int i, int j, float stiffness) {
    const float give = 0.0f;
    float dx = b->pos.d[4 * i + 0] - b->pos.d[4 * j + 0];
    float dy = b->pos.d[4 * i + 1] - b->pos.d[4 * j + 1];
    float dz = b->pos.d[4 * i + 2] - b->pos.d[4 * j + 2];
    float lsq = dx * dx + dy * dy + dz * dz;
    float len = lsq ? sqrtf(lsq) : 0.0f;
    ipush(&b->sidx, i);
    ipush(&b->sidx, j);
    fpush(&b->slen, (1.0f + give) * len);
    fpush(&b->sk, stiffness);
}

/* helpers.h:838-924 CreateSpringGrid (dz == 1 for cloth) */
static void create_spring_grid(builder *b, const float lower[3], int dx, int dy, int dz, float radius, int phase,
                               float ks, float kb, float ksh, float invMass) {
    int baseIndex = b->pos.n / 4;
    for (int z = 0; z < dz; ++z)
        for (int y = 0; y < dy; ++y)
            for (int x = 0; x < dx; ++x) {
                /* Vec3 position = lower + radius * Vec3(float(x), float(z), float(y)) */
                float px = lower[0] + radius * (float)x;
                float py = lower[1] + radius * (float)z;
                float pz = lower[2] + radius * (float)y;
                fpush(&b->pos, px); fpush(&b->pos, py); fpush(&b->pos, pz); fpush(&b->pos, invMass);
                fpush(&b->vel, 0.0f); fpush(&b->vel, 0.0f); fpush(&b->vel, 0.0f);
                ipush(&b->phase, phase);
                if (x > 0 && y > 0) {
                    ipush(&b->tris, baseIndex + (y - 1) * dx + (x - 1));
                    ipush(&b->tris, baseIndex + (y - 1) * dx + x);
                    ipush(&b->tris, baseIndex + y * dx + x);
                    ipush(&b->tris, baseIndex + (y - 1) * dx + (x - 1));
                    ipush(&b->tris, baseIndex + y * dx + x);
                    ipush(&b->tris, baseIndex + y * dx + (x - 1));
                    for (int k = 0; k < 2; ++k) { fpush(&b->tnrm, 0.0f); fpush(&b->tnrm, 1.0f); fpush(&b->tnrm, 0.0f); }
                }
            }
    /* horizontal */
    for (int y = 0; y < dy; ++y)
        for (int x = 0; x < dx; ++x) {
            int index0 = y * dx + x;
            if (x > 0) create_spring(b, baseIndex + index0, baseIndex + y * dx + x - 1, ks);
            if (x > 1) create_spring(b, baseIndex + index0, baseIndex + y * dx + x - 2, kb);
            if (y > 0 && x < dx - 1) create_spring(b, baseIndex + index0, baseIndex + (y - 1) * dx + x + 1, ksh);
            if (y > 0 && x > 0) create_spring(b, baseIndex + index0, baseIndex + (y - 1) * dx + x - 1, ksh);
        }
    /* vertical */
    for (int x = 0; x < dx; ++x)
        for (int y = 0; y < dy; ++y) {
            int index0 = y * dx + x;
            if (y > 0) create_spring(b, baseIndex + index0, baseIndex + (y - 1) * dx + x, ks);
            if (y > 1) create_spring(b, baseIndex + index0, baseIndex + (y - 2) * dx + x, kb);
        }
}

static void swap_int(int *a, int *b) { int t = *a; *a = *b; *b = t; }

/* main.cpp:904-919: area-weighted vertex normals, SafeNormalize with fallback (0,1,0) */
static void compute_normals(const float *pos, const int *tris, int n, int t, float *nrm) {
    memset(nrm, 0, sizeof(float) * 4 * n);
    for (int i = 0; i < t; ++i) {
        const float *v0 = pos + 4 * tris[3 * i], *v1 = pos + 4 * tris[3 * i + 1], *v2 = pos + 4 * tris[3 * i + 2];
        float ax = v1[0] - v0[0], ay = v1[1] - v0[1], az = v1[2] - v0[2];
        float bx = v2[0] - v0[0], by = v2[1] - v0[1], bz = v2[2] - v0[2];
        float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
        for (int k = 0; k < 3; ++k) {
            float *d = nrm + 4 * tris[3 * i + k];
            d[0] += nx; d[1] += ny; d[2] += nz;
        }
    }
    for (int i = 0; i < n; ++i) {
        float *d = nrm + 4 * i;
        float l = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        if (l > 0.0f) {
            float inv = 1.0f / sqrtf(l);
            d[0] *= inv; d[1] *= inv; d[2] *= inv;
        } else { d[0] = 0.0f; d[1] = 1.0f; d[2] = 0.0f; }
        d[3] = 0.0f;
    }
}

int orc_set_scene(orc_sim *s, const float *ptr, const float *verts, int n_vert_floats, const int *stretch,
                  int n_stretch_ints, const int *bend, int n_bend_ints, const int *shear, int n_shear_ints,
                  const int *faces, int n_face_ints) {
    free_scene(s); /* Init destroys the previous solver, buffers and shapes (main.cpp:623-706) */
    orc_params *p = &s->p;
    /* ---- Init defaults, main.cpp:717-828 (only the fields that reach the cloth step) */
    p->dt = 1.0f / 100.0f;
    p->gravity[0] = 0.0f; p->gravity[1] = -9.8f; p->gravity[2] = 0.0f;
    p->radius = 0.15f;
    p->dynamicFriction = 0.0f; p->staticFriction = 0.0f; p->particleFriction = 0.0f;
    p->numIterations = 3;
    p->solidRestDistance = 0.0f;
    p->dissipation = 0.0f; p->damping = 0.0f; p->particleCollisionMargin = 0.0f; p->shapeCollisionMargin = 0.0f;
    p->collisionDistance = 0.0f; p->sleepThreshold = 0.0f; p->restitution = 0.0f; p->adhesion = 0.0f;
    p->maxSpeed = FLT_MAX; p->maxAcceleration = 100.0f;
    p->relaxationMode = 1; p->relaxationFactor = 1.0f;
    p->numSubsteps = 20; p->numPlanes = 1;
    p->maxNeighbors = 96; p->maxContacts = 6;
    for (int k = 0; k < 3; ++k) { s->scene_lower[k] = FLT_MAX; s->scene_upper[k] = -FLT_MAX; }

    /* ---- SoftgymCloth::Initialize, softgym_cloth.h:33-175 */
    builder b; memset(&b, 0, sizeof(b));
    float initX = ptr[0], initY = ptr[1], initZ = ptr[2];
    int dimx = (int)ptr[3], dimz = (int)ptr[4];
    float radius = 0.00625f;
    float ks = ptr[5], kb = ptr[6], ksh = ptr[7];
    int phase = (0 & PH_GROUP_MASK) | ((PH_SELF_COLLIDE | PH_SELF_COLLIDE_FILTER) & 0x00f00000) | PH_CHANNEL_MASK;
    int flip_mesh = (int)ptr[18];
    int num_verts = n_vert_floats / 3;
    if (num_verts > 0) {
        float mass = ptr[17] / (float)num_verts;
        float invMass = 1.0f / mass;
        float lower[3] = {initX, -initY, initZ};
        int baseIndex = 0;
        for (int i = 0; i < num_verts; ++i) {
            fpush(&b.pos, verts[3 * i] + lower[0]); fpush(&b.pos, verts[3 * i + 1] + lower[1]);
            fpush(&b.pos, verts[3 * i + 2] + lower[2]); fpush(&b.pos, invMass + 0.0f);
            fpush(&b.vel, 0.0f); fpush(&b.vel, 0.0f); fpush(&b.vel, 0.0f);
            ipush(&b.phase, phase);
        }
        int num_faces = n_face_ints / 3;
        for (int i = 0; i < num_faces; ++i) {
            for (int k = 0; k < 3; ++k) ipush(&b.tris, baseIndex + faces[3 * i + k]);
            const float *p1 = b.pos.d + 4 * (baseIndex + faces[3 * i]);
            const float *p2 = b.pos.d + 4 * (baseIndex + faces[3 * i + 1]);
            const float *p3 = b.pos.d + 4 * (baseIndex + faces[3 * i + 2]);
            float Ux = p2[0] - p1[0], Uy = p2[1] - p1[1], Uz = p2[2] - p1[2];
            float Vx = p3[0] - p1[0], Vy = p3[1] - p1[1], Vz = p3[2] - p1[2];
            float nx = Uy * Vz - Uz * Vy, ny = Uz * Vx - Ux * Vz, nz = Ux * Vy - Uy * Vx;
            float lsq = nx * nx + ny * ny + nz * nz;
            float l = lsq ? sqrtf(lsq) : 0.0f;
            fpush(&b.tnrm, nx / l); fpush(&b.tnrm, ny / l); fpush(&b.tnrm, nz / l);
        }
        for (int i = 0; i < n_stretch_ints / 2; ++i) create_spring(&b, baseIndex + stretch[2 * i], baseIndex + stretch[2 * i + 1], ks);
        for (int i = 0; i < n_bend_ints / 2; ++i) create_spring(&b, baseIndex + bend[2 * i], baseIndex + bend[2 * i + 1], kb);
        for (int i = 0; i < n_shear_ints / 2; ++i) create_spring(&b, baseIndex + shear[2 * i], baseIndex + shear[2 * i + 1], ksh);
    } else {
        float mass = ptr[17] / (float)(dimx * dimz);
        float lower[3] = {initX, -initY, initZ};
        create_spring_grid(&b, lower, dimx, dimz, 1, radius, phase, ks, kb, ksh, 1.0f / mass);
    }
    if (flip_mesh) { /* softgym_cloth.h:137-152 */
        for (int j = 0; j < dimz - 1; ++j)
            for (int i = (dimx - 1) * 1 / 8; i < (dimx - 1) * 1 / 8 + 5; ++i) {
                int idx = j * (dimx - 1) + i;
                if (i != (dimx - 1) * 1 / 8 + 4) swap_int(&b.tris.d[idx * 6], &b.tris.d[idx * 6 + 1]);
                if (i != (dimx - 1) * 1 / 8) swap_int(&b.tris.d[idx * 6 + 3], &b.tris.d[idx * 6 + 4]);
            }
    }
    p->numSubsteps = 4;            /* softgym_cloth.h:154 */
    p->numIterations = 30;         /* :155 */
    p->dynamicFriction = 0.75f;    /* :157 */
    p->particleFriction = 1.0f;    /* :158 */
    p->damping = 1.0f;             /* :159 */
    p->sleepThreshold = 0.02f;     /* :160 */
    p->relaxationFactor = 1.0f;    /* :162 */
    p->shapeCollisionMargin = 0.04f; /* :163 */
    for (int k = 0; k < 3; ++k) { s->scene_lower[k] = -1.0f; s->scene_upper[k] = 1.0f; } /* :165-166 */
    p->radius = radius * 1.8f;     /* :169 */
    p->collisionDistance = 0.005f; /* :170 */

    /* ---- back in Init, main.cpp:844-884: derived params and bounds */
    if (p->solidRestDistance == 0.0f) p->solidRestDistance = p->radius;
    if (p->collisionDistance == 0.0f) p->collisionDistance = p->solidRestDistance * 0.5f;
    if (p->particleFriction == 0.0f) p->particleFriction = p->dynamicFriction * 0.1f;
    if (p->shapeCollisionMargin == 0.0f) p->shapeCollisionMargin = p->collisionDistance * 0.5f;

    s->n = b.pos.n / 4; s->m = b.slen.n; s->t = b.tris.n / 3;
    for (int i = 0; i < s->n; ++i)
        for (int k = 0; k < 3; ++k) {
            float v = b.pos.d[4 * i + k];
            if (v < s->scene_lower[k]) s->scene_lower[k] = v;
            if (v > s->scene_upper[k]) s->scene_upper[k] = v;
        }
    for (int k = 0; k < 3; ++k) { s->scene_lower[k] -= p->collisionDistance; s->scene_upper[k] += p->collisionDistance; }
    p->planes[0][0] = 0.0f; p->planes[0][1] = 1.0f; p->planes[0][2] = 0.0f; p->planes[0][3] = 0.0f; /* :882-884, tilt 0 */

    int n = s->n, m = s->m, t = s->t;
    s->pos = b.pos.d; s->vel = b.vel.d; s->phase = b.phase.d; s->sidx = b.sidx.d; s->slen = b.slen.d; s->sk = b.sk.d;
    s->tris = b.tris.d; s->tnrm = b.tnrm.d;
    if (!s->sidx) s->sidx = (int *)calloc(2, sizeof(int));
    s->rest = (float *)malloc(sizeof(float) * 4 * (n + 1));
    memcpy(s->rest, s->pos, sizeof(float) * 4 * n); /* main.cpp:971-973 */
    s->nrm = (float *)malloc(sizeof(float) * 4 * (n + 1));
    compute_normals(s->pos, s->tris, n, t, s->nrm);

    /* CSR adjacency, ascending spring id per particle */
    s->adj_off = (int *)calloc(n + 2, sizeof(int));
    s->adj_spr = (int *)malloc(sizeof(int) * (2 * m + 1));
    for (int e = 0; e < m; ++e) { s->adj_off[s->sidx[2 * e] + 1]++; s->adj_off[s->sidx[2 * e + 1] + 1]++; }
    for (int i = 0; i < n; ++i) s->adj_off[i + 1] += s->adj_off[i];
    int *fill = (int *)calloc(n + 1, sizeof(int));
    for (int e = 0; e < m; ++e)
        for (int k = 0; k < 2; ++k) {
            int i = s->sidx[2 * e + k];
            s->adj_spr[s->adj_off[i] + fill[i]++] = e;
        }
    free(fill);

    s->xp = (float *)malloc(sizeof(float) * 4 * (n + 1));
    s->xn = (float *)malloc(sizeof(float) * 4 * (n + 1));
    s->x0 = (float *)malloc(sizeof(float) * 4 * (n + 1));
    s->v0 = (float *)malloc(sizeof(float) * 3 * (n + 1));
    s->ncount = (int *)calloc(n + 1, sizeof(int));
    s->nlist = (int *)malloc(sizeof(int) * ORC_MAX_NEIGHBORS * (n + 1));
    s->cell_key = (int *)malloc(sizeof(int) * 4 * (n + 1));
    s->cell_order = (int *)malloc(sizeof(int) * (n + 1));
    s->ns = 0;
    s->smask = (unsigned *)calloc(n + 1, sizeof(unsigned));
    s->missed_shape_contacts = 0;
    s->accel_clamps = 0;
    s->degenerate_normals = 0;
#ifdef ORC_ALT_MAXACCEL_PER_FRAME
    s->vframe = (float *)calloc((size_t)3 * (n + 1), sizeof(float));
#endif
#ifdef ORC_ALT_CONTACT_PLANES
    s->splane = (float *)calloc((size_t)ORC_MAX_SHAPES * 4 * (n + 1), sizeof(float));
#endif
#ifdef ORC_ALT_FRICTION_POST
    s->lam = (float *)calloc((size_t)(ORC_MAX_NEIGHBORS + 8 + ORC_MAX_SHAPES) * (n + 1), sizeof(float));
#endif
#ifdef ORC_ALT_STIFFNESS_ITER
    for (int e = 0; e < m; ++e) { /* Mueller 2007 section 3.3: k' = 1 - (1 - k)^(1 / iterations); tethers keep their sign */
        float k = fabsf(s->sk[e]);
        float kk = (float)(1.0 - pow(1.0 - (double)(k > 1.0f ? 1.0f : k), 1.0 / (double)p->numIterations));
        s->sk[e] = s->sk[e] < 0.0f ? -kk : kk;
    }
#endif
    return 0;
}

/* ---------------------------------------------------------------- solver step */

static int cmp_int(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }

static __thread const int *g_sort_keys; /* qsort context: 3 ints per particle (thread-local: independent simulations
                                              may run on different threads, e.g. bench.py's CPU baseline) */
static int cmp_cell(const void *a, const void *b) {
    const int *ka = g_sort_keys + 3 * (*(const int *)a), *kb = g_sort_keys + 3 * (*(const int *)b);
    for (int k = 0; k < 3; ++k)
        if (ka[k] != kb[k]) return ka[k] < kb[k] ? -1 : 1;
    return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b);
}

/* lower bound over sorted cell_order by (cx,cy,cz) */
static int cell_lower_bound(const orc_sim *s, const int *keys, int cx, int cy, int cz) {
    int lo = 0, hi = s->n;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        const int *k = keys + 3 * s->cell_order[mid];
        int less = (k[0] != cx) ? (k[0] < cx) : (k[1] != cy) ? (k[1] < cy) : (k[2] < cz);
        if (less) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/*
 * MODEL switches -- sensitivity builds (oracle/Makefile liboracle_alt_<name>.so; tests/parity_table.py -> PARITY.md).
 * The solver step restates closed-source code, so every [I] below is a READING of NvFlex.h + Macklin 2014.  Each switch
 * replaces exactly ONE such reading by its most plausible alternative; the default build (no switch) is THE oracle the HIP
 * kernels are compared with bit for bit, the alternatives only measure how far a different reading would move a trajectory
 * (= the error bar this repo can state on "within 1e-4 of PyFleX" without PyFleX).  Never used as a parity oracle.
 *   -DORC_ALT_FRICTION_POST          contacts inside the iterations are frictionless; friction is applied once per substep
 *                                    after the position solve, with the normal correction each contact accumulated over
 *                                    the iterations as the Coulomb bound (default: per contact inside every iteration,
 *                                    bound = that iteration's penetration depth, Macklin 2014 section 6.1)
 *   -DORC_ALT_NEIGHBORS_BY_DISTANCE  a candidate list longer than maxNeighborsPerParticle keeps the 96 NEAREST
 *                                    (default: the 96 smallest ids)
 *   -DORC_ALT_SHAPE_END_POSE         kinematic spheres stand at their end-of-frame pose in every substep and carry the
 *                                    frame's mean velocity (default: linear sweep prev -> current over the substeps)
 *   -DORC_ALT_SLEEP_VELOCITY_ONLY    a particle below sleepThreshold gets zero velocity but keeps its new position
 *   -DORC_ALT_SLEEP_AT_PREDICT       "considered fixed" is applied at predict (x* = x when the gravity-integrated speed is
 *                                    below the threshold), nothing at finalize
 *   -DORC_ALT_NO_SLEEP               sleepThreshold ignored (how much the rule matters at all)
 *                                    (default: at finalize, position kept AND velocity zeroed)
 *   -DORC_ALT_APPLY_PER_TYPE         one applyDeltas after each constraint type -- springs, then particle contacts, then
 *                                    shapes, each seeing the previous type's result (default: one per iteration over all)
 *   -DORC_ALT_DAMPING_MULT           v = (v + h g) (1 - h damping)   (default: v += h (g - damping v))
 *   -DORC_ALT_STIFFNESS_ITER         spring stiffness made iteration-count independent, k' = 1 - (1 - k)^(1/iterations)
 *                                    (Mueller 2007 section 3.3)   (default: k used as is in every iteration)
 *   -DORC_ALT_SHAPE_EVERY_ITERATION  no collideShapes stage: every plane and every sphere is tested for every particle in
 *                                    every iteration (rounds 1-4 of this repo; default: only the candidates found once per
 *                                    substep within collisionDistance + shapeCollisionMargin, at most maxContactsPerParticle)
 *   -DORC_ALT_CONTACT_PLANES         a sphere candidate is FROZEN at collideShapes into its tangent plane at the predicted
 *                                    position -- the data model NvFlexGetContacts documents (NvFlex.h:1074-1080: "contact
 *                                    planes", "velocity of the contact point on the shape") -- and the iterations project
 *                                    on that plane (default: the sphere itself, normal re-derived from the current iterate)
 *   -DORC_ALT_COUNT_CANDIDATES       the Local-relaxation divisor counts every LISTED contact of the particle (particle
 *                                    candidates and shape candidates), violated or not (NvFlex.h:89 "divided by the
 *                                    particle's constraint count"; default: only the constraints that pushed this iteration)
 *   -DORC_ALT_NEIGHBORS_AT_START     particle-contact candidates are searched on the positions at the START of the substep
 *                                    (default: on the predicted positions, Macklin 2014 Algorithm 1 "N_i(x*_i)")
 *   -DORC_ALT_NO_MAXACCEL            the maxAcceleration clamp of finalize is skipped (how much the rule matters at all;
 *                                    NvFlex.h:112-113 "clamped to this value at the end of each step")
 *   -DORC_ALT_MAXACCEL_PER_FRAME     "each step" read as each NvFlexUpdateSolver call: the velocity change since the START OF
 *                                    THE FRAME is clamped to maxAcceleration * dt once, after the last substep (default: per
 *                                    substep, change since the substep's start clamped to maxAcceleration * dt / substeps)
 *   -DORC_ALT_MAXACCEL_POSITION      a clamped particle's position follows its clamped velocity, x = x0 + h v (default: the
 *                                    velocity alone is clamped, the position keeps what the iterations produced -- so the
 *                                    particle moved farther than v h says); combines with either clamp period
 *   -DORC_ALT_KINEMATIC_VELOCITY_KEPT  finalize leaves the velocity of an invMass-0 particle alone, so a particle the picker
 *                                    releases resumes with the velocity it had when it was pinned (default: zeroed while
 *                                    pinned = what v = (x* - x) / h gives for a particle the solver does not move)
 * The static-friction branch has no alternative worth a build: with mu_s <= mu_k (0 <= 0.75 for shapes, 1 = 1 between
 * particles) "full stick below mu_s * depth" and "clamp to mu_k * depth" give the same scale for every input
 * (tests/test_oracle_cpu.py::test_static_friction_branch_is_redundant).
 */
#if defined(ORC_ALT_SLEEP_VELOCITY_ONLY) + defined(ORC_ALT_SLEEP_AT_PREDICT) + defined(ORC_ALT_NO_SLEEP) > 1
#error "one sleep alternative at a time"
#endif
#if defined(ORC_ALT_NO_MAXACCEL) && (defined(ORC_ALT_MAXACCEL_PER_FRAME) || defined(ORC_ALT_MAXACCEL_POSITION))
#error "no clamp at all excludes the other clamp alternatives"
#endif
#if defined(ORC_ALT_CONTACT_PLANES) && (defined(ORC_ALT_SHAPE_EVERY_ITERATION) || defined(ORC_ALT_FRICTION_POST))
#error "contact planes need the candidate stage and the in-iteration friction"
#endif
#ifdef ORC_ALT_COUNT_CANDIDATES
#define ORC_COUNT_LISTED(cnt) ((cnt)++)
#else
#define ORC_COUNT_LISTED(cnt) ((void)0)
#endif

/*
 * Particle-contact candidates, built once per substep on the predicted positions
 * ([D] stage order NvFlex.h:200-204 vs per-iteration :211-215):
 *  pair (i,j), i != j, |x*_i - x*_j|^2 < radius^2 (particleCollisionMargin = 0, NvFlex.h:146),
 *  same group needs eNvFlexPhaseSelfCollide on both (NvFlex.h:165),
 *  eNvFlexPhaseSelfCollideFilter on either drops pairs with |rest_i - rest_j|^2 < radius^2 (NvFlex.h:166,564-565),
 *  list = ascending j, truncated to the 96 smallest (maxNeighborsPerParticle, main.cpp:826)   [I: ordering/truncation]
 */
static void find_neighbors(orc_sim *s_) {
    /* [I] WHICH positions the search looks at: the predicted ones (Macklin 2014, Algorithm 1: "find neighboring particles
       N_i(x*_i)" after the prediction); ORC_ALT_NEIGHBORS_AT_START searches the positions at the start of the substep. */
#ifdef ORC_ALT_NEIGHBORS_AT_START
    orc_sim view = *s_;
    view.xp = s_->x0;
    orc_sim *s = &view;
#else
    orc_sim *s = s_;
#endif
    const int n = s->n;
    const float r = s->p.radius + s->p.particleCollisionMargin;
    const float r2 = r * r;
    const float inv = 1.0f / r;
    int *keys = s->cell_key;
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) keys[3 * i + k] = (int)floorf(s->xp[4 * i + k] * inv);
        s->cell_order[i] = i;
    }
    g_sort_keys = keys;
    qsort(s->cell_order, n, sizeof(int), cmp_cell);
    int tmp[4096];
    for (int i = 0; i < n; ++i) {
        int cnt = 0, total_cap = 4096;
        const float *xi = s->xp + 4 * i;
        int gi = s->phase[i] & PH_GROUP_MASK;
        for (int dx = -1; dx <= 1; ++dx)
            for (int dy = -1; dy <= 1; ++dy) {
                /* the three dz cells are contiguous in the sort order */
                int cx = keys[3 * i] + dx, cy = keys[3 * i + 1] + dy, cz0 = keys[3 * i + 2] - 1;
                int q = cell_lower_bound(s, keys, cx, cy, cz0);
                for (; q < n; ++q) {
                    int j = s->cell_order[q];
                    const int *kj = keys + 3 * j;
                    if (kj[0] != cx || kj[1] != cy || kj[2] > cz0 + 2) break;
                    if (j == i) continue;
                    const float *xj = s->xp + 4 * j;
                    float ddx = xi[0] - xj[0], ddy = xi[1] - xj[1], ddz = xi[2] - xj[2];
                    float d2 = ddx * ddx + ddy * ddy + ddz * ddz;
                    if (!(d2 < r2)) continue;
                    int gj = s->phase[j] & PH_GROUP_MASK;
                    if (gi == gj) {
                        if (!((s->phase[i] & PH_SELF_COLLIDE) && (s->phase[j] & PH_SELF_COLLIDE))) continue;
                        if ((s->phase[i] | s->phase[j]) & PH_SELF_COLLIDE_FILTER) {
                            const float *ri = s->rest + 4 * i, *rj = s->rest + 4 * j;
                            float ex = ri[0] - rj[0], ey = ri[1] - rj[1], ez = ri[2] - rj[2];
                            float e2 = ex * ex + ey * ey + ez * ez;
                            if (e2 < r2) continue;
                        }
                    }
                    if (cnt < total_cap) tmp[cnt++] = j;
                }
            }
        if (cnt > s_->max_list) s_->max_list = cnt; /* before truncation */
#ifdef ORC_ALT_NEIGHBORS_BY_DISTANCE
        if (cnt > s->p.maxNeighbors) { /* keep the nearest: selection by (distance^2, id), then back to ascending id */
            for (int a = 0; a < s->p.maxNeighbors; ++a) {
                int best = a;
                float bd = 0.0f;
                for (int b = a; b < cnt; ++b) {
                    const float *xj = s->xp + 4 * tmp[b];
                    float ddx = xi[0] - xj[0], ddy = xi[1] - xj[1], ddz = xi[2] - xj[2];
                    float d2 = ddx * ddx + ddy * ddy + ddz * ddz;
                    if (b == a || d2 < bd || (d2 == bd && tmp[b] < tmp[best])) { best = b; bd = d2; }
                }
                int t_ = tmp[a]; tmp[a] = tmp[best]; tmp[best] = t_;
            }
            cnt = s->p.maxNeighbors;
        }
#endif
        qsort(tmp, cnt, sizeof(int), cmp_int);
        if (cnt > s->p.maxNeighbors) cnt = s->p.maxNeighbors;
        s->ncount[i] = cnt;
        memcpy(s->nlist + (size_t)ORC_MAX_NEIGHBORS * i, tmp, sizeof(int) * cnt);
    }
}

/* Reciprocal square root used for every length inside the constraint sweeps.  [I] The closed-source reference certainly uses
   the hardware approximation of its GPU here; so does the product: gfx950's v_rsq_f32 (1 ulp) on max(x, FLT_MIN).  The
   instruction's result is a pure function of the input bits, but no formula for it is published, so the oracle carries it as
   DATA: oracle/v_rsq_f32_gfx950.npz holds, for each of the 2^24 inputs (exponent parity, mantissa) in [1, 4), the difference
   in ulps (-1, 0, +1) between what the chip returns and float32(1 / sqrt(float64(x))) -- IEEE double sqrt and division and one
   rounding, the same bits on every host -- dumped on an MI355X through the product's own fs_eval_rsqrt by
   tests/golden/make_rsq_table.py.  Other exponents only rescale the result by a power of two (checked over every normal
   exponent by the same script and by tests/test_parity_gpu.py::test_hw_rsqrt_matches_committed_table, which re-reads the whole
   table from the box it runs on).  The Python front end hands the table to orc_set_rsqrt_table(); without it the oracle
   refuses to step. */
#include <stdint.h>
#include <stdio.h>
/* Build switches for BOUNDING the arithmetic [I] choices (tests/test_oracle_cpu.py::test_approximations_stay_within_1e-4
   of exact math; never used as the parity oracle):
     -DORC_EXACT_RSQRT   lengths through correctly rounded sqrtf() and IEEE divisions instead of the hardware reciprocal root
     -DORC_NEWTON_RSQRT  the reciprocal root of rounds 1-3: integer seed + three Newton steps in IEEE mul / fma (~2 ulp)
     -DORC_NO_FMA        every multiply-add rounded twice (a*b, then +c), as a compiler without contraction emits it
     -DORC_EXACT_MATH    ORC_EXACT_RSQRT + ORC_NO_FMA: the plain IEEE restatement of the same step */
#ifdef ORC_EXACT_MATH
#define ORC_EXACT_RSQRT 1
#define ORC_NO_FMA 1
#endif
#ifdef ORC_NO_FMA
#define ORC_FMA(a, b, c) ((a) * (b) + (c))
#else
#define ORC_FMA(a, b, c) fmaf((a), (b), (c))
#endif
static const unsigned char *g_rsq_delta; /* 2^24 two-bit fields (delta + 2), four per byte, index = parity << 23 | mantissa */
void orc_set_rsqrt_table(const unsigned char *packed_2bit) { g_rsq_delta = packed_2bit; }
static inline float orc_hw_rsqrt(float x) {
    union { float f; uint32_t u; } v, r, sc;
    v.f = (x >= FLT_MIN) ? x : FLT_MIN;          /* v_max_f32(x, FLT_MIN): zero, denormals (and NaN) take the clamp */
    if (v.u >= 0x7f800000u) return 0.0f;         /* +inf -> +0 like the instruction */
    const int e = (int)(v.u >> 23) - 127;
    const uint32_t par = (uint32_t)e & 1u, man = v.u & 0x7fffffu, idx = par << 23 | man;
    v.u = (127u + par) << 23 | man;              /* the same mantissa in [1, 4) */
    r.f = (float)(1.0 / sqrt((double)v.f));
    r.u += (uint32_t)((int)((g_rsq_delta[idx >> 2] >> (2 * (idx & 3u))) & 3u) - 2);
    sc.u = (uint32_t)(127 - (e - (int)par) / 2) << 23; /* 2^-((e - parity) / 2): exact rescaling, result stays normal */
    return r.f * sc.f;
}
int orc_eval_rsqrt(const float *x, float *y, int n) { /* white box: what the constraint sweeps use, for n values */
    if (!g_rsq_delta) return -1;
    for (int i = 0; i < n; ++i) y[i] = orc_hw_rsqrt(x[i]);
    return 0;
}
#if defined(ORC_EXACT_RSQRT)
static inline float orc_rsqrt(float x) { return 1.0f / sqrtf(x); }
#define ORC_LEN(l2, inv) ((void)(inv), sqrtf(l2))            /* |e| */
#define ORC_OVER_LEN(a, len, inv) ((void)(inv), (a) / (len)) /* a / |e| */
#define ORC_NEEDS_TABLE 0
#elif defined(ORC_NEWTON_RSQRT)
static inline float orc_rsqrt(float x) {
    union { float f; uint32_t u; } v;
    v.f = x;
    v.u = 0x5f3759dfu - (v.u >> 1);
    float y = v.f;
    const float xh = 0.5f * x;
    y = y * ORC_FMA(-(xh * y), y, 1.5f);
    y = y * ORC_FMA(-(xh * y), y, 1.5f);
    y = y * ORC_FMA(-(xh * y), y, 1.5f);
    return y;
}
#define ORC_LEN(l2, inv) ((l2) * (inv))
#define ORC_OVER_LEN(a, len, inv) ((a) * (inv))
#define ORC_NEEDS_TABLE 0
#else
#define orc_rsqrt orc_hw_rsqrt
#define ORC_LEN(l2, inv) ((l2) * (inv))
#define ORC_OVER_LEN(a, len, inv) ((a) * (inv))
#define ORC_NEEDS_TABLE 1
#endif

/* a . b with two fused multiply-adds -- the constraint sweeps use fmaf exactly where the specification says so (the
   build keeps -ffp-contract=off, so nothing else is ever fused); the HIP kernels use v_fma_f32 at the same places. */
static inline float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return ORC_FMA(az, bz, ORC_FMA(ay, by, ax * bx));
}

/* friction on a contact: t = tangential relative displacement since substep start (length tl = tl2 * inv_tl),
   pen = penetration depth.  [I] Macklin 2014 section 6.1: full stick below mu_s*pen, else scaled by
   min(mu_k*pen/|t|, 1).  Returns the scale applied to t. */
static inline float friction_scale(float tl, float inv_tl, float pen, float mu_s, float mu_k) {
    if (tl < mu_s * pen) return 1.0f;
    float lim = mu_k * pen;
    return (tl > lim) ? ORC_OVER_LEN(lim, tl, inv_tl) : 1.0f;
}

/* which constraint types one Jacobi pass accumulates (the default iteration = one pass over all of them) */
#define ORC_T_SPRINGS 1
#define ORC_T_PARTICLES 2
#define ORC_T_SHAPES 4
#define ORC_T_ALL 7

#ifdef ORC_ALT_FRICTION_POST
#define ORC_LAM_STRIDE (ORC_MAX_NEIGHBORS + 8 + ORC_MAX_SHAPES) /* per particle: neighbour slots, planes, spheres */
#define ORC_MU(x) 0.0f /* contacts inside the iterations are frictionless */
#else
#define ORC_MU(x) (x)
#endif

/* One Jacobi pass with local relaxation (NvFlex.h:86-90,152-153) over the constraint types in `types`: xp -> xn. */
static void jacobi_pass(orc_sim *s, const float *xp, float *xn, const float *x0, float (*sc)[3], float (*sd)[3], int types) {
    const orc_params *p = &s->p;
    const int n = s->n;
    const float restd = p->solidRestDistance, restd2 = restd * restd;
    const float cd = p->collisionDistance;
    const float mu_pp = ORC_MU(p->particleFriction);
    for (int i = 0; i < n; ++i) {
        const float wi = xp[4 * i + 3];
        const float xi0 = xp[4 * i], xi1 = xp[4 * i + 1], xi2 = xp[4 * i + 2];
        if (!(wi > 0.0f)) { xn[4 * i] = xi0; xn[4 * i + 1] = xi1; xn[4 * i + 2] = xi2; xn[4 * i + 3] = wi; continue; }
        float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f;
        int cnt = 0;
#ifdef ORC_ALT_FRICTION_POST
        float cn_now[ORC_LAM_STRIDE];
        memset(cn_now, 0, sizeof(cn_now));
#endif
        /* 4a. distance constraints (NvFlex.h:656-667), ascending spring id */
        if (types & ORC_T_SPRINGS)
        for (int a = s->adj_off[i]; a < s->adj_off[i + 1]; ++a) {
            int e = s->adj_spr[a];
            int j = (s->sidx[2 * e] == i) ? s->sidx[2 * e + 1] : s->sidx[2 * e];
            const float wj = xp[4 * j + 3];
            float ex = xi0 - xp[4 * j], ey = xi1 - xp[4 * j + 1], ez = xi2 - xp[4 * j + 2];
            float l2 = dot3(ex, ey, ez, ex, ey, ez);
            float inv_len = orc_rsqrt(l2);
            float len = ORC_LEN(l2, inv_len);
            if (!(len > 0.0f)) continue;
            float C = len - s->slen[e];
            float k = s->sk[e];
            if (k < 0.0f) { if (!(C > 0.0f)) continue; k = -k; } /* tether: unilateral */
            float ratio = wi / (wi + wj);
            float sc_ = (k * ratio) * ORC_OVER_LEN(C, len, inv_len);
            d0 = ORC_FMA(-ex, sc_, d0); d1 = ORC_FMA(-ey, sc_, d1); d2 = ORC_FMA(-ez, sc_, d2);
            cnt++;
        }
        /* 4b. particle-particle contacts (NvFlex.h:101 solidRestDistance, :107 particleFriction, :108 inelastic) */
        const float ri0 = xi0 - x0[4 * i], ri1 = xi1 - x0[4 * i + 1], ri2 = xi2 - x0[4 * i + 2];
        if (types & ORC_T_PARTICLES)
        for (int a = 0; a < s->ncount[i]; ++a) {
            int j = s->nlist[(size_t)ORC_MAX_NEIGHBORS * i + a];
            const float wj = xp[4 * j + 3];
            float ex = xi0 - xp[4 * j], ey = xi1 - xp[4 * j + 1], ez = xi2 - xp[4 * j + 2];
            float l2 = dot3(ex, ey, ez, ex, ey, ez);
            if (!(l2 < restd2)) { ORC_COUNT_LISTED(cnt); continue; }
            float inv = orc_rsqrt(l2);
            float dist = ORC_LEN(l2, inv);
            float nx, ny, nz;
            if (dist > 0.0f) { nx = ORC_OVER_LEN(ex, dist, inv); ny = ORC_OVER_LEN(ey, dist, inv); nz = ORC_OVER_LEN(ez, dist, inv); }
            else { nx = 0.0f; ny = 1.0f; nz = 0.0f; s->degenerate_normals++; /* (white box only) */ }
            float pen = restd - dist;
            float ratio = wi / (wi + wj);
            float cn = pen * ratio;
            float c0 = nx * cn, c1 = ny * cn, c2 = nz * cn;
            if (mu_pp > 0.0f) {
                float rx = ri0 - (xp[4 * j] - x0[4 * j]);
                float ry = ri1 - (xp[4 * j + 1] - x0[4 * j + 1]);
                float rz = ri2 - (xp[4 * j + 2] - x0[4 * j + 2]);
                float rn = dot3(rx, ry, rz, nx, ny, nz);
                float tx = ORC_FMA(-nx, rn, rx), ty = ORC_FMA(-ny, rn, ry), tz = ORC_FMA(-nz, rn, rz);
                float tl2 = dot3(tx, ty, tz, tx, ty, tz);
                if (tl2 > 0.0f) {
                    float inv_tl = orc_rsqrt(tl2);
                    float tl = ORC_LEN(tl2, inv_tl);
                    float fs = friction_scale(tl, inv_tl, pen, mu_pp, mu_pp) * ratio;
                    c0 = ORC_FMA(-tx, fs, c0); c1 = ORC_FMA(-ty, fs, c1); c2 = ORC_FMA(-tz, fs, c2);
                }
            }
            d0 = d0 + c0; d1 = d1 + c1; d2 = d2 + c2;
            cnt++;
#ifdef ORC_ALT_FRICTION_POST
            cn_now[a] = cn;
#endif
        }
        /* 4c. planes (NvFlex.h:145 collisionDistance, :149 plane form, :105-106 friction): the candidates of collideShapes */
        const unsigned smask = s->smask[i];
        if (types & ORC_T_SHAPES)
        for (int q = 0; q < p->numPlanes; ++q) {
            const float *pl = p->planes[q];
            float sdist = dot3(pl[0], pl[1], pl[2], xi0, xi1, xi2) + pl[3];
            if (!((smask >> q) & 1u)) { if (sdist < cd) s->missed_shape_contacts++; continue; } /* (white box only) */
            if (!(sdist < cd)) { ORC_COUNT_LISTED(cnt); continue; }
            float pen = cd - sdist;
            float c0 = pl[0] * pen, c1 = pl[1] * pen, c2 = pl[2] * pen;
#ifndef ORC_ALT_FRICTION_POST
            float rn = dot3(ri0, ri1, ri2, pl[0], pl[1], pl[2]);
            float tx = ORC_FMA(-pl[0], rn, ri0), ty = ORC_FMA(-pl[1], rn, ri1), tz = ORC_FMA(-pl[2], rn, ri2);
            float tl2 = dot3(tx, ty, tz, tx, ty, tz);
            if (tl2 > 0.0f) {
                float inv_tl = orc_rsqrt(tl2);
                float tl = ORC_LEN(tl2, inv_tl);
                float fs = friction_scale(tl, inv_tl, pen, p->staticFriction, p->dynamicFriction);
                c0 = ORC_FMA(-tx, fs, c0); c1 = ORC_FMA(-ty, fs, c1); c2 = ORC_FMA(-tz, fs, c2);
            }
#else
            cn_now[ORC_MAX_NEIGHBORS + q] = pen;
#endif
            d0 = d0 + c0; d1 = d1 + c1; d2 = d2 + c2;
            cnt++;
        }
        /* 4d. kinematic spheres (NvFlex.h:941-987), all channels set so every particle collides (NvFlex.h:163,965) */
        if (types & ORC_T_SHAPES)
        for (int q = 0; q < s->ns; ++q) {
#ifdef ORC_ALT_CONTACT_PLANES
            if (!((smask >> (ORC_SPHERE_BIT + q)) & 1u)) continue;
            const float *tp = s->splane + ((size_t)ORC_MAX_SHAPES * i + q) * 4; /* frozen tangent plane of this candidate */
            const float nx = tp[0], ny = tp[1], nz = tp[2];
            float sdist = dot3(nx, ny, nz, xi0, xi1, xi2) + tp[3];
            if (!(sdist < cd)) { ORC_COUNT_LISTED(cnt); continue; }
            float pen = cd - sdist;
#else
            float ex = xi0 - sc[q][0], ey = xi1 - sc[q][1], ez = xi2 - sc[q][2];
            float l2 = dot3(ex, ey, ez, ex, ey, ez);
            float lim = s->sh_radius[q] + cd;
            if (!((smask >> (ORC_SPHERE_BIT + q)) & 1u)) { if (l2 < lim * lim) s->missed_shape_contacts++; continue; } /* (white box only) */
            if (!(l2 < lim * lim)) { ORC_COUNT_LISTED(cnt); continue; }
            float inv = orc_rsqrt(l2);
            float dist = ORC_LEN(l2, inv);
            float nx, ny, nz;
            if (dist > 0.0f) { nx = ORC_OVER_LEN(ex, dist, inv); ny = ORC_OVER_LEN(ey, dist, inv); nz = ORC_OVER_LEN(ez, dist, inv); }
            else { nx = 0.0f; ny = 1.0f; nz = 0.0f; s->degenerate_normals++; /* (white box only) */ }
            float pen = lim - dist;
#endif
            float c0 = nx * pen, c1 = ny * pen, c2 = nz * pen;
#ifndef ORC_ALT_FRICTION_POST
            float rx = ri0 - sd[q][0], ry = ri1 - sd[q][1], rz = ri2 - sd[q][2];
            float rn = dot3(rx, ry, rz, nx, ny, nz);
            float tx = ORC_FMA(-nx, rn, rx), ty = ORC_FMA(-ny, rn, ry), tz = ORC_FMA(-nz, rn, rz);
            float tl2 = dot3(tx, ty, tz, tx, ty, tz);
            if (tl2 > 0.0f) {
                float inv_tl = orc_rsqrt(tl2);
                float tl = ORC_LEN(tl2, inv_tl);
                float fs = friction_scale(tl, inv_tl, pen, p->staticFriction, p->dynamicFriction);
                c0 = ORC_FMA(-tx, fs, c0); c1 = ORC_FMA(-ty, fs, c1); c2 = ORC_FMA(-tz, fs, c2);
            }
#else
            cn_now[ORC_MAX_NEIGHBORS + 8 + q] = pen;
#endif
            d0 = d0 + c0; d1 = d1 + c1; d2 = d2 + c2;
            cnt++;
        }
        /* 4e. applyDeltas, eNvFlexRelaxationLocal: delta / constraint count * relaxationFactor */
        if (cnt > 0) {
            float sc_ = p->relaxationFactor / (float)cnt;
            xn[4 * i] = ORC_FMA(d0, sc_, xi0); xn[4 * i + 1] = ORC_FMA(d1, sc_, xi1); xn[4 * i + 2] = ORC_FMA(d2, sc_, xi2);
#ifdef ORC_ALT_FRICTION_POST
            float *lam = s->lam + (size_t)ORC_LAM_STRIDE * i; /* normal correction each contact really applied so far */
            for (int c = 0; c < ORC_LAM_STRIDE; ++c) lam[c] += cn_now[c] * sc_;
#endif
        } else { xn[4 * i] = xi0; xn[4 * i + 1] = xi1; xn[4 * i + 2] = xi2; }
        xn[4 * i + 3] = wi;
    }
}

#ifdef ORC_ALT_FRICTION_POST
/* Friction after the position solve: one Jacobi pass in which every contact that pushed during the iterations removes
   tangential relative displacement (since the substep start) up to mu x the normal correction it accumulated. */
static void friction_pass(orc_sim *s, const float *xp, float *xn, const float *x0, float (*sc)[3], float (*sd)[3]) {
    const orc_params *p = &s->p;
    const int n = s->n;
    for (int i = 0; i < n; ++i) {
        const float wi = xp[4 * i + 3];
        const float xi0 = xp[4 * i], xi1 = xp[4 * i + 1], xi2 = xp[4 * i + 2];
        xn[4 * i] = xi0; xn[4 * i + 1] = xi1; xn[4 * i + 2] = xi2; xn[4 * i + 3] = wi;
        if (!(wi > 0.0f)) continue;
        const float *lam = s->lam + (size_t)ORC_LAM_STRIDE * i;
        const float ri0 = xi0 - x0[4 * i], ri1 = xi1 - x0[4 * i + 1], ri2 = xi2 - x0[4 * i + 2];
        float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f;
        int cnt = 0;
        for (int c = 0; c < ORC_LAM_STRIDE; ++c) {
            if (!(lam[c] > 0.0f)) continue;
            float nx, ny, nz, rx, ry, rz, mu_s, mu_k, ratio = 1.0f;
            if (c < ORC_MAX_NEIGHBORS) {
                int j = s->nlist[(size_t)ORC_MAX_NEIGHBORS * i + c];
                float ex = xi0 - xp[4 * j], ey = xi1 - xp[4 * j + 1], ez = xi2 - xp[4 * j + 2];
                float l2 = dot3(ex, ey, ez, ex, ey, ez);
                if (l2 > 0.0f) { float inv = orc_rsqrt(l2); nx = ex * inv; ny = ey * inv; nz = ez * inv; }
                else { nx = 0.0f; ny = 1.0f; nz = 0.0f; }
                rx = ri0 - (xp[4 * j] - x0[4 * j]); ry = ri1 - (xp[4 * j + 1] - x0[4 * j + 1]); rz = ri2 - (xp[4 * j + 2] - x0[4 * j + 2]);
                mu_s = mu_k = p->particleFriction;
                ratio = wi / (wi + xp[4 * j + 3]);
            } else if (c < ORC_MAX_NEIGHBORS + 8) {
                const float *pl = p->planes[c - ORC_MAX_NEIGHBORS];
                nx = pl[0]; ny = pl[1]; nz = pl[2];
                rx = ri0; ry = ri1; rz = ri2;
                mu_s = p->staticFriction; mu_k = p->dynamicFriction;
            } else {
                int q = c - ORC_MAX_NEIGHBORS - 8;
                float ex = xi0 - sc[q][0], ey = xi1 - sc[q][1], ez = xi2 - sc[q][2];
                float l2 = dot3(ex, ey, ez, ex, ey, ez);
                if (l2 > 0.0f) { float inv = orc_rsqrt(l2); nx = ex * inv; ny = ey * inv; nz = ez * inv; }
                else { nx = 0.0f; ny = 1.0f; nz = 0.0f; }
                rx = ri0 - sd[q][0]; ry = ri1 - sd[q][1]; rz = ri2 - sd[q][2];
                mu_s = p->staticFriction; mu_k = p->dynamicFriction;
            }
            if (!(mu_k > 0.0f)) continue;
            float rn = dot3(rx, ry, rz, nx, ny, nz);
            float tx = ORC_FMA(-nx, rn, rx), ty = ORC_FMA(-ny, rn, ry), tz = ORC_FMA(-nz, rn, rz);
            float tl2 = dot3(tx, ty, tz, tx, ty, tz);
            if (!(tl2 > 0.0f)) continue;
            float inv_tl = orc_rsqrt(tl2);
            float tl = ORC_LEN(tl2, inv_tl);
            float fs = friction_scale(tl, inv_tl, lam[c], mu_s, mu_k) * ratio;
            d0 = ORC_FMA(-tx, fs, d0); d1 = ORC_FMA(-ty, fs, d1); d2 = ORC_FMA(-tz, fs, d2);
            cnt++;
        }
        if (cnt > 0) {
            float sc_ = p->relaxationFactor / (float)cnt;
            xn[4 * i] = ORC_FMA(d0, sc_, xi0); xn[4 * i + 1] = ORC_FMA(d1, sc_, xi1); xn[4 * i + 2] = ORC_FMA(d2, sc_, xi2);
        }
    }
}
#endif

/*
 * 3. collideShapes -- once per substep, before the iterations ([D] stage order NvFlex.h:205 vs the per-iteration timers
 *    :211-215; "will include all contact planes generated within NvFlexParams::shapeCollisionMargin", NvFlex.h:1074).
 *    Per dynamic particle: the planes and kinematic spheres whose surface is closer to the PREDICTED position than
 *    collisionDistance + shapeCollisionMargin (NvFlex.h:145 "distance particles maintain against shapes", :147 "increases
 *    the radius used during contact finding against kinematic shapes"; softgym_cloth.h:163,170: 0.005 + 0.04), at most
 *    maxContactsPerParticle of them (NvFlex.h:361; main.cpp:828: 6) -- planes in index order, then spheres in index order
 *    [I: which ones survive the cap; FlingBot's scenes have 1 + 2 <= 6, so the cap never bites].  A sphere is taken at the
 *    pose the iterations of this substep use (end of its sweep over the substep) [I].  The iterations test ONLY this set.
 */
static void collide_shapes(orc_sim *s, float (*sc)[3]) {
    const orc_params *p = &s->p;
    const float reach = p->collisionDistance + p->shapeCollisionMargin;
    for (int i = 0; i < s->n; ++i) {
        unsigned mask = 0u;
        int listed = 0;
#ifdef ORC_ALT_SHAPE_EVERY_ITERATION
        mask = ~0u; (void)listed; (void)reach; (void)sc;
#else
        const float *x = s->xp + 4 * i; /* (kinematic particles get a list as well; nothing ever reads it) */
        {
            for (int q = 0; q < p->numPlanes && listed < p->maxContacts; ++q) {
                const float *pl = p->planes[q];
                float sdist = dot3(pl[0], pl[1], pl[2], x[0], x[1], x[2]) + pl[3];
                if (sdist < reach) { mask |= 1u << q; listed++; }
            }
            for (int q = 0; q < s->ns && listed < p->maxContacts; ++q) {
                float ex = x[0] - sc[q][0], ey = x[1] - sc[q][1], ez = x[2] - sc[q][2];
                float l2 = dot3(ex, ey, ez, ex, ey, ez);
                float lim = s->sh_radius[q] + reach;
                if (!(l2 < lim * lim)) continue;
                mask |= 1u << (ORC_SPHERE_BIT + q); listed++;
#ifdef ORC_ALT_CONTACT_PLANES
                float inv = orc_rsqrt(l2);
                float dist = ORC_LEN(l2, inv);
                float nx = 0.0f, ny = 1.0f, nz = 0.0f;
                if (dist > 0.0f) { nx = ORC_OVER_LEN(ex, dist, inv); ny = ORC_OVER_LEN(ey, dist, inv); nz = ORC_OVER_LEN(ez, dist, inv); }
                float *tp = s->splane + ((size_t)ORC_MAX_SHAPES * i + q) * 4; /* signed distance to it: n . (x - c) - r */
                tp[0] = nx; tp[1] = ny; tp[2] = nz;
                tp[3] = -(dot3(nx, ny, nz, sc[q][0], sc[q][1], sc[q][2]) + s->sh_radius[q]);
#endif
            }
        }
#endif
        s->smask[i] = mask;
    }
}

static void substep(orc_sim *s, int sub, float h, float inv_h) {
    const orc_params *p = &s->p;
    const int n = s->n;
    float *xp = s->xp, *xn = s->xn, *x0 = s->x0, *v0 = s->v0;
    const float S = (float)p->numSubsteps;

#ifdef ORC_ALT_MAXACCEL_PER_FRAME
    if (sub == 0) memcpy(s->vframe, s->vel, sizeof(float) * 3 * n);
#endif
    /* 1. predict (NvFlex.h:99 gravity, :117 damping [form I], :545 invMass 0 = kinematic) */
    for (int i = 0; i < n; ++i) {
        const float w = s->pos[4 * i + 3];
        for (int k = 0; k < 3; ++k) { x0[4 * i + k] = s->pos[4 * i + k]; v0[3 * i + k] = s->vel[3 * i + k]; }
        x0[4 * i + 3] = w;
        xp[4 * i + 3] = w;
        if (w > 0.0f) {
#ifdef ORC_ALT_SLEEP_AT_PREDICT
            float vv[3];
#endif
            for (int k = 0; k < 3; ++k) {
                float v = s->vel[3 * i + k];
#ifdef ORC_ALT_DAMPING_MULT
                v = (v + h * p->gravity[k]) * (1.0f - h * p->damping);
#else
                v = v + h * (p->gravity[k] - p->damping * v);
#endif
                xp[4 * i + k] = x0[4 * i + k] + h * v;
#ifdef ORC_ALT_SLEEP_AT_PREDICT
                vv[k] = v;
#endif
            }
#ifdef ORC_ALT_SLEEP_AT_PREDICT
            if (vv[0] * vv[0] + vv[1] * vv[1] + vv[2] * vv[2] < p->sleepThreshold * p->sleepThreshold)
                for (int k = 0; k < 3; ++k) xp[4 * i + k] = x0[4 * i + k];
#endif
        } else {
            for (int k = 0; k < 3; ++k) xp[4 * i + k] = x0[4 * i + k];
        }
    }

    /* 2. neighbours (once per substep) */
    find_neighbors(s);

    /* 3a. shapes at this substep: linear sweep prev -> current over the frame [I] (prev transforms: NvFlex.h:981-982) */
    float sc[ORC_MAX_SHAPES][3], sd[ORC_MAX_SHAPES][3];
    for (int q = 0; q < s->ns; ++q) {
        float a1 = (float)(sub + 1) / S, a0 = (float)sub / S;
        for (int k = 0; k < 3; ++k) {
            float dlt = s->sh_pos[q][k] - s->sh_prev[q][k];
#ifdef ORC_ALT_SHAPE_END_POSE
            (void)a1; (void)a0;
            sc[q][k] = s->sh_pos[q][k];
            sd[q][k] = dlt / S;
#else
            float c1 = s->sh_prev[q][k] + dlt * a1;
            float c0 = s->sh_prev[q][k] + dlt * a0;
            sc[q][k] = c1;
            sd[q][k] = c1 - c0;
#endif
        }
    }

    collide_shapes(s, sc);

#ifdef ORC_ALT_FRICTION_POST
    memset(s->lam, 0, sizeof(float) * (size_t)ORC_LAM_STRIDE * n);
#endif
    /* 4. Jacobi iterations with local relaxation (NvFlex.h:86-90,152-153); one applyDeltas per iteration over all
          constraint types [I: the single applyDeltas timer, NvFlex.h:215] */
    for (int it = 0; it < p->numIterations; ++it) {
#ifdef ORC_ALT_APPLY_PER_TYPE
        static const int order[3] = {ORC_T_SPRINGS, ORC_T_PARTICLES, ORC_T_SHAPES};
        for (int ty = 0; ty < 3; ++ty) {
            jacobi_pass(s, xp, xn, x0, sc, sd, order[ty]);
            float *t = xp; xp = xn; xn = t;
        }
#else
        jacobi_pass(s, xp, xn, x0, sc, sd, ORC_T_ALL);
        float *t = xp; xp = xn; xn = t;
#endif
    }
#ifdef ORC_ALT_FRICTION_POST
    friction_pass(s, xp, xn, x0, sc, sd);
    { float *t = xp; xp = xn; xn = t; }
#endif
    s->xp = xp; s->xn = xn;

    /* 5. finalize: velocity from displacement, maxAcceleration / maxSpeed clamps (NvFlex.h:112-113), sleeping
          (NvFlex.h:110 "velocity magnitude < threshold => considered fixed"; Macklin 2014 section 4.5 freezes the
          position when the particle moved less than the threshold; we also zero the velocity [I]) */
    /* [I] "at the end of each step" (NvFlex.h:112-113): the default reads a step as a substep -- the velocity change since the
       substep's start is clamped to maxAcceleration * h, and only the VELOCITY is clamped (the position keeps what the
       iterations produced).  ORC_ALT_MAXACCEL_PER_FRAME / _POSITION / ORC_ALT_NO_MAXACCEL are the other readings. */
#if defined(ORC_ALT_MAXACCEL_PER_FRAME)
    const float maxdv = p->maxAcceleration * p->dt;
    const int clamp_now = (sub == p->numSubsteps - 1);
#elif defined(ORC_ALT_NO_MAXACCEL)
    const float maxdv = FLT_MAX;
    const int clamp_now = 0;
#else
    const float maxdv = p->maxAcceleration * h;
    const int clamp_now = 1;
#endif
    const float thr2 = p->sleepThreshold * p->sleepThreshold;
    for (int i = 0; i < n; ++i) {
        const float w = s->pos[4 * i + 3];
#ifdef ORC_ALT_KINEMATIC_VELOCITY_KEPT
        if (!(w > 0.0f)) continue;
#else
        if (!(w > 0.0f)) { for (int k = 0; k < 3; ++k) s->vel[3 * i + k] = 0.0f; continue; }
#endif
#ifdef ORC_ALT_MAXACCEL_PER_FRAME
        const float *vref = s->vframe + 3 * i;
#else
        const float *vref = v0 + 3 * i;
#endif
        float v[3], dv[3], xe[3];
        for (int k = 0; k < 3; ++k) { xe[k] = xp[4 * i + k]; v[k] = (xe[k] - x0[4 * i + k]) * inv_h; dv[k] = v[k] - vref[k]; }
        float dv2 = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2];
        if (clamp_now && dv2 > maxdv * maxdv) {
            float sc_ = maxdv / sqrtf(dv2);
            for (int k = 0; k < 3; ++k) v[k] = vref[k] + dv[k] * sc_;
            s->accel_clamps++; /* (white box only) */
#ifdef ORC_ALT_MAXACCEL_POSITION
            for (int k = 0; k < 3; ++k) xe[k] = x0[4 * i + k] + h * v[k];
#endif
        }
        float v2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
        if (p->maxSpeed < FLT_MAX && v2 > p->maxSpeed * p->maxSpeed) {
            float sc_ = p->maxSpeed / sqrtf(v2);
            for (int k = 0; k < 3; ++k) v[k] = v[k] * sc_;
            v2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
#ifdef ORC_ALT_MAXACCEL_POSITION
            for (int k = 0; k < 3; ++k) xe[k] = x0[4 * i + k] + h * v[k];
#endif
        }
#if defined(ORC_ALT_NO_SLEEP) || defined(ORC_ALT_SLEEP_AT_PREDICT)
        (void)thr2;
        for (int k = 0; k < 3; ++k) { s->vel[3 * i + k] = v[k]; s->pos[4 * i + k] = xe[k]; }
#elif defined(ORC_ALT_SLEEP_VELOCITY_ONLY)
        for (int k = 0; k < 3; ++k) { s->vel[3 * i + k] = (v2 < thr2) ? 0.0f : v[k]; s->pos[4 * i + k] = xe[k]; }
#else
        if (v2 < thr2) { /* asleep: "considered fixed" -> keeps its position, zero velocity [I] */
            for (int k = 0; k < 3; ++k) s->vel[3 * i + k] = 0.0f;
        } else {
            for (int k = 0; k < 3; ++k) { s->vel[3 * i + k] = v[k]; s->pos[4 * i + k] = xe[k]; }
        }
#endif
    }
}

int orc_step(orc_sim *s, int n_steps) {
    if (!s || s->n <= 0) return -1;
    if (ORC_NEEDS_TABLE && !g_rsq_delta) {
        fprintf(stderr, "flex_oracle: orc_set_rsqrt_table() was not called (oracle/v_rsq_f32_gfx950.npz)\n");
        return -2;
    }
    for (int f = 0; f < n_steps; ++f) {
        const float h = s->p.dt / (float)s->p.numSubsteps;
        const float inv_h = 1.0f / h;
        for (int sub = 0; sub < s->p.numSubsteps; ++sub) substep(s, sub, h, inv_h);
    }
    return 0;
}

/* ---------------------------------------------------------------- accessors (pyflex.cpp:326-922) */

int orc_n_particles(const orc_sim *s) { return s->n; }
int orc_n_springs(const orc_sim *s) { return s->m; }
int orc_n_triangles(const orc_sim *s) { return s->t; }
int orc_n_shapes(const orc_sim *s) { return s->ns; }

int orc_get_positions(const orc_sim *s, float *o) { memcpy(o, s->pos, sizeof(float) * 4 * s->n); return 0; }
int orc_set_positions(orc_sim *s, const float *in) { memcpy(s->pos, in, sizeof(float) * 4 * s->n); return 0; }
int orc_get_velocities(const orc_sim *s, float *o) { memcpy(o, s->vel, sizeof(float) * 3 * s->n); return 0; }
int orc_set_velocities(orc_sim *s, const float *in) { memcpy(s->vel, in, sizeof(float) * 3 * s->n); return 0; }
int orc_get_phases(const orc_sim *s, int *o) { memcpy(o, s->phase, sizeof(int) * s->n); return 0; }
int orc_set_phases(orc_sim *s, const int *in) { memcpy(s->phase, in, sizeof(int) * s->n); return 0; }
int orc_get_rest_positions(const orc_sim *s, float *o) { memcpy(o, s->rest, sizeof(float) * 4 * s->n); return 0; }
int orc_get_normals(orc_sim *s, float *o) {
    compute_normals(s->pos, s->tris, s->n, s->t, s->nrm);
    memcpy(o, s->nrm, sizeof(float) * 4 * s->n);
    return 0;
}
int orc_get_edges(const orc_sim *s, int *o) { memcpy(o, s->sidx, sizeof(int) * 2 * s->m); return 0; }
int orc_get_faces(const orc_sim *s, int *o) { memcpy(o, s->tris, sizeof(int) * 3 * s->t); return 0; }
int orc_get_spring_lengths(const orc_sim *s, float *o) { memcpy(o, s->slen, sizeof(float) * s->m); return 0; }
int orc_get_spring_stiffness(const orc_sim *s, float *o) { memcpy(o, s->sk, sizeof(float) * s->m); return 0; }
int orc_get_scene_bounds(const orc_sim *s, float *lo, float *up) {
    memcpy(lo, s->scene_lower, sizeof(float) * 3);
    memcpy(up, s->scene_upper, sizeof(float) * 3);
    return 0;
}

int orc_get_params(const orc_sim *s, float *o) {
    const orc_params *p = &s->p;
    memset(o, 0, sizeof(float) * 32);
    o[0] = (float)p->numIterations; o[1] = (float)p->numSubsteps; o[2] = p->dt;
    o[3] = p->gravity[0]; o[4] = p->gravity[1]; o[5] = p->gravity[2];
    o[6] = p->radius; o[7] = p->solidRestDistance; o[8] = p->collisionDistance; o[9] = p->shapeCollisionMargin;
    o[10] = p->particleCollisionMargin; o[11] = p->dynamicFriction; o[12] = p->staticFriction; o[13] = p->particleFriction;
    o[14] = p->damping; o[15] = p->sleepThreshold; o[16] = p->relaxationFactor; o[17] = p->maxAcceleration;
    o[18] = p->maxSpeed; o[19] = p->restitution; o[20] = p->adhesion; o[21] = p->dissipation;
    o[22] = (float)p->numPlanes; o[23] = p->planes[0][0]; o[24] = p->planes[0][1]; o[25] = p->planes[0][2];
    o[26] = p->planes[0][3]; o[27] = (float)p->maxNeighbors; o[28] = (float)p->maxContacts; o[29] = (float)p->relaxationMode;
    return 0;
}

/* helpers.h:484-499 AddSphere: prev := current */
int orc_add_sphere(orc_sim *s, float radius, const float *pos, const float *quat) {
    if (s->ns >= ORC_MAX_SHAPES) return -1;
    int q = s->ns++;
    s->sh_radius[q] = radius;
    for (int k = 0; k < 3; ++k) { s->sh_pos[q][k] = pos[k]; s->sh_prev[q][k] = pos[k]; }
    for (int k = 0; k < 4; ++k) { s->sh_rot[q][k] = quat[k]; s->sh_prevrot[q][k] = quat[k]; }
    return 0;
}
int orc_clear_shapes(orc_sim *s) { s->ns = 0; return 0; }
/* pyflex.cpp:789-822 layout: pos3, prevPos3, quat4, prevQuat4 */
int orc_get_shape_states(const orc_sim *s, float *o) {
    for (int q = 0; q < s->ns; ++q) {
        for (int k = 0; k < 3; ++k) { o[14 * q + k] = s->sh_pos[q][k]; o[14 * q + 3 + k] = s->sh_prev[q][k]; }
        for (int k = 0; k < 4; ++k) { o[14 * q + 6 + k] = s->sh_rot[q][k]; o[14 * q + 10 + k] = s->sh_prevrot[q][k]; }
    }
    return 0;
}
int orc_set_shape_states(orc_sim *s, const float *in) {
    for (int q = 0; q < s->ns; ++q) {
        for (int k = 0; k < 3; ++k) { s->sh_pos[q][k] = in[14 * q + k]; s->sh_prev[q][k] = in[14 * q + 3 + k]; }
        for (int k = 0; k < 4; ++k) { s->sh_rot[q][k] = in[14 * q + 6 + k]; s->sh_prevrot[q][k] = in[14 * q + 10 + k]; }
    }
    return 0;
}

int orc_max_neighbor_list(const orc_sim *s) { return s->max_list; }
int orc_get_last_shape_candidates(const orc_sim *s, unsigned *masks) { memcpy(masks, s->smask, sizeof(unsigned) * s->n); return 0; }
int orc_missed_shape_contacts(const orc_sim *s) { return s->missed_shape_contacts; }
long orc_accel_clamps(const orc_sim *s) { return s->accel_clamps; }
long orc_degenerate_normals(const orc_sim *s) { return s->degenerate_normals; }

int orc_get_last_neighbors(const orc_sim *s, int *counts, int *lists) {
    memcpy(counts, s->ncount, sizeof(int) * s->n);
    memcpy(lists, s->nlist, sizeof(int) * (size_t)ORC_MAX_NEIGHBORS * s->n);
    return 0;
}
