/*
 * flex_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the FlingBot cloth hot path:
 *   - scene build:  PyFlex/bindings/softgym_scenes/softgym_cloth.h:33-175,
 *                   PyFlex/bindings/helpers.h:144-150 (CreateSpring), :838-924 (CreateSpringGrid),
 *                   PyFlex/bindings/main.cpp:613-1122 (Init: defaults, derived params, normals, rest pose)
 *   - solver step:  NvFlexUpdateSolver (PyFlex/include/NvFlex.h:476-481) -- CLOSED SOURCE, library absent from
 *                   the checkout (.MISSING_LARGE_BLOBS:5).  The step below follows the parameter semantics of
 *                   NvFlex.h:86-154, the stage order of NvFlex.h:197-223 and the published algorithm of
 *                   Macklin et al., "Unified Particle Physics for Real-Time Applications" (SIGGRAPH 2014):
 *                   PARITY vs real PyFleX positions is UNPINNED (no runnable reference, no golden vectors exist).
 *   - mirrors:      PyFlex/bindings/pyflex.cpp:311-922 accessor semantics.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 */
#ifndef FLEX_ORACLE_H
#define FLEX_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_sim orc_sim;

orc_sim *orc_create(void);
void orc_destroy(orc_sim *s);

/* pyflex.set_scene (pyflex.cpp:229-244 -> main.cpp Init).  Array lengths are element counts (floats / ints). */
int orc_set_scene(orc_sim *s, const float *scene_params /*[19]*/, const float *verts, int n_vert_floats,
                  const int *stretch, int n_stretch_ints, const int *bend, int n_bend_ints, const int *shear,
                  int n_shear_ints, const int *faces, int n_face_ints);

/* The reciprocal square root of the constraint sweeps is the product's: gfx950's v_rsq_f32 on max(x, FLT_MIN), carried as a
   table of the chip's results (oracle/v_rsq_f32_gfx950.npz `delta2bit`, 4 MiB: see flex_oracle.c).  Process-global; must be
   set before orc_step (which returns -2 without it).  orc_eval_rsqrt: y[i] = that function of x[i]. */
void orc_set_rsqrt_table(const unsigned char *packed_2bit);
int orc_eval_rsqrt(const float *x, float *y, int n);

/* pyflex.step xN (pyflex.cpp:213-222 -> main.cpp UpdateFrame:2120) */
int orc_step(orc_sim *s, int n_steps);

int orc_n_particles(const orc_sim *s);
int orc_n_springs(const orc_sim *s);
int orc_n_triangles(const orc_sim *s);
int orc_n_shapes(const orc_sim *s);

int orc_get_positions(const orc_sim *s, float *out4n);
int orc_set_positions(orc_sim *s, const float *in4n);
int orc_get_velocities(const orc_sim *s, float *out3n);
int orc_set_velocities(orc_sim *s, const float *in3n);
int orc_get_phases(const orc_sim *s, int *outn);
int orc_set_phases(orc_sim *s, const int *inn);
int orc_get_rest_positions(const orc_sim *s, float *out4n);
int orc_get_normals(orc_sim *s, float *out4n); /* recomputed from current positions */
int orc_get_edges(const orc_sim *s, int *out2m);
int orc_get_faces(const orc_sim *s, int *out3t);
int orc_get_spring_lengths(const orc_sim *s, float *outm);
int orc_get_spring_stiffness(const orc_sim *s, float *outm);
int orc_get_params(const orc_sim *s, float *out32); /* packed effective parameter table, see .c */
int orc_get_scene_bounds(const orc_sim *s, float *lower3, float *upper3);

int orc_add_sphere(orc_sim *s, float radius, const float *pos3, const float *quat4);
int orc_clear_shapes(orc_sim *s);
int orc_get_shape_states(const orc_sim *s, float *out14s);
int orc_set_shape_states(orc_sim *s, const float *in14s);

/* neighbour (particle-contact candidate) lists of the LAST substep run, for white-box tests:
   out_counts[n], out_lists[n*96] */
/* longest candidate list (before truncation to 96 it is capped there) any particle has had since set_scene */
int orc_max_neighbor_list(const orc_sim *s);

int orc_get_last_neighbors(const orc_sim *s, int *out_counts, int *out_lists);

/* collideShapes of the LAST substep run (white box): per particle the candidate mask, bit q = plane q, bit 8 + q = sphere q;
   and how often, since set_scene, an iteration found a plane / sphere violated that was NOT a candidate (0 = the candidate
   stage never changed a result) */
int orc_get_last_shape_candidates(const orc_sim *s, unsigned *out_masks);
int orc_missed_shape_contacts(const orc_sim *s);
/* white box (PARITY.md): how often the maxAcceleration clamp of finalize changed a velocity / a contact normal fell back to
   (0,1,0) for a coincident pair, in particle-substeps / contacts since set_scene */
long orc_accel_clamps(const orc_sim *s);
long orc_degenerate_normals(const orc_sim *s);

#ifdef __cplusplus
}
#endif
#endif
