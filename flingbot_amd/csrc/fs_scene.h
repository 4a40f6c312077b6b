// fs_scene.h -- host-side cloth scene builder (topology + parameters) of libflingsim.
//
// Rebuilds, with bit-identical index order and fp32 values, what the reference assembles in
//   SoftgymCloth::Initialize  PyFlex/bindings/softgym_scenes/softgym_cloth.h:33-175
//   CreateSpringGrid          PyFlex/bindings/helpers.h:838-924
//   CreateSpring              PyFlex/bindings/helpers.h:144-150
//   Init (defaults, derived params, bounds, normals, rest pose)  PyFlex/bindings/main.cpp:613-1122
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "fs_types.h"

struct FsHostScene {
    int n = 0, m = 0, t = 0;
    std::vector<float> pos;      // 4n
    std::vector<float> vel;      // 3n
    std::vector<int> phase;      // n
    std::vector<int> springs;    // 2m
    std::vector<float> spring_len, spring_k;  // m
    std::vector<int> tris;       // 3t
    std::vector<float> tri_normals;  // 3t (initial)
    FsParams params;
    float scene_lower[3], scene_upper[3];
    // camera from scene_params (softgym_cloth.h:177-183 CenterCamera)
    float cam_pos[3], cam_angle[3];
    int cam_width, cam_height;
    int render_mode;
    // adjacency derived from `springs`
    int max_deg = 0;
    std::vector<int> adj_off, adj_j;
    std::vector<float> adj_len, adj_k;
    std::vector<int> ell_j;
    std::vector<float> ell_len, ell_k;
    // compact adjacency for the fused kernel: distinct (rest length, stiffness) pairs form a dictionary (<= 256
    // entries, else dict_size = 0); per particle 16 slots of two 16-bit fields (packed 2 per word, [8][n] each):
    // dictionary code * 8 and neighbour id * 16, i.e. ready-made LDS byte offsets.  Empty slots point at the
    // particle itself (zero length => no-op).
    int dict_size = 0;
    std::vector<float> dict;        // 2 * 256: (len, k) pairs
    std::vector<uint32_t> code_w;   // [8][n]
    std::vector<uint32_t> nbr_w;    // [8][n]
    // compact adjacency for the streaming kernels (any n): dictionary of distinct (neighbour offset j - i, rest length,
    // stiffness) triples -- a grid cloth has < 200 of them -- and ONE byte per spring slot: 16 bytes per particle instead
    // of 12 x (index, length, stiffness) = 144.  sdict[c] = (bits(j - i), len, k, 0); code 255 = empty slot and
    // sdict[255] = 0 (the slot then gathers the particle itself with zero length: a no-op).  sdict_size = 0 when
    // max_deg > 16 or there are more than 255 distinct triples (the kernels then stream the ELL arrays).
    int sdict_size = 0;
    std::vector<float> sdict;       // [256][4]
    std::vector<uint32_t> scode;    // [n][4], slot s in byte s of the particle's 16
    // grid pattern (streaming kernels): when the cloth is the dimx x dimz grid and every particle's incident springs, in
    // spring-id order, are exactly the in-bounds members of ONE canonical list of (dx, dz) offsets taken in that list's
    // order, a kernel can compute all neighbour ids of particle (ix, iz) arithmetically -- the spring gathers then need
    // no adjacency load in front of them.  gp_count = 0 when the pattern does not hold (meshes, tiny grids).
    int gp_count = 0, gp_dimx = 0, gp_dimz = 0;
    int gp_dx[16] = {0}, gp_dz[16] = {0};
    // grid-64 form (fused kernel): a dimx == 64 grid cloth (<= 64 rows) whose springs follow the canonical list
    // FS_G64_DX_LIST / FS_G64_DZ_LIST with a positive, exactly halvable stiffness per slot, whose x-direction rest
    // lengths (slots with dz == 0) depend on the column only and whose z-direction ones (dx == 0) on the row only --
    // what lower + spacing * (x, 0, z) in fp32 produces (helpers.h:852).  g64_L: rest lengths [12][n] in canonical slot
    // order; g64_k: stiffness per slot.
    // gp_L_ok: the table alone (any grid size, any sign of stiffness, one stiffness per slot) -- what the streaming
    // kernels' grid form reads instead of an adjacency; gp_magic = ceil(2^32 / dimx): row = (i * gp_magic) >> 32.
    int g64_ok = 0, gp_L_ok = 0, gp_halvable = 0;  // gp_halvable: every stiffness positive and exactly halvable
    uint32_t gp_magic = 0;
    std::vector<float> g64_L;
    float g64_k[FS_G64_SLOTS] = {0};
    // rest-pose neighbours for the SelfCollideFilter test (NvFlex.h:166,564-565): ids of the particles closer than the
    // interaction radius in the rest pose, 16 slots of 16 bits packed two per word, [8][n], 0xffff = empty.
    // restnear_ok = 0 when some particle has more than 16 of them or n > 65535 (the kernels then test rest positions);
    // 2 when the cloth is a canonical grid and every particle's set is exactly its in-grid 8-neighbourhood (always the case
    // for the reference's cloths: pitch 0.00625, radius 0.01125) -- the kernels then test index differences.
    int restnear_ok = 0;
    std::vector<uint32_t> restnear_w;
    // vertex -> incident triangles (ascending triangle id), for the vertex-normal gather
    std::vector<int> vt_off, vt_tri;
};

// Returns "" on success, otherwise an error message.
std::string fs_build_scene(FsHostScene &out, const float *scene_params, int n_params, const float *verts,
                           int n_vert_floats, const int *stretch, int n_stretch, const int *bend, int n_bend,
                           const int *shear, int n_shear, const int *faces, int n_faces);
