// fs_scene.cpp -- see fs_scene.h.  Compiled with -ffp-contract=off: every fp32 value below must come out exactly as the
// reference's x86 build computes it (no FMA fusion), because spring rest lengths are part of the bit-exact topology.
#include "fs_scene.h"

#include <cfloat>
#include <unordered_map>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <utility>

namespace {

struct SceneAssembler {
    FsHostScene &s;
    explicit SceneAssembler(FsHostScene &scene) : s(scene) {}

    int add_particle(float x, float y, float z, float inv_mass, int phase) {
        int id = int(s.pos.size() / 4);
        s.pos.insert(s.pos.end(), {x, y, z, inv_mass});
        s.vel.insert(s.vel.end(), {0.0f, 0.0f, 0.0f});
        s.phase.push_back(phase);
        return id;
    }

    void add_triangle(int a, int b, int c, float nx, float ny, float nz) {
        s.tris.insert(s.tris.end(), {a, b, c});
        s.tri_normals.insert(s.tri_normals.end(), {nx, ny, nz});
    }

    // reference helpers.h:144-150; rest length = (1 + give) * Length(p_i - p_j), give = 0
    void add_spring(int i, int j, float stiffness) {
        const float *pi = &s.pos[4 * size_t(i)], *pj = &s.pos[4 * size_t(j)];
        float ex = pi[0] - pj[0], ey = pi[1] - pj[1], ez = pi[2] - pj[2];
        float sq = ex * ex + ey * ey + ez * ez;
        float length = (sq != 0.0f) ? sqrtf(sq) : 0.0f;
        s.springs.push_back(i);
        s.springs.push_back(j);
        s.spring_len.push_back((1.0f + 0.0f) * length);
        s.spring_k.push_back(stiffness);
    }

    // reference helpers.h:838-924 with dz == 1: particles row-major (index = y*dx + x), two triangles per quad,
    // then a row-major pass of [stretch x-1, bend x-2, shear (x+1,y-1), shear (x-1,y-1)] and a column-major pass of
    // [stretch y-1, bend y-2].
    void grid_cloth(const float lower[3], int dx, int dy, float spacing, int phase, float k_stretch, float k_bend,
                    float k_shear, float inv_mass) {
        const int base = int(s.pos.size() / 4);
        auto at = [&](int x, int y) { return base + y * dx + x; };
        for (int y = 0; y < dy; ++y) {
            for (int x = 0; x < dx; ++x) {
                add_particle(lower[0] + spacing * float(x), lower[1] + spacing * float(0), lower[2] + spacing * float(y),
                             inv_mass, phase);
                if (x > 0 && y > 0) {
                    add_triangle(at(x - 1, y - 1), at(x, y - 1), at(x, y), 0.0f, 1.0f, 0.0f);
                    add_triangle(at(x - 1, y - 1), at(x, y), at(x - 1, y), 0.0f, 1.0f, 0.0f);
                }
            }
        }
        for (int y = 0; y < dy; ++y) {
            for (int x = 0; x < dx; ++x) {
                if (x > 0) add_spring(at(x, y), at(x - 1, y), k_stretch);
                if (x > 1) add_spring(at(x, y), at(x - 2, y), k_bend);
                if (y > 0 && x < dx - 1) add_spring(at(x, y), at(x + 1, y - 1), k_shear);
                if (y > 0 && x > 0) add_spring(at(x, y), at(x - 1, y - 1), k_shear);
            }
        }
        for (int x = 0; x < dx; ++x) {
            for (int y = 0; y < dy; ++y) {
                if (y > 0) add_spring(at(x, y), at(x, y - 1), k_stretch);
                if (y > 1) add_spring(at(x, y), at(x, y - 2), k_bend);
            }
        }
    }

    // reference softgym_cloth.h:69-132
    void mesh_cloth(const float lower[3], const float *verts, int nv, const int *faces, int nf, const int *stretch,
                    int n_stretch, const int *bend, int n_bend, const int *shear, int n_shear, float inv_mass, int phase,
                    float k_stretch, float k_bend, float k_shear) {
        const int base = int(s.pos.size() / 4);
        for (int i = 0; i < nv; ++i)
            add_particle(verts[3 * i] + lower[0], verts[3 * i + 1] + lower[1], verts[3 * i + 2] + lower[2],
                         inv_mass + 0.0f, phase);
        for (int f = 0; f < nf; ++f) {
            int a = base + faces[3 * f], b = base + faces[3 * f + 1], c = base + faces[3 * f + 2];
            const float *p1 = &s.pos[4 * size_t(a)], *p2 = &s.pos[4 * size_t(b)], *p3 = &s.pos[4 * size_t(c)];
            float ux = p2[0] - p1[0], uy = p2[1] - p1[1], uz = p2[2] - p1[2];
            float vx = p3[0] - p1[0], vy = p3[1] - p1[1], vz = p3[2] - p1[2];
            float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
            float sq = nx * nx + ny * ny + nz * nz;
            float len = (sq != 0.0f) ? sqrtf(sq) : 0.0f;
            add_triangle(a, b, c, nx / len, ny / len, nz / len);
        }
        for (int e = 0; e < n_stretch; ++e) add_spring(base + stretch[2 * e], base + stretch[2 * e + 1], k_stretch);
        for (int e = 0; e < n_bend; ++e) add_spring(base + bend[2 * e], base + bend[2 * e + 1], k_bend);
        for (int e = 0; e < n_shear; ++e) add_spring(base + shear[2 * e], base + shear[2 * e + 1], k_shear);
    }
};

void default_params(FsParams &p) {  // reference main.cpp:717-828
    memset(&p, 0, sizeof(p));
    p.dt = 1.0f / 100.0f;
    p.gravity[1] = -9.8f;
    p.radius = 0.15f;
    p.numIterations = 3;
    p.numSubsteps = 20;
    p.maxSpeed = FLT_MAX;
    p.maxAcceleration = 100.0f;
    p.relaxationMode = 1;  // eNvFlexRelaxationLocal
    p.relaxationFactor = 1.0f;
    p.numPlanes = 1;
    p.maxNeighbors = FS_MAX_NEIGHBORS;
    p.maxContacts = 6;
}

void build_adjacency(FsHostScene &s) {
    const int n = s.n, m = s.m;
    s.adj_off.assign(size_t(n) + 1, 0);
    for (int e = 0; e < 2 * m; ++e) s.adj_off[size_t(s.springs[e]) + 1]++;
    int deg_max = 0;
    for (int i = 0; i < n; ++i) {
        if (s.adj_off[i + 1] > deg_max) deg_max = s.adj_off[i + 1];
        s.adj_off[i + 1] += s.adj_off[i];
    }
    s.max_deg = deg_max;
    s.adj_j.assign(size_t(2) * m, 0);
    s.adj_len.assign(size_t(2) * m, 0.0f);
    s.adj_k.assign(size_t(2) * m, 0.0f);
    std::vector<int> cursor(s.adj_off.begin(), s.adj_off.end() - 1);
    for (int e = 0; e < m; ++e) {  // ascending spring id per particle
        for (int side = 0; side < 2; ++side) {
            int i = s.springs[2 * e + side], j = s.springs[2 * e + 1 - side];
            int slot = cursor[i]++;
            s.adj_j[slot] = j;
            s.adj_len[slot] = s.spring_len[e];
            s.adj_k[slot] = s.spring_k[e];
        }
    }
    s.ell_j.assign(size_t(deg_max) * n, -1);
    s.ell_len.assign(size_t(deg_max) * n, 0.0f);
    s.ell_k.assign(size_t(deg_max) * n, 0.0f);
    for (int i = 0; i < n; ++i)
        for (int a = s.adj_off[i]; a < s.adj_off[i + 1]; ++a) {
            size_t slot = size_t(a - s.adj_off[i]) * n + i;
            s.ell_j[slot] = s.adj_j[a];
            s.ell_len[slot] = s.adj_len[a];
            s.ell_k[slot] = s.adj_k[a];
        }
}

void build_compact_adjacency(FsHostScene &s) {
    s.dict_size = 0;
    s.dict.assign(512, 0.0f);
    s.code_w.clear();
    s.nbr_w.clear();
    const int n = s.n;
    if (s.max_deg > 16 || n > 4096) return;  // offsets must fit 16 bits: 4095 * 16 = 65520
    std::vector<std::pair<uint32_t, uint32_t>> entries;  // bit patterns of (len, k)
    std::vector<uint8_t> codes(size_t(16) * n, 0);
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    for (int i = 0; i < n; ++i)
        for (int a = s.adj_off[i]; a < s.adj_off[i + 1]; ++a) {
            std::pair<uint32_t, uint32_t> key(bits(s.adj_len[a]), bits(s.adj_k[a]));
            size_t c = 0;
            for (; c < entries.size(); ++c)
                if (entries[c] == key) break;
            if (c == entries.size()) {
                if (entries.size() == 256) return;  // too many distinct springs: compact form unavailable
                entries.push_back(key);
                s.dict[2 * c] = s.adj_len[a];
                s.dict[2 * c + 1] = s.adj_k[a];
            }
            codes[size_t(a - s.adj_off[i]) * n + i] = uint8_t(c);
        }
    // 16-bit fields, two per word, already scaled to LDS byte offsets: neighbour id * 16 (float4 array) and
    // dictionary code * 8 (float2 array)
    s.code_w.assign(size_t(8) * n, 0u);
    s.nbr_w.assign(size_t(8) * n, 0u);
    for (int i = 0; i < n; ++i) {
        const int deg = s.adj_off[i + 1] - s.adj_off[i];
        for (int slot = 0; slot < 16; ++slot) {
            const uint32_t j = slot < deg ? uint32_t(s.adj_j[s.adj_off[i] + slot]) : uint32_t(i);
            const uint32_t c = slot < deg ? codes[size_t(slot) * n + i] : 0u;
            s.code_w[size_t(slot / 2) * n + i] |= ((c * 8u) & 0xffffu) << (16 * (slot % 2));
            s.nbr_w[size_t(slot / 2) * n + i] |= ((j * 16u) & 0xffffu) << (16 * (slot % 2));
        }
    }
    s.dict_size = int(entries.size());
}

void build_stream_codes(FsHostScene &s) {
    s.sdict_size = 0;
    s.sdict.assign(1024, 0.0f);
    s.scode.clear();
    const int n = s.n;
    if (s.max_deg > 16 || n <= 0) return;
    struct Key {
        int32_t d; uint32_t l, k;
        bool operator==(const Key &o) const { return d == o.d && l == o.l && k == o.k; }
    };
    struct KeyHash {
        size_t operator()(const Key &x) const {
            uint64_t h = (uint64_t)(uint32_t)x.d * 0x9E3779B97F4A7C15ull;
            h ^= (uint64_t)x.l * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
            h ^= (uint64_t)x.k * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
            return (size_t)h;
        }
    };
    std::vector<Key> entries;
    std::unordered_map<Key, size_t, KeyHash> index;  // entry -> dictionary code (codes are handed out in order of first use)
    s.scode.assign(size_t(4) * n, 0xffffffffu);
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    for (int i = 0; i < n; ++i)
        for (int a = s.adj_off[i]; a < s.adj_off[i + 1]; ++a) {
            const Key key{s.adj_j[a] - i, bits(s.adj_len[a]), bits(s.adj_k[a])};
            const auto hit = index.find(key);
            const size_t c = hit == index.end() ? entries.size() : hit->second;
            if (c == entries.size()) {
                if (entries.size() == 255) { s.scode.clear(); return; }  // too many distinct springs
                entries.push_back(key);
                index.emplace(key, c);
                memcpy(&s.sdict[4 * c], &key.d, 4);
                s.sdict[4 * c + 1] = s.adj_len[a];
                s.sdict[4 * c + 2] = s.adj_k[a];
            }
            const int slot = a - s.adj_off[i];
            uint32_t &w = s.scode[size_t(4) * i + slot / 4];
            w = (w & ~(0xffu << (8 * (slot % 4)))) | (uint32_t(c) << (8 * (slot % 4)));
        }
    s.sdict_size = int(entries.size());
}

void build_grid_pattern(FsHostScene &s, int dimx, int dimz) {
    s.gp_count = 0;
    s.gp_dimx = dimx; s.gp_dimz = dimz;
    const int n = s.n;
    if (dimx < 5 || dimz < 5 || (long long)dimx * dimz != n || s.max_deg > 16) return;
    // canonical list: the incident springs of an interior particle (two cells away from every border)
    const int ic = 2 * dimx + 2;
    const int cdeg = s.adj_off[ic + 1] - s.adj_off[ic];
    if (cdeg <= 0 || cdeg > 16) return;
    int cdx[16], cdz[16];
    for (int q = 0; q < cdeg; ++q) {
        const int j = s.adj_j[s.adj_off[ic] + q];
        cdz[q] = j / dimx - 2;
        cdx[q] = j % dimx - 2;
        if (cdx[q] < -2 || cdx[q] > 2 || cdz[q] < -2 || cdz[q] > 2) return;
    }
    // every particle: its list == the in-bounds canonical entries, in canonical order
    for (int i = 0; i < n; ++i) {
        const int ix = i % dimx, iz = i / dimx;
        int a = s.adj_off[i];
        for (int q = 0; q < cdeg; ++q) {
            const int jx = ix + cdx[q], jz = iz + cdz[q];
            if (jx < 0 || jx >= dimx || jz < 0 || jz >= dimz) continue;
            if (a >= s.adj_off[i + 1] || s.adj_j[a] != jz * dimx + jx) return;
            ++a;
        }
        if (a != s.adj_off[i + 1]) return;
    }
    for (int q = 0; q < cdeg; ++q) { s.gp_dx[q] = cdx[q]; s.gp_dz[q] = cdz[q]; }
    s.gp_count = cdeg;
}

// Rest lengths in canonical slot order for ANY grid cloth whose springs follow FS_G64_DX_LIST / FS_G64_DZ_LIST with one
// stiffness per slot (gp_L_ok), and on top of that the structure the grid-64 fused kernel needs (g64_ok): see fs_scene.h.
void build_grid64(FsHostScene &s) {
    s.g64_ok = 0;
    s.gp_L_ok = 0;
    s.gp_halvable = 0;
    s.g64_L.clear();
    s.gp_magic = 0;
    static const int cdx[FS_G64_SLOTS] = FS_G64_DX_LIST, cdz[FS_G64_SLOTS] = FS_G64_DZ_LIST;
    if (s.gp_count != FS_G64_SLOTS || s.gp_dimx < 5 || s.gp_dimz < 5 || s.gp_dimx > 4096) return;
    for (int q = 0; q < FS_G64_SLOTS; ++q)
        if (s.gp_dx[q] != cdx[q] || s.gp_dz[q] != cdz[q]) return;
    const int n = s.n, dimx = s.gp_dimx, dimz = s.gp_dimz;
    std::vector<float> L(size_t(FS_G64_SLOTS) * n, 0.0f);
    bool have_k[FS_G64_SLOTS] = {false};
    bool halvable = true;
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    for (int i = 0; i < n; ++i) {
        const int ix = i % dimx, iz = i / dimx;
        int a = s.adj_off[i];
        for (int q = 0; q < FS_G64_SLOTS; ++q) {
            const int jx = ix + cdx[q], jz = iz + cdz[q];
            if (jx < 0 || jx >= dimx || jz < 0 || jz >= dimz) continue;
            // build_grid_pattern has verified that adjacency entry `a` is exactly this neighbour
            const float k = s.adj_k[a];
            if (have_k[q] && bits(s.g64_k[q]) != bits(k)) return;       // one stiffness per slot
            if (!(k > 0.0f) || (k * 0.5f) * 2.0f != k) halvable = false;  // tethers / non-halvable stiffness
            s.g64_k[q] = k; have_k[q] = true;
            L[size_t(q) * n + i] = s.adj_len[a];
            ++a;
        }
    }
    for (int q = 0; q < FS_G64_SLOTS; ++q)
        if (!have_k[q]) return;
    // row = (i * magic) >> 32 for every i < n (checked): one v_mul_hi instead of an integer division per thread
    const uint64_t magic = ((uint64_t(1) << 32) + dimx - 1) / dimx;
    if (magic > 0xffffffffull) return;
    for (int i = 0; i < n; ++i)
        if (int((uint64_t(i) * magic) >> 32) != i / dimx) return;
    s.gp_magic = uint32_t(magic);
    s.g64_L.swap(L);
    s.gp_L_ok = 1;
    s.gp_halvable = halvable ? 1 : 0;
    // grid-64 form: 64 columns, at most 64 rows, positive halvable stiffness; x-direction slots: one rest length per
    // column (taken from row 2 by the kernel); z-direction slots: one per row
    if (dimx != 64 || dimz > 64 || !halvable) return;
    const std::vector<float> &T = s.g64_L;
    for (int i = 0; i < n; ++i) {
        const int ix = i % 64, iz = i / 64;
        for (int q = 0; q < FS_G64_SLOTS; ++q) {
            const int jx = ix + cdx[q], jz = iz + cdz[q];
            if (jx < 0 || jx >= 64 || jz < 0 || jz >= dimz) continue;
            if (cdz[q] == 0 && bits(T[size_t(q) * n + i]) != bits(T[size_t(q) * n + 2 * 64 + ix])) return;
            if (cdx[q] == 0 && bits(T[size_t(q) * n + i]) != bits(T[size_t(q) * n + iz * 64 + 2])) return;
        }
    }
    s.g64_ok = 1;
}

// Rest-near sets with exactly the device's fp32 test: e = rest_i - rest_j, e.x*e.x + e.y*e.y + e.z*e.z < r*r.
void build_restnear(FsHostScene &s) {
    const int n = s.n;
    s.restnear_ok = 0;
    s.restnear_w.assign(size_t(8) * n, 0xffffffffu);
    if (n > 65535) return;
    const float r = s.params.radius + s.params.particleCollisionMargin;
    const float r2 = r * r;
    if (!(r > 0.0f)) return;
    // uniform grid over the rest pose, cell = r
    std::vector<long long> key(n);
    std::vector<int> order(n);
    auto cell = [&](int i, int k) { return (long long)floorf(s.pos[4 * size_t(i) + k] / r); };
    for (int i = 0; i < n; ++i) {
        key[i] = ((cell(i, 0) & 0x1fffff) << 42) | ((cell(i, 1) & 0x1fffff) << 21) | (cell(i, 2) & 0x1fffff);
        order[i] = i;
    }
    std::sort(order.begin(), order.end(), [&](int a, int b) { return key[a] != key[b] ? key[a] < key[b] : a < b; });
    std::vector<long long> sorted_keys(n);
    for (int q = 0; q < n; ++q) sorted_keys[q] = key[order[q]];
    // runs of equal keys = cells; the 27 neighbour cells are looked up once per CELL (a hash map from key to run), not once
    // per particle by binary search: 12.7 -> ~3 ms for a 90 x 90 cloth, which evaluate.run_tasks pays per episode
    std::vector<int> run_begin, run_end;
    std::unordered_map<long long, int> run_of;
    for (int q = 0; q < n;) {
        int e = q;
        while (e < n && sorted_keys[e] == sorted_keys[q]) ++e;
        run_of.emplace(sorted_keys[q], (int)run_begin.size());
        run_begin.push_back(q);
        run_end.push_back(e);
        q = e;
    }
    std::vector<int> found;
    // a canonical grid cloth (index = row * dimx + column, gp_magic valid): does every particle's set equal its 8 grid
    // neighbours?  Then the kernels test two index differences instead of sixteen packed ids (restnear_ok = 2).
    bool stencil = s.gp_L_ok && s.gp_dimx > 0 && (long long)s.gp_dimx * s.gp_dimz == n;
    for (size_t c = 0; c < run_begin.size(); ++c) {
        const int first = order[run_begin[c]];
        int nb[27], t = 0;
        for (long long dx = -1; dx <= 1; ++dx)
            for (long long dy = -1; dy <= 1; ++dy)
                for (long long dz = -1; dz <= 1; ++dz) {
                    const long long k = (((cell(first, 0) + dx) & 0x1fffff) << 42) | (((cell(first, 1) + dy) & 0x1fffff) << 21) |
                                        ((cell(first, 2) + dz) & 0x1fffff);
                    const auto hit = run_of.find(k);
                    nb[t++] = hit == run_of.end() ? -1 : hit->second;
                }
        for (int qi = run_begin[c]; qi < run_end[c]; ++qi) {
            const int i = order[qi];
            found.clear();
            const float *ri = &s.pos[4 * size_t(i)];
            for (t = 0; t < 27; ++t) {  // same visiting order as before: dx, dy, dz ascending, ids ascending inside a cell
                if (nb[t] < 0) continue;
                for (int q = run_begin[nb[t]]; q < run_end[nb[t]]; ++q) {
                    const int j = order[q];
                    if (j == i) continue;
                    const float *rj = &s.pos[4 * size_t(j)];
                    const float ex = ri[0] - rj[0], ey = ri[1] - rj[1], ez = ri[2] - rj[2];
                    const float e2 = ex * ex + ey * ey + ez * ez;
                    if (e2 < r2) found.push_back(j);
                }
            }
            if (found.size() > 16) return;  // restnear_ok stays 0
            if (stencil) {  // is the set exactly the particle's in-grid 8-neighbourhood?
                const int dimx = s.gp_dimx, dimz = s.gp_dimz, ix = i % dimx, iz = i / dimx;
                size_t want = 0;
                for (int dz = -1; dz <= 1 && stencil; ++dz)
                    for (int dx = -1; dx <= 1; ++dx) {
                        if ((dx == 0 && dz == 0) || ix + dx < 0 || ix + dx >= dimx || iz + dz < 0 || iz + dz >= dimz) continue;
                        ++want;
                        if (std::find(found.begin(), found.end(), (iz + dz) * dimx + ix + dx) == found.end()) { stencil = false; break; }
                    }
                if (want != found.size()) stencil = false;
            }
            for (size_t q = 0; q < found.size(); ++q) {
                uint32_t &w = s.restnear_w[size_t(q / 2) * n + i];
                w = (w & ~(0xffffu << (16 * (q % 2)))) | (uint32_t(found[q]) << (16 * (q % 2)));
            }
        }
    }
    s.restnear_ok = stencil ? 2 : 1;
}

void build_vertex_triangles(FsHostScene &s) {
    s.vt_off.assign(size_t(s.n) + 1, 0);
    for (size_t c = 0; c < s.tris.size(); ++c) s.vt_off[size_t(s.tris[c]) + 1]++;
    for (int i = 0; i < s.n; ++i) s.vt_off[i + 1] += s.vt_off[i];
    s.vt_tri.assign(s.tris.size(), 0);
    std::vector<int> cursor(s.vt_off.begin(), s.vt_off.end() - 1);
    for (int t = 0; t < s.t; ++t)
        for (int c = 0; c < 3; ++c) s.vt_tri[cursor[s.tris[3 * size_t(t) + c]]++] = t;
}

}  // namespace

std::string fs_build_scene(FsHostScene &s, const float *sp, int n_params, const float *verts, int n_vert_floats,
                           const int *stretch, int n_stretch_ints, const int *bend, int n_bend_ints, const int *shear,
                           int n_shear_ints, const int *faces, int n_face_ints) {
    if (!sp || n_params < 19) return "scene_params needs 19 floats (flex_utils.py:332-342)";
    s = FsHostScene();
    FsParams &p = s.params;
    default_params(p);

    const float cloth_spacing = 0.00625f;  // softgym_cloth.h:47
    const int dimx = int(sp[3]), dimz = int(sp[4]);
    const float k_stretch = sp[5], k_bend = sp[6], k_shear = sp[7];
    s.render_mode = int(sp[8]);
    for (int k = 0; k < 3; ++k) { s.cam_pos[k] = sp[9 + k]; s.cam_angle[k] = sp[12 + k]; }
    s.cam_width = int(sp[15]);
    s.cam_height = int(sp[16]);
    const int flip_mesh = int(sp[18]);
    // NvFlexMakePhase(0, SelfCollide | SelfCollideFilter): group 0, all shape channels (NvFlex.h:187-192)
    const int phase = (0 & FS_PHASE_GROUP_MASK) |
                      ((FS_PHASE_SELF_COLLIDE | FS_PHASE_SELF_COLLIDE_FILTER) & 0x00f00000) | FS_PHASE_CHANNEL_MASK;
    const float lower[3] = {sp[0], -sp[1], sp[2]};  // y negated: softgym_cloth.h:76,136

    SceneAssembler asmb(s);
    const int nv = n_vert_floats / 3;
    if (nv > 0) {
        if (!verts) return "vertices pointer is null";
        const int nf = n_face_ints / 3;
        auto in_range = [&](const int *a, int cnt) {
            for (int i = 0; i < cnt; ++i)
                if (a[i] < 0 || a[i] >= nv) return false;
            return true;
        };
        if (!in_range(faces, nf * 3) || !in_range(stretch, n_stretch_ints / 2 * 2) ||
            !in_range(bend, n_bend_ints / 2 * 2) || !in_range(shear, n_shear_ints / 2 * 2))
            return "mesh index out of range";
        const float mass = sp[17] / float(nv);
        asmb.mesh_cloth(lower, verts, nv, faces, nf, stretch, n_stretch_ints / 2, bend, n_bend_ints / 2, shear,
                        n_shear_ints / 2, 1.0f / mass, phase, k_stretch, k_bend, k_shear);
    } else {
        if (dimx < 1 || dimz < 1 || (long long)dimx * dimz > (1 << 24)) return "bad cloth grid size";
        const float mass = sp[17] / float(dimx * dimz);
        asmb.grid_cloth(lower, dimx, dimz, cloth_spacing, phase, k_stretch, k_bend, k_shear, 1.0f / mass);
    }
    if (flip_mesh) {  // softgym_cloth.h:137-152 (folding task; triangle winding only)
        const int first = (dimx - 1) * 1 / 8;
        for (int j = 0; j < dimz - 1; ++j)
            for (int i = first; i < first + 5; ++i) {
                size_t q = size_t(j * (dimx - 1) + i) * 6;
                if (q + 5 >= s.tris.size()) continue;
                if (i != first + 4) std::swap(s.tris[q], s.tris[q + 1]);
                if (i != first) std::swap(s.tris[q + 3], s.tris[q + 4]);
            }
    }

    // softgym_cloth.h:154-170
    p.numSubsteps = 4;
    p.numIterations = 30;
    p.dynamicFriction = 0.75f;
    p.particleFriction = 1.0f;
    p.damping = 1.0f;
    p.sleepThreshold = 0.02f;
    p.relaxationFactor = 1.0f;
    p.shapeCollisionMargin = 0.04f;
    p.radius = cloth_spacing * 1.8f;
    p.collisionDistance = 0.005f;
    for (int k = 0; k < 3; ++k) { s.scene_lower[k] = -1.0f; s.scene_upper[k] = 1.0f; }

    // main.cpp:844-864 derived parameters
    if (p.solidRestDistance == 0.0f) p.solidRestDistance = p.radius;
    if (p.collisionDistance == 0.0f) p.collisionDistance = p.solidRestDistance * 0.5f;
    if (p.particleFriction == 0.0f) p.particleFriction = p.dynamicFriction * 0.1f;
    if (p.shapeCollisionMargin == 0.0f) p.shapeCollisionMargin = p.collisionDistance * 0.5f;

    s.n = int(s.pos.size() / 4);
    s.m = int(s.spring_len.size());
    s.t = int(s.tris.size() / 3);
    if (s.n == 0) return "scene has no particles";

    // main.cpp:866-879 scene bounds (no shapes exist at Init time: main.cpp:701-706)
    for (int i = 0; i < s.n; ++i)
        for (int k = 0; k < 3; ++k) {
            float v = s.pos[4 * size_t(i) + k];
            if (v < s.scene_lower[k]) s.scene_lower[k] = v;
            if (v > s.scene_upper[k]) s.scene_upper[k] = v;
        }
    for (int k = 0; k < 3; ++k) { s.scene_lower[k] -= p.collisionDistance; s.scene_upper[k] += p.collisionDistance; }
    // main.cpp:882-884 ground plane (g_waveFloorTilt = 0)
    p.planes[0][0] = 0.0f; p.planes[0][1] = 1.0f; p.planes[0][2] = 0.0f; p.planes[0][3] = 0.0f;

    build_adjacency(s);
    build_compact_adjacency(s);
    build_stream_codes(s);
    build_grid_pattern(s, nv > 0 ? 0 : dimx, nv > 0 ? 0 : dimz);
    build_grid64(s);
    build_restnear(s);
    build_vertex_triangles(s);
    return "";
}
