// ref_sphere_probe.cpp -- TEST INFRASTRUCTURE.  Builds against the REFERENCE's own PyFlex/core/mesh.cpp + maths.cpp +
// platform.cpp (compiled where they lie under /root/reference, never copied) and prints the mesh the reference draws
// for a kinematic sphere shape: CreateSphere(20, 20, radius) (core/mesh.cpp:858-902) moved by
// TranslationMatrix(prev position) * RotationMatrix(prev rotation) (bindings/main.cpp:1739-1751).
// Output, consumed by tests/golden/make_golden.py: "counts V N I", then I indices, V positions, V normals (%.9g).
// usage: sphere_ref radius px py pz qx qy qz qw
#include <cstdio>
#include <cstdlib>

#include "core/maths.h"
#include "core/mesh.h"

int main(int argc, char **argv) {
    if (argc < 9) return 1;
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = (float)atof(argv[i + 1]);
    Mesh *sphere = CreateSphere(20, 20, a[0]);
    Matrix44 xform = TranslationMatrix(Point3(a[1], a[2], a[3])) * RotationMatrix(Quat(a[4], a[5], a[6], a[7]));
    sphere->Transform(xform);
    const int nv = (int)sphere->m_positions.size(), nn = (int)sphere->m_normals.size(), ni = (int)sphere->m_indices.size();
    printf("counts %d %d %d\n", nv, nn, ni);
    printf("indices");
    for (int i = 0; i < ni; ++i) printf(" %d", (int)sphere->m_indices[i]);
    printf("\npositions");
    for (int i = 0; i < nv; ++i) printf(" %.9g %.9g %.9g", sphere->m_positions[i].x, sphere->m_positions[i].y, sphere->m_positions[i].z);
    printf("\nnormals");
    for (int i = 0; i < nn; ++i) printf(" %.9g %.9g %.9g", sphere->m_normals[i].x, sphere->m_normals[i].y, sphere->m_normals[i].z);
    printf("\n");
    delete sphere;
    return 0;
}
