// ref_camera_probe.cpp -- builds against the REFERENCE's own PyFlex/core/maths.h (compiled where it lies under
// /root/reference, never copied) and prints the camera / light matrices RenderScene computes
// (PyFlex/bindings/main.cpp:1411-1438).  Output: one number per line group, consumed by tests/golden/make_golden.py.
// usage: camera_ref px py pz ax ay az width height lowx lowy lowz upx upy upz
#include <cstdio>
#include <cstdlib>

#include "core/maths.h"

static void dump(const char *name, const Matrix44 &m) {
    // Matrix44 is column-major (columns[c][r]); print row-major
    printf("%s", name);
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) printf(" %.9g", m.columns[c][r]);
    printf("\n");
}

int main(int argc, char **argv) {
    if (argc < 15) return 1;
    float a[14];
    for (int i = 0; i < 14; ++i) a[i] = (float)atof(argv[i + 1]);
    Vec3 g_camPos(a[0], a[1], a[2]), g_camAngle(a[3], a[4], a[5]);
    int g_screenWidth = (int)a[6], g_screenHeight = (int)a[7];
    Vec3 g_sceneLower(a[8], a[9], a[10]), g_sceneUpper(a[11], a[12], a[13]);
    float fov = kPi * 39.5978f / 180.0f;  // main.cpp:474
    float g_camNear = 0.01f, g_camFar = 3.0f, g_lightDistance = 10.0f;

    float aspect = float(g_screenWidth) / g_screenHeight;
    Matrix44 proj = ProjectionMatrix(RadToDeg(fov), aspect, g_camNear, g_camFar);
    Matrix44 view = RotationMatrix(-g_camAngle.x, Vec3(0.0f, 1.0f, 0.0f)) *
                    RotationMatrix(-g_camAngle.y, Vec3(cosf(-g_camAngle.x), 0.0f, sinf(-g_camAngle.x))) *
                    TranslationMatrix(-Point3(g_camPos));
    g_sceneLower = Min(g_sceneLower, Vec3(-2.0f, 0.0f, -2.0f));
    g_sceneUpper = Max(g_sceneUpper, Vec3(2.0f, 2.0f, 2.0f));
    Vec3 sceneExtents = g_sceneUpper - g_sceneLower;
    Vec3 sceneCenter = 0.5f * (g_sceneUpper + g_sceneLower);
    Vec3 g_lightDir = Normalize(Vec3(5.0f, 15.0f, 7.5f));
    Vec3 g_lightPos = sceneCenter + g_lightDir * Length(sceneExtents) * g_lightDistance;
    Vec3 g_lightTarget = sceneCenter;
    float lightFov = 2.0f * atanf(Length(g_sceneUpper - sceneCenter) / Length(g_lightPos - sceneCenter));
    lightFov = Clamp(lightFov, DegToRad(25.0f), DegToRad(65.0f));
    Matrix44 lightPerspective = ProjectionMatrix(RadToDeg(lightFov), 1.0f, 1.0f, 1000.0f);
    Matrix44 lightView = LookAtMatrix(Point3(g_lightPos), Point3(g_lightTarget));
    Matrix44 lightTransform = lightPerspective * lightView;
    Vec3 ld = Normalize(g_lightTarget - g_lightPos);
    dump("view", view);
    dump("proj", proj);
    dump("light", lightTransform);
    printf("lightpos %.9g %.9g %.9g\n", g_lightPos.x, g_lightPos.y, g_lightPos.z);
    printf("lightdir %.9g %.9g %.9g\n", ld.x, ld.y, ld.z);
    return 0;
}
