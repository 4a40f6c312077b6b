// pyflex_module.cpp -- pybind11 module named `pyflex` with the reference's Python surface
// (PyFlex/bindings/pyflex.cpp:1135-1208: the same 40 names, argument order and defaults), implemented purely on the
// C-ABI of include/flingsim.h.  Drop-in for environment/flex_utils.py, environment/simEnv.py, environment/tasks.py:
// put this module's directory on PYTHONPATH instead of PyFlex/bindings/build.
//
// Like the reference, the module is a process-global singleton = env 0 of a 1-episode context.  Unlike the reference,
// failures raise RuntimeError instead of printf/exit (a strict superset; pyflex.cpp:103-107).
// Batched / multi-episode use goes through flingbot_amd.sim (same library, additive API).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <cstdlib>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "flingsim.h"

namespace py = pybind11;
using farr = py::array_t<float, py::array::c_style | py::array::forcecast>;  // float64 inputs are cast like the reference
using iarr = py::array_t<int, py::array::c_style | py::array::forcecast>;

static fs_ctx *g_ctx = nullptr;
static float g_shape_color[3] = {0.9f, 0.9f, 0.9f};

static void fail(const char *what) { throw std::runtime_error(std::string(what) + ": " + fs_last_error()); }
static void ck(int rc, const char *what) {
    if (rc < 0) fail(what);
}
// How many processes share the device decides the back-end (csrc/fs_tenants.cpp): a lone cloth steps fastest on the streaming
// kernels (129 small launches per frame spread over the chip), but the chip dispatches such launches at one rate IN TOTAL, so
// sixteen processes doing that -- the reference's documented deployment, one PyFleX per Ray worker, `--num_processes 16`,
// README.md:147-148 -- share one process's rate; the fused kernel is ONE launch per frame on one compute unit, and sixteen of
// those do run side by side.  FS_SOLVER_COTENANT takes the fused kernel for every launch that fits it (<= 4096 particles) and
// AUTO's choice otherwise; results are identical either way (every back-end equals the oracle bit for bit).
//   FLINGSIM_SHARED_GPU unset  the module finds out by itself: pyflex.init registers the process in the device's co-tenant
//                              table, set_scene (and every 64th step) counts the live tenants; two or more => COTENANT
//   FLINGSIM_SHARED_GPU=1 / 0  the caller's word wins: always / never COTENANT (tenants in other containers or of other users
//                              are invisible to the table)
static int g_shared_env = -1;      // -1: unset (detect), 0 / 1: FLINGSIM_SHARED_GPU, read once in pyflex.init
static bool g_cotenant = false;    // the back-end in force
static char g_device_key[64] = "";
static unsigned g_steps_since_check = 0;
static int g_raw_tenants = -1;     // occupied slots of the table when the back-end was last chosen
static fs_ctx *ctx() {
    if (!g_ctx) throw std::runtime_error("pyflex.init() has not been called");
    return g_ctx;
}
static void choose_backend(bool prune) {
    bool shared = g_shared_env > 0;
    if (g_shared_env < 0 && g_device_key[0]) shared = fs_tenants_count(g_device_key, prune ? 1 : 0) >= 2;
    if (shared != g_cotenant || prune) ck(fs_set_solver(ctx(), shared ? FS_SOLVER_COTENANT : FS_SOLVER_AUTO), "pyflex");
    g_cotenant = shared;
    g_steps_since_check = 0;
    g_raw_tenants = g_device_key[0] ? fs_tenants_count(g_device_key, -1) : -1;
}

// pyflex.cpp:15-124.  m.def has no py::arg there either: four required positionals.
static void pyflex_init(bool headless, bool render, int camera_width, int camera_height) {
    (void)headless;
    (void)render;
    if (g_ctx) return;  // the reference initialises once per process
    int device = 0;
    if (const char *s = std::getenv("FLINGSIM_DEVICE")) device = std::atoi(s);
    g_ctx = fs_create(device, 1, camera_width, camera_height);
    if (!g_ctx) fail("pyflex.init");
    const char *shared = std::getenv("FLINGSIM_SHARED_GPU");
    g_shared_env = (shared && *shared) ? (std::atoi(shared) != 0 ? 1 : 0) : -1;
    // always registered, whatever the switch says about THIS process: the others count us
    if (fs_device_key(g_ctx, g_device_key, (int)sizeof(g_device_key)) != FS_OK || fs_tenants_register(g_device_key) < 0)
        g_device_key[0] = 0;   // no table (no writable /dev/shm or /tmp): behave as a lone tenant unless the switch says otherwise
    choose_backend(true);
}

static void pyflex_clean() {
    if (g_ctx) {
        if (g_device_key[0]) fs_tenants_unregister(g_device_key);
        fs_destroy(g_ctx);
    }
    g_ctx = nullptr;
}

// pyflex.cpp:229-244; scene_idx indexes g_scenes which only holds SoftgymCloth => only 0 is valid; thread_idx unused
static void pyflex_set_scene(int scene_idx, farr scene_params, farr vertices, iarr stretch_edges, iarr bend_edges,
                             iarr shear_edges, iarr faces, int thread_idx) {
    (void)thread_idx;
    if (scene_idx != 0) throw std::runtime_error("pyflex.set_scene: only scene_idx 0 (SoftgymCloth) exists");
    ck(fs_set_scene(ctx(), 0, scene_params.data(), (int)scene_params.size(), vertices.data(), (int)vertices.size(),
                    stretch_edges.data(), (int)stretch_edges.size(), bend_edges.data(), (int)bend_edges.size(),
                    shear_edges.data(), (int)shear_edges.size(), faces.data(), (int)faces.size()),
       "pyflex.set_scene");
    // every episode reset re-reads the co-tenant table (and clears the entries of workers that died): the back-end follows
    // the deployment without the caller knowing about it
    choose_backend(true);
}

// pyflex.cpp:213-222: update_params / capture / path are ignored by the cloth scene; render only toggles drawing
static void pyflex_step(py::object update_params, int capture, py::object path, int render) {
    (void)update_params; (void)capture; (void)path; (void)render;
    // workers start together: one that reached its first set_scene before its neighbours had called pyflex.init would step the
    // whole episode as a lone tenant -- so every step reads the number of occupied slots (no system call) and the live count is
    // taken again when that number has moved, and every 64 steps anyway (a killed neighbour leaves its slot occupied)
    if (g_shared_env < 0 && g_device_key[0] &&
        (++g_steps_since_check >= 64 || fs_tenants_count(g_device_key, -1) != g_raw_tenants))
        choose_backend(false);
    ck(fs_step(ctx(), 0, 1), "pyflex.step");
}

static std::tuple<py::array_t<unsigned char>, py::array_t<float>> pyflex_render() {
    float cam[8];
    ck(fs_get_camera_params(ctx(), 0, cam), "pyflex.render");
    const int px = (int)cam[0] * (int)cam[1];
    py::array_t<unsigned char> img(px * 4);
    py::array_t<float> depth(px);
    ck(fs_render(ctx(), 0, img.mutable_data(), px * 4, depth.mutable_data(), px), "pyflex.render");
    return std::make_tuple(img, depth);
}

static py::array_t<float> pyflex_get_camera_params() {
    py::array_t<float> out(8);
    ck(fs_get_camera_params(ctx(), 0, out.mutable_data()), "pyflex.get_camera_params");
    return out;
}
static void pyflex_set_camera_params(farr p) {
    if (p.size() < 8) throw std::runtime_error("pyflex.set_camera_params needs 8 floats");
    ck(fs_set_camera_params(ctx(), 0, p.data()), "pyflex.set_camera_params");
}

static int n_particles() { int n = fs_n_particles(ctx(), 0); ck(n, "pyflex"); return n; }
static int n_shapes() { int n = fs_n_shapes(ctx(), 0); ck(n, "pyflex"); return n; }

template <typename F> static py::array_t<float> getf(F fn, int count, const char *what) {
    py::array_t<float> out(count);
    ck(fn(ctx(), 0, out.mutable_data(), count), what);
    return out;
}
template <typename F> static py::array_t<int> geti(F fn, int count, const char *what) {
    py::array_t<int> out(count);
    ck(fn(ctx(), 0, out.mutable_data(), count), what);
    return out;
}
// the reference trusts the caller's length (loops run to the internal count, pyflex.cpp:471); we check instead
static void need(py::ssize_t have, int want, const char *what) {
    if (have < want) throw std::runtime_error(std::string(what) + ": array too short");
}

static py::array_t<float> pyflex_get_positions() { return getf(fs_get_positions, 4 * n_particles(), "pyflex.get_positions"); }
static void pyflex_set_positions(farr a) {
    need(a.size(), 4 * n_particles(), "pyflex.set_positions");
    ck(fs_set_positions(ctx(), 0, a.data(), (int)a.size()), "pyflex.set_positions");
}
static py::array_t<float> pyflex_get_velocities() { return getf(fs_get_velocities, 3 * n_particles(), "pyflex.get_velocities"); }
static void pyflex_set_velocities(farr a) {
    need(a.size(), 3 * n_particles(), "pyflex.set_velocities");
    ck(fs_set_velocities(ctx(), 0, a.data(), (int)a.size()), "pyflex.set_velocities");
}
static py::array_t<int> pyflex_get_phases() { return geti(fs_get_phases, n_particles(), "pyflex.get_phases"); }
static void pyflex_set_phases(iarr a) {
    need(a.size(), n_particles(), "pyflex.set_phases");
    ck(fs_set_phases(ctx(), 0, a.data(), (int)a.size()), "pyflex.set_phases");
}
// groups = low 20 bits of the phase (pyflex.cpp:343-376)
static py::array_t<int> pyflex_get_groups() {
    py::array_t<int> ph = pyflex_get_phases();
    int *p = ph.mutable_data();
    for (py::ssize_t i = 0; i < ph.size(); ++i) p[i] &= 0xfffff;
    return ph;
}
static void pyflex_set_groups(iarr groups) {
    need(groups.size(), n_particles(), "pyflex.set_groups");
    py::array_t<int> ph = pyflex_get_phases();
    int *p = ph.mutable_data();
    const int *g = groups.data();
    for (py::ssize_t i = 0; i < ph.size(); ++i) p[i] = (p[i] & ~0xfffff) | (g[i] & 0xfffff);
    ck(fs_set_phases(ctx(), 0, p, (int)ph.size()), "pyflex.set_groups");
}
static py::array_t<float> pyflex_get_restPositions() {
    return getf(fs_get_rest_positions, 4 * n_particles(), "pyflex.get_restPositions");
}
static py::array_t<int> pyflex_get_edges() {
    int m = fs_n_springs(ctx(), 0); ck(m, "pyflex.get_edges");
    return geti(fs_get_edges, 2 * m, "pyflex.get_edges");
}
static py::array_t<int> pyflex_get_faces() {
    int t = fs_n_triangles(ctx(), 0); ck(t, "pyflex.get_faces");
    return geti(fs_get_faces, 3 * t, "pyflex.get_faces");
}

static py::array_t<float> pyflex_get_shape_states() { return getf(fs_get_shape_states, 14 * n_shapes(), "pyflex.get_shape_states"); }
static void pyflex_set_shape_states(farr a) {
    need(a.size(), 14 * n_shapes(), "pyflex.set_shape_states");
    ck(fs_set_shape_states(ctx(), 0, a.data(), (int)a.size()), "pyflex.set_shape_states");
}
static void pyflex_add_sphere(float radius, farr pos, farr quat) {
    need(pos.size(), 3, "pyflex.add_sphere"); need(quat.size(), 4, "pyflex.add_sphere");
    ck(fs_add_sphere(ctx(), 0, radius, pos.data(), quat.data()), "pyflex.add_sphere");
}
static void pyflex_clear_shapes() { ck(fs_clear_shapes(ctx(), 0), "pyflex.clear_shapes"); }
static void pyflex_set_shape_color(farr c) {
    need(c.size(), 3, "pyflex.set_shape_color");
    for (int k = 0; k < 3; ++k) g_shape_color[k] = c.data()[k];
}
// Boxes / capsules / rigid bodies are never created by FlingBot's cloth path ("exported for API
// completeness"): adding one is refused loudly rather than silently simulated wrong.
static void pyflex_add_box(py::object, py::object, py::object, int) {
    throw std::runtime_error("pyflex.add_box: box shapes are not part of the cloth hot path (not implemented)");
}
static void pyflex_add_capsule(py::object, py::object, py::object) {
    throw std::runtime_error("pyflex.add_capsule: capsule shapes are not part of the cloth hot path (not implemented)");
}
static void pyflex_pop_box(int) { throw std::runtime_error("pyflex.pop_box: no boxes exist"); }
static void pyflex_add_rigid_body(py::object, py::object, int, py::object) {
    throw std::runtime_error("pyflex.add_rigid_body: rigid bodies are not part of the cloth hot path (not implemented)");
}

static py::array_t<float> bounds(bool upper) {
    float lo[3], up[3];
    ck(fs_get_scene_bounds(ctx(), 0, lo, up), "pyflex.get_scene_bounds");
    py::array_t<float> out(3);
    for (int k = 0; k < 3; ++k) out.mutable_data()[k] = upper ? up[k] : lo[k];
    return out;
}
static py::array_t<float> empty_f() { return py::array_t<float>(0); }
static py::array_t<int> empty_i() { return py::array_t<int>(0); }

PYBIND11_MODULE(pyflex, m) {
    m.doc() = "MI355X-native drop-in for the PyFleX `pyflex` module (cloth scene), backed by libflingsim";
    m.def("main", []() {});
    m.def("init", &pyflex_init);
    m.def("set_scene", &pyflex_set_scene, py::arg("scene_idx") = 0, py::arg("scene_params") = farr(),
          py::arg("vertices") = farr(), py::arg("stretch_edges") = iarr(), py::arg("bend_edges") = iarr(),
          py::arg("shear_edges") = iarr(), py::arg("faces") = iarr(), py::arg("thread_idx") = 0);
    m.def("clean", &pyflex_clean);
    m.def("step", &pyflex_step, py::arg("update_params") = py::none(), py::arg("capture") = 0,
          py::arg("path") = py::none(), py::arg("render") = 0);
    m.def("render", &pyflex_render);
    m.def("get_camera_params", &pyflex_get_camera_params, "Get camera parameters");
    m.def("set_camera_params", &pyflex_set_camera_params, "Set camera parameters");
    m.def("add_box", &pyflex_add_box, py::arg("halfEdge_") = 0, py::arg("center_") = 0, py::arg("quat_") = 0,
          py::arg("trigger") = 0, "Add box to the scene");
    m.def("add_sphere", &pyflex_add_sphere, "Add sphere to the scene");
    m.def("add_capsule", &pyflex_add_capsule, "Add capsule to the scene");
    m.def("pop_box", &pyflex_pop_box, "remove box from the scene");
    m.def("get_n_particles", &n_particles, "Get the number of particles");
    m.def("get_n_shapes", &n_shapes, "Get the number of shapes");
    m.def("get_n_rigids", []() { return 0; }, "Get the number of rigids");
    m.def("get_n_rigidPositions", []() { return 0; }, "Get the number of rigid positions");
    m.def("get_phases", &pyflex_get_phases, "Get particle phases");
    m.def("set_phases", &pyflex_set_phases, "Set particle phases");
    m.def("get_groups", &pyflex_get_groups, "Get particle groups");
    m.def("set_groups", &pyflex_set_groups, "Set particle groups");
    m.def("get_positions", &pyflex_get_positions, "Get particle positions");
    m.def("set_positions", &pyflex_set_positions, "Set particle positions");
    m.def("get_edges", &pyflex_get_edges, "Get mesh edges");
    m.def("get_faces", &pyflex_get_faces, "Get mesh faces");
    m.def("get_restPositions", &pyflex_get_restPositions, "Get particle restPositions");
    m.def("get_rigidOffsets", &empty_i, "Get rigid offsets");
    m.def("get_rigidIndices", &empty_i, "Get rigid indices");
    m.def("get_rigidLocalPositions", &empty_f, "Get rigid local positions");
    m.def("get_rigidGlobalPositions", &empty_f, "Get rigid global positions");
    m.def("get_rigidRotations", &empty_f, "Get rigid rotations");
    m.def("get_rigidTranslations", &empty_f, "Get rigid translations");
    m.def("get_velocities", &pyflex_get_velocities, "Get particle velocities");
    m.def("set_velocities", &pyflex_set_velocities, "Set particle velocities");
    m.def("get_shape_states", &pyflex_get_shape_states, "Get shape states");
    m.def("set_shape_states", &pyflex_set_shape_states, "Set shape states");
    m.def("clear_shapes", &pyflex_clear_shapes, "Clear shapes");
    m.def("get_scene_upper", []() { return bounds(true); });
    m.def("get_scene_lower", []() { return bounds(false); });
    m.def("add_rigid_body", &pyflex_add_rigid_body);
    m.def("set_shape_color", &pyflex_set_shape_color, "Set the color of the shape");

    // ---- additive names: the reference's host loops around step(), run on the device ----------
    m.def("picker_reset", [](double picker_threshold, double particle_radius, double picker_radius) {
        if (fs_picker_reset(ctx(), 0, picker_threshold, particle_radius) != FS_OK) fail("pyflex.picker_reset");
        if (picker_radius >= 0.0 && fs_picker_set_radius(ctx(), 0, picker_radius) != FS_OK) fail("pyflex.picker_reset");
    }, py::arg("picker_threshold") = 0.005, py::arg("particle_radius") = 0.00625, py::arg("picker_radius") = -1.0,
          "Picker.reset bookkeeping (flex_utils.py:85,99-101) for the spheres added with add_sphere");
    m.def("movep", [](py::array targets, py::array_t<int, py::array::c_style | py::array::forcecast> grasp, double speed,
                      int limit, py::object min_steps, double eps) {
        int iters = 0, rc;
        const int ms = min_steps.is_none() ? -1 : min_steps.cast<int>();
        if (targets.dtype().is(py::dtype::of<float>())) {  // float32 targets: movep's arithmetic stays in float32
            auto t = py::array_t<float, py::array::c_style | py::array::forcecast>::ensure(targets);
            rc = fs_movep_batch_f32(ctx(), 1, (const int[]){0}, t.data(), grasp.data(), speed, limit, ms, eps, &iters);
        } else {
            auto t = py::array_t<double, py::array::c_style | py::array::forcecast>::ensure(targets);
            rc = fs_movep(ctx(), 0, t.data(), grasp.data(), speed, limit, ms, eps, &iters);
        }
        if (rc == FS_ERR_LIMIT) throw std::runtime_error("MoveJointsException: movep limit reached");
        if (rc != FS_OK) fail("pyflex.movep");
        return iters;
    }, py::arg("targets"), py::arg("grasp"), py::arg("speed") = 0.1, py::arg("limit") = 1000,
          py::arg("min_steps") = py::none(), py::arg("eps") = 1e-4,
          "SimEnv.movep (simEnv.py:739-769): every simulation step on the device; returns the loop iterations");
    // the name SURVEY.md 8(f) f1 gives the same entry point: pyflex.step_n(targets, speed, grasp, max_steps)
    m.def("step_n", [](py::array targets, double speed, py::array_t<int, py::array::c_style | py::array::forcecast> grasp,
                       int max_steps) {
        return py::module_::import("pyflex").attr("movep")(targets, grasp, speed, max_steps);
    }, py::arg("targets"), py::arg("speed") = 0.1, py::arg("grasp") = py::array_t<int>(), py::arg("max_steps") = 1000,
          "alias of movep with the argument order of SURVEY.md 8(f): both pickers towards `targets` by `speed` per simulation "
          "step with grasp flags `grasp`, at most `max_steps` loop iterations, every step on the device");
    m.def("wait_until_stable", [](int max_steps, double tolerance) {
        int env = 0, steps = 0, stable = 0;
        if (fs_wait_until_stable(ctx(), 1, &env, max_steps, tolerance, &steps, &stable) != FS_OK) fail("pyflex.wait_until_stable");
        return py::make_tuple(stable != 0, steps);
    }, py::arg("max_steps") = 300, py::arg("tolerance") = 1e-2,
          "flex_utils.wait_until_stable (flex_utils.py:430-441) looped on the device; returns (stable, steps taken)");
    // white box for the tests and for bench_dropin.py: (live tenants of this process's device, back-end in force)
    m.def("_tenants", []() {
        const int n = g_device_key[0] ? fs_tenants_count(g_device_key, 0) : 0;
        return py::make_tuple(n, g_cotenant ? "cotenant" : "auto", std::string(g_device_key));
    }, "co-tenant table of the device (csrc/fs_tenants.cpp): (live processes, back-end in force, device key)");
    // a worker that ends without pyflex.clean (the reference's never call it) leaves the table by itself; one that is killed is
    // cleared by the next tenant that prunes
    py::module_::import("atexit").attr("register")(py::cpp_function([]() {
        if (g_device_key[0]) fs_tenants_unregister(g_device_key);
    }));
}
