// bindings.cpp — pybind11 surface `_psdr_core` with the reference's class / method names for the
// in-scope subset (reference src/psdr.cpp:120-439).  drjit arrays are replaced by numpy arrays for
// host data and by raw device pointers (ints) for images; the torch-facing conveniences (Matrix4fD
// stand-ins, autograd) live in psdr_jit_amd/__init__.py on top of this module.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "scene_host.h"
#include "exr_piz.h"

namespace py = pybind11;
using namespace py::literals;
using namespace psdr_host;

using farr = py::array_t<float, py::array::c_style | py::array::forcecast>;
using iarr = py::array_t<int, py::array::c_style | py::array::forcecast>;

static M16 to_m16(const farr &a) {
    if (a.size() != 16) throw Exception("expected a 4x4 matrix");
    M16 m; std::memcpy(m.data(), a.data(), 64); return m;
}
static farr from_m16(const M16 &m) { farr a({4, 4}); std::memcpy(a.mutable_data(), m.data(), 64); return a; }
static std::array<float, 3> to_a3(const farr &a) {
    if (a.size() == 1) return {a.data()[0], a.data()[0], a.data()[0]};
    if (a.size() != 3) throw Exception("expected 3 floats");
    return {a.data()[0], a.data()[1], a.data()[2]};
}
static farr from_vec(const std::vector<float> &v, ssize_t cols) {
    farr a({(ssize_t) v.size() / cols, cols});
    if (!v.empty()) std::memcpy(a.mutable_data(), v.data(), v.size() * sizeof(float));
    return a;
}
static iarr from_ivec(const std::vector<int> &v, ssize_t cols) {
    iarr a({(ssize_t) v.size() / cols, cols});
    if (!v.empty()) std::memcpy(a.mutable_data(), v.data(), v.size() * sizeof(int));
    return a;
}
static std::vector<float> to_fvec(const farr &a) { return std::vector<float>(a.data(), a.data() + a.size()); }
static std::vector<int> to_ivec(const iarr &a) { return std::vector<int>(a.data(), a.data() + a.size()); }

PYBIND11_MODULE(_psdr_core, m) {
    m.doc() = "Path-space differentiable renderer — MI355X host core";
    py::register_exception<Exception>(m, "PsdrException", PyExc_RuntimeError);

    m.def("hip_abi_version", []() { return psdr_hip_abi_version(); });
    m.def("hip_device_count", []() { return psdr_hip_device_count(); });
    m.def("hip_set_device", [](int d) { if (psdr_hip_set_device(d)) throw Exception(psdr_hip_last_error()); });

    py::class_<Object>(m, "Object", py::dynamic_attr())
        .def("type_name", &Object::type_name)
        .def_readwrite("id", &Object::m_id)
        .def("__repr__", &Object::to_string);

    py::class_<RenderOption>(m, "RenderOption")
        .def(py::init<>())
        .def(py::init<int, int, int>(), "width"_a, "height"_a, "spp/sppe"_a)
        .def(py::init<int, int, int, int>(), "width"_a, "height"_a, "spp"_a, "sppe"_a)
        .def(py::init<int, int, int, int, int>(), "width"_a, "height"_a, "spp"_a, "sppe"_a, "sppse"_a)
        .def_readwrite("width", &RenderOption::width).def_readwrite("height", &RenderOption::height)
        .def_readwrite("spp", &RenderOption::spp).def_readwrite("sppe", &RenderOption::sppe).def_readwrite("sppse", &RenderOption::sppse)
        .def_readwrite("log_level", &RenderOption::log_level)
        .def("__repr__", [](const RenderOption &ro) {
            std::ostringstream oss;
            oss << "[width: " << ro.width << ", height: " << ro.height << ", spp: " << ro.spp << ", sppe: " << ro.sppe << ", sppse: " << ro.sppse
                << ", log_level: " << ro.log_level << "]";
            return oss.str();
        });

    py::class_<BSDF, Object>(m, "BSDF", py::dynamic_attr()).def_readwrite("twoSide", &BSDF::m_twoSide).def("anisotropic", &BSDF::anisotropic)
        // Bitmap::m_rot / m_scale / m_trans of bitmap slot 0..2 as [rotate, scale, translate.x, translate.y] (reference psdr.cpp:204-206, 217-219)
        .def("_get_uv_xf", [](const BSDF &b, int slot, bool tangent) {
            if (slot < 0 || slot > 2) throw Exception("BSDF: bitmap slot out of range");
            farr a(4);
            std::memcpy(a.mutable_data(), tangent ? b.d_uv_xf[slot] : b.uv_xf[slot], 16);
            return a; })
        .def("_set_uv_xf", [](BSDF &b, int slot, const farr &v, const farr &t) {
            if (slot < 0 || slot > 2 || v.size() != 4) throw Exception("BSDF: a uv transform is [rotate, scale, translate.x, translate.y] of bitmap slot 0..2");
            std::memcpy(b.uv_xf[slot], v.data(), 16);
            if (t.size() == 4) std::memcpy(b.d_uv_xf[slot], t.data(), 16); else std::memset(b.d_uv_xf[slot], 0, 16); });
    py::class_<Diffuse, BSDF>(m, "DiffuseBSDF", py::dynamic_attr())
        .def(py::init<>())
        .def(py::init([](const farr &r) { return new Diffuse(to_a3(r)); }))
        .def("_get", [](const Diffuse &d, const std::string &, bool tangent) {
            if (d.tex_w > 0) {          // textured: [H, W, 3]
                farr a({(py::ssize_t) d.tex_h, (py::ssize_t) d.tex_w, (py::ssize_t) 3});
                const std::vector<float> &src = tangent ? d.d_tex : d.tex;
                if (src.size() == (size_t) a.size()) std::memcpy(a.mutable_data(), src.data(), sizeof(float) * src.size()); else std::memset(a.mutable_data(), 0, sizeof(float) * a.size());
                return a;
            }
            auto &r = tangent ? d.d_reflectance : d.reflectance; farr a(3); std::memcpy(a.mutable_data(), r.data(), 12); return a; })
        .def("_set", [](Diffuse &d, const std::string &, const farr &v, const farr &t) {
            if (v.ndim() == 3) {        // Bitmap3fD(width, height, data): a reflectance texture
                if (v.shape(2) != 3 || v.shape(0) < 2 || v.shape(1) < 2) throw Exception("Bitmap: invalid resolution!");
                d.tex_h = (int) v.shape(0); d.tex_w = (int) v.shape(1);
                d.tex.assign(v.data(), v.data() + v.size());
                if (t.size() == v.size()) d.d_tex.assign(t.data(), t.data() + t.size()); else d.d_tex.assign((size_t) v.size(), 0.f);
                return;
            }
            d.tex_w = d.tex_h = 0; d.tex.clear(); d.d_tex.clear();
            d.reflectance = to_a3(v); d.d_reflectance = to_a3(t); })
        .def_readonly("_tex_width", &Diffuse::tex_w).def_readonly("_tex_height", &Diffuse::tex_h);

    py::class_<Microfacet, BSDF>(m, "MicrofacetBSDF", py::dynamic_attr())
        .def(py::init<>())
        .def(py::init([](const farr &s, const farr &d, float r) { return new Microfacet(to_a3(s), to_a3(d), r); }))
        .def("_get", [](const Microfacet &b, const std::string &name, bool tangent) {
            const ParamTex &tx = name == "roughness" ? b.roughness_tex : (name == "specularReflectance" ? b.specular_tex : b.diffuse_tex);
            if (tx.w > 0) {             // a bitmap parameter: [H, W, 3] or [H, W]
                const std::vector<float> &src = tangent ? tx.d : tx.v;
                farr a = name == "roughness" ? farr({(py::ssize_t) tx.h, (py::ssize_t) tx.w}) : farr({(py::ssize_t) tx.h, (py::ssize_t) tx.w, (py::ssize_t) 3});
                if (src.size() == (size_t) a.size()) std::memcpy(a.mutable_data(), src.data(), sizeof(float) * src.size()); else std::memset(a.mutable_data(), 0, sizeof(float) * a.size());
                return a;
            }
            if (name == "roughness") { farr a(1); a.mutable_data()[0] = tangent ? b.d_roughness : b.roughness; return a; }
            const auto &r = name == "specularReflectance" ? (tangent ? b.d_specular : b.specular) : (tangent ? b.d_diffuse : b.diffuse);
            farr a(3); std::memcpy(a.mutable_data(), r.data(), 12); return a; })
        .def("_set", [](Microfacet &b, const std::string &name, const farr &v, const farr &t) {
            ParamTex &tx = name == "roughness" ? b.roughness_tex : (name == "specularReflectance" ? b.specular_tex : b.diffuse_tex);
            const int want = name == "roughness" ? 2 : 3;
            if (v.ndim() == want && v.ndim() >= 2) {          // Bitmap3fD / Bitmap1fD (width, height, data)
                if (v.shape(0) < 2 || v.shape(1) < 2 || (want == 3 && v.shape(2) != 3)) throw Exception("Bitmap: invalid resolution!");
                tx.h = (int) v.shape(0); tx.w = (int) v.shape(1);
                tx.v.assign(v.data(), v.data() + v.size());
                if (t.size() == v.size()) tx.d.assign(t.data(), t.data() + t.size()); else tx.d.assign((size_t) v.size(), 0.f);
                return;
            }
            tx = ParamTex();
            if (name == "roughness") { b.roughness = v.data()[0]; b.d_roughness = t.size() ? t.data()[0] : 0.f; }
            else if (name == "specularReflectance") { b.specular = to_a3(v); b.d_specular = to_a3(t); }
            else { b.diffuse = to_a3(v); b.d_diffuse = to_a3(t); } });

    py::class_<RoughConductor, BSDF>(m, "RoughConductorBSDF", py::dynamic_attr())
        .def(py::init<>())
        .def("_get", [](const RoughConductor &b, const std::string &name, bool tangent) {
            const ParamTex *tx = (name == "alpha_u" || name == "alpha_v") ? &b.alpha_tex : (name == "eta" ? &b.eta_tex : (name == "k" ? &b.k_tex : nullptr));
            if (tx && tx->w > 0) {
                const bool one = tx == &b.alpha_tex;
                farr a = one ? farr({(py::ssize_t) tx->h, (py::ssize_t) tx->w}) : farr({(py::ssize_t) tx->h, (py::ssize_t) tx->w, (py::ssize_t) 3});
                const std::vector<float> &src = tangent ? tx->d : tx->v;
                if (src.size() == (size_t) a.size()) std::memcpy(a.mutable_data(), src.data(), sizeof(float) * src.size()); else std::memset(a.mutable_data(), 0, sizeof(float) * a.size());
                return a;
            }
            if (name == "alpha_u" || name == "alpha_v") {
                farr a(1);
                a.mutable_data()[0] = name == "alpha_u" ? (tangent ? b.d_alpha_u : b.alpha_u) : (tangent ? b.d_alpha_v : b.alpha_v);
                return a;
            }
            const auto &r = name == "eta" ? (tangent ? b.d_eta : b.eta) : (name == "k" ? (tangent ? b.d_k : b.k) : (tangent ? b.d_specular : b.specular));
            farr a(3); std::memcpy(a.mutable_data(), r.data(), 12); return a; })
        .def("_set", [](RoughConductor &b, const std::string &name, const farr &v, const farr &t) {
            ParamTex *tx = (name == "alpha_u" || name == "alpha_v") ? &b.alpha_tex : (name == "eta" ? &b.eta_tex : (name == "k" ? &b.k_tex : nullptr));
            if (tx && v.ndim() == (tx == &b.alpha_tex ? 2 : 3) && v.shape(0) >= 2 && v.shape(1) >= 2) {     // a bitmap above 1x1
                if (tx != &b.alpha_tex && v.shape(2) != 3) throw Exception("Bitmap: invalid resolution!");
                tx->h = (int) v.shape(0); tx->w = (int) v.shape(1);
                tx->v.assign(v.data(), v.data() + v.size());
                if (t.size() == v.size()) tx->d.assign(t.data(), t.data() + t.size()); else tx->d.assign((size_t) v.size(), 0.f);
                return;
            }
            if (tx) *tx = ParamTex();
            if (name == "alpha_u") { b.alpha_u = v.data()[0]; b.d_alpha_u = t.size() ? t.data()[0] : 0.f; }
            else if (name == "alpha_v") { b.alpha_v = v.data()[0]; b.d_alpha_v = t.size() ? t.data()[0] : 0.f; }
            else if (name == "eta") { b.eta = to_a3(v); b.d_eta = to_a3(t); }
            else if (name == "k") { b.k = to_a3(v); b.d_k = to_a3(t); }
            else { b.specular = to_a3(v); b.d_specular = to_a3(t); } });

    py::class_<NormalMap, BSDF>(m, "NormalMapBSDF", py::dynamic_attr())
        .def(py::init<>())
        .def(py::init([](const farr &n) { return new NormalMap(to_a3(n)); }))
        .def_property_readonly("_nested", [](py::object self) {        // the owner is held by a visible attribute, see Scene.param_map
            py::object o = py::cast(self.cast<NormalMap &>().m_bsdf, py::return_value_policy::reference);
            if (!o.is_none()) o.attr("_psdr_owner") = self;
            return o; })
        .def("_set_nested", [](NormalMap &b, const BSDF *n) { b.set_nested(n); })
        .def("_get", [](const NormalMap &d, const std::string &, bool tangent) {
            if (d.tex_w > 0) {
                farr a({(py::ssize_t) d.tex_h, (py::ssize_t) d.tex_w, (py::ssize_t) 3});
                const std::vector<float> &src = tangent ? d.d_tex : d.tex;
                if (src.size() == (size_t) a.size()) std::memcpy(a.mutable_data(), src.data(), sizeof(float) * src.size()); else std::memset(a.mutable_data(), 0, sizeof(float) * a.size());
                return a;
            }
            auto &r = tangent ? d.d_normal : d.normal; farr a(3); std::memcpy(a.mutable_data(), r.data(), 12); return a; })
        .def("_set", [](NormalMap &d, const std::string &, const farr &v, const farr &t) {
            if (v.ndim() == 3) {
                if (v.shape(2) != 3 || v.shape(0) < 2 || v.shape(1) < 2) throw Exception("Bitmap: invalid resolution!");
                d.tex_h = (int) v.shape(0); d.tex_w = (int) v.shape(1);
                d.tex.assign(v.data(), v.data() + v.size());
                if (t.size() == v.size()) d.d_tex.assign(t.data(), t.data() + t.size()); else d.d_tex.assign((size_t) v.size(), 0.f);
                return;
            }
            d.tex_w = d.tex_h = 0; d.tex.clear(); d.d_tex.clear();
            d.normal = to_a3(v); d.d_normal = to_a3(t); });

    py::class_<MicrofacetPerVertex, BSDF>(m, "MicrofacetBSDFPerVertex", py::dynamic_attr())
        .def(py::init<>())
        .def("_get", [](const MicrofacetPerVertex &b, const std::string &name, bool tangent) {
            const bool r = name == "roughness";
            const std::vector<float> &v = r ? b.roughness : (name == "specularReflectance" ? b.specular : b.diffuse);
            const std::vector<float> &d = r ? b.d_roughness : (name == "specularReflectance" ? b.d_specular : b.d_diffuse);
            const py::ssize_t n = (py::ssize_t) (r ? v.size() : v.size() / 3);
            farr a = r ? farr(n) : farr({n, (py::ssize_t) 3});
            const std::vector<float> &src = tangent ? d : v;
            if (src.size() == (size_t) a.size()) std::memcpy(a.mutable_data(), src.data(), sizeof(float) * src.size()); else if (a.size()) std::memset(a.mutable_data(), 0, sizeof(float) * a.size());
            return a; })
        .def("_set", [](MicrofacetPerVertex &b, const std::string &name, const farr &v, const farr &t) {
            const bool r = name == "roughness";
            if (!r && (v.size() % 3) != 0) throw Exception("MicrofacetBSDFPerVertex: expected [n, 3] values");
            std::vector<float> &dst = r ? b.roughness : (name == "specularReflectance" ? b.specular : b.diffuse);
            std::vector<float> &ddst = r ? b.d_roughness : (name == "specularReflectance" ? b.d_specular : b.d_diffuse);
            dst.assign(v.data(), v.data() + v.size());
            if (t.size() == v.size()) ddst.assign(t.data(), t.data() + t.size()); else ddst.clear(); });

    py::class_<RoughDielectric, BSDF>(m, "RoughDielectricBSDF", py::dynamic_attr())
        .def(py::init<>())
        .def(py::init<float, float>())
        .def("_get", [](const RoughDielectric &b, const std::string &name, bool tangent) {
            if ((name == "alpha_u" || name == "alpha_v") && b.alpha_tex.w > 0) {
                farr m2({(py::ssize_t) b.alpha_tex.h, (py::ssize_t) b.alpha_tex.w});
                const std::vector<float> &src = tangent ? b.alpha_tex.d : b.alpha_tex.v;
                if (src.size() == (size_t) m2.size()) std::memcpy(m2.mutable_data(), src.data(), sizeof(float) * src.size()); else std::memset(m2.mutable_data(), 0, sizeof(float) * m2.size());
                return m2;
            }
            farr a(1);
            a.mutable_data()[0] = name == "alpha_u" ? (tangent ? b.d_alpha_u : b.alpha_u) : name == "alpha_v" ? (tangent ? b.d_alpha_v : b.alpha_v)
                                : name == "eta" ? (tangent ? b.d_eta : b.eta) : (tangent ? b.d_inv_eta : b.inv_eta);
            return a; })
        .def("_set", [](RoughDielectric &b, const std::string &name, const farr &v, const farr &t) {
            if ((name == "alpha_u" || name == "alpha_v") && v.ndim() == 2 && v.shape(0) >= 2 && v.shape(1) >= 2) {
                ParamTex &tx = b.alpha_tex;
                tx.h = (int) v.shape(0); tx.w = (int) v.shape(1);
                tx.v.assign(v.data(), v.data() + v.size());
                if (t.size() == v.size()) tx.d.assign(t.data(), t.data() + t.size()); else tx.d.assign((size_t) v.size(), 0.f);
                return;
            }
            if (name == "alpha_u" || name == "alpha_v") b.alpha_tex = ParamTex();
            const float x = v.data()[0], dx = t.size() ? t.data()[0] : 0.f;
            if (name == "alpha_u") { b.alpha_u = x; b.d_alpha_u = dx; }
            else if (name == "alpha_v") { b.alpha_v = x; b.d_alpha_v = dx; }
            else if (name == "eta") { b.eta = x; b.d_eta = dx; b.inv_eta = 1.f / x; b.d_inv_eta = -dx / (x * x); }
            else { b.inv_eta = x; b.d_inv_eta = dx; } });

    py::class_<Emitter, Object>(m, "Emitter", py::dynamic_attr());
    py::class_<AreaLight, Emitter>(m, "AreaLight", py::dynamic_attr())
        .def(py::init([](const farr &r) { return new AreaLight(to_a3(r)); }))
        .def_readonly("sampling_weight", &AreaLight::m_sampling_weight)
        .def("_get", [](const AreaLight &d, const std::string &, bool tangent) { auto &r = tangent ? d.d_radiance : d.radiance; farr a(3); std::memcpy(a.mutable_data(), r.data(), 12); return a; })
        .def("_set", [](AreaLight &d, const std::string &, const farr &v, const farr &t) { d.radiance = to_a3(v); d.d_radiance = to_a3(t); });

    auto get_xf = [](const Transformable &t, const std::string &name, bool tangent) -> farr {
        if (name == "to_world") return from_m16(tangent ? t.d_to_world_raw : t.to_world_raw);
        if (name == "to_world_left") return from_m16(tangent ? t.d_to_world_left : t.to_world_left);
        if (name == "to_world_right") return from_m16(tangent ? t.d_to_world_right : t.to_world_right);
        throw Exception("unknown parameter: " + name);
    };
    auto set_xf = [](Transformable &t, const std::string &name, const farr &v, const farr &d) {
        if (name == "to_world") { t.to_world_raw = to_m16(v); t.d_to_world_raw = to_m16(d); }
        else if (name == "to_world_left") { t.to_world_left = to_m16(v); t.d_to_world_left = to_m16(d); }
        else if (name == "to_world_right") { t.to_world_right = to_m16(v); t.d_to_world_right = to_m16(d); }
        else throw Exception("unknown parameter: " + name);
    };

    py::class_<Sensor, Object>(m, "Sensor", py::dynamic_attr())
        .def("_get", [get_xf](const Sensor &s, const std::string &n, bool tg) { return get_xf(s, n, tg); })
        .def("_set", [set_xf](Sensor &s, const std::string &n, const farr &v, const farr &d) { set_xf(s, n, v, d); })
        .def("_set_transform", [](Sensor &s, const farr &v, const farr &d, bool left) { s.set_transform(to_m16(v), to_m16(d), left); })
        .def("_append_transform", [](Sensor &s, const farr &v, const farr &d, bool left) { s.append_transform(to_m16(v), to_m16(d), left); })
        .def_readonly("enable_edges", &Sensor::m_enable_edges);
    m.def("_make_orthographic", [](float near_, float far_) { auto *c = new PerspectiveCamera(0.f, near_, far_); c->m_orthographic = true; return c; },
          py::return_value_policy::take_ownership);
    py::class_<PerspectiveCamera, Sensor>(m, "PerspectiveCamera", py::dynamic_attr())
        .def_readonly("orthographic", &PerspectiveCamera::m_orthographic)
        .def(py::init<float, float, float>())
        .def_property_readonly("world_to_sample", [](const PerspectiveCamera &c) { farr a({4, 4}); std::memcpy(a.mutable_data(), c.rec.world_to_sample, 64); return a; })
        .def("_direct_params", [](const PerspectiveCamera &c) {        // what sample_direct needs (perspective.cpp:181-197)
            return py::make_tuple(c.m_width, c.m_height, c.rec.cam_pos[0], c.rec.cam_pos[1], c.rec.cam_pos[2], c.rec.cam_dir[0], c.rec.cam_dir[1], c.rec.cam_dir[2], c.rec.inv_area); })
        .def("_primary_edge_ids", [](const PerspectiveCamera &c) { return from_ivec(c.m_edges.ids, 3); })
        .def("_camera_params", [](const PerspectiveCamera &c) { return py::make_tuple(c.m_fov_x, c.m_near_clip, c.m_far_clip); })
        .def("_primary_edges", [](const PerspectiveCamera &c, bool tangent) {
            const PrimaryEdges &e = c.m_edges;
            const size_t n = e.length.size();
            farr a({(ssize_t) n, (ssize_t) 7});
            float *o = a.mutable_data();
            for (size_t i = 0; i < n; ++i) {
                const std::vector<float> &p0 = tangent ? e.d_p0 : e.p0, &p1 = tangent ? e.d_p1 : e.p1;
                o[7 * i] = p0[2 * i]; o[7 * i + 1] = p0[2 * i + 1]; o[7 * i + 2] = p1[2 * i]; o[7 * i + 3] = p1[2 * i + 1];
                o[7 * i + 4] = tangent ? 0.f : e.normal[2 * i]; o[7 * i + 5] = tangent ? 0.f : e.normal[2 * i + 1]; o[7 * i + 6] = tangent ? 0.f : e.length[i];
            }
            return a;
        });

    py::class_<Mesh, Object>(m, "Mesh", py::dynamic_attr())
        .def(py::init<>())
        .def("load", &Mesh::load, "filename"_a, "verbose"_a = false)
        .def("load_raw", [](Mesh &me, const farr &v, const iarr &f, const farr &uv, const iarr &fuv, bool verbose) {
            me.load_raw(to_fvec(v), to_ivec(f), to_fvec(uv), to_ivec(fuv), verbose); }, "v"_a, "f"_a, "uv"_a = farr(0), "f_uv"_a = iarr(0), "verbose"_a = false)
        .def("configure", &Mesh::configure)
        .def("dump", &Mesh::dump)
        .def("_get", [get_xf](const Mesh &me, const std::string &n, bool tg) -> farr {
            if (n == "vertex_positions") return from_vec(tg ? me.d_vertex_positions_raw : me.vertex_positions_raw, 3);
            return get_xf(me, n, tg); })
        .def("_set", [set_xf](Mesh &me, const std::string &n, const farr &v, const farr &d) {
            if (n == "vertex_positions") {
                if ((ssize_t) v.size() != 3 * (ssize_t) me.m_num_vertices || d.size() != v.size()) throw Exception("vertex_positions: size mismatch");
                me.vertex_positions_raw = to_fvec(v); me.d_vertex_positions_raw = to_fvec(d); me.m_ready = false; return;
            }
            set_xf(me, n, v, d); me.m_ready = false; })
        .def("_set_transform", [](Mesh &me, const farr &v, const farr &d, bool left) { me.set_transform(to_m16(v), to_m16(d), left); me.m_ready = false; })
        .def("_append_transform", [](Mesh &me, const farr &v, const farr &d, bool left) { me.append_transform(to_m16(v), to_m16(d), left); me.m_ready = false; })
        .def_readonly("num_vertices", &Mesh::m_num_vertices)
        .def_readonly("num_faces", &Mesh::m_num_faces)
        .def_readonly("bsdf", &Mesh::m_bsdf)
        .def_property_readonly("vertex_positions_T", [](const Mesh &me) { return from_vec(me.vertex_positions, 3); })
        .def_property_readonly("vertex_normals", [](const Mesh &me) { return from_vec(me.vertex_normals_raw, 3); })
        .def_property_readonly("vertex_uv", [](const Mesh &me) { return from_vec(me.vertex_uv, 2); })
        .def_property_readonly("face_indices", [](const Mesh &me) { return from_ivec(me.face_indices, 3); })
        .def_property_readonly("face_uv_indices", [](const Mesh &me) { return from_ivec(me.face_uv_indices, 3); })
        .def_readwrite("use_face_normal", &Mesh::m_use_face_normals)
        .def_readwrite("enable_edges", &Mesh::m_enable_edges)
        .def("num_edges", [](const Mesh &me) { return (int64_t) me.edges.size(); })
        .def("edge_indices", [](const Mesh &me) {
            iarr a({(ssize_t) me.edges.size(), (ssize_t) 5});
            int *o = a.mutable_data();
            for (size_t i = 0; i < me.edges.size(); ++i) { const MeshEdge &e = me.edges[i]; o[5 * i] = e.v0; o[5 * i + 1] = e.v1; o[5 * i + 2] = e.f0; o[5 * i + 3] = e.f1; o[5 * i + 4] = e.opp; }
            return a; });

    py::class_<EnvironmentMap, Emitter>(m, "EnvironmentMap", py::dynamic_attr())
        .def(py::init<>())
        .def(py::init([](const farr &rgb) {
            if (rgb.ndim() != 3 || rgb.shape(2) != 3) throw Exception("EnvironmentMap: radiance must be a [height, width, 3] array");
            std::vector<float> d(rgb.data(), rgb.data() + rgb.size());
            return new EnvironmentMap((int) rgb.shape(1), (int) rgb.shape(0), d); }))
        .def_readonly("sampling_weight", &EnvironmentMap::m_sampling_weight)
        .def_readonly("width", &EnvironmentMap::width).def_readonly("height", &EnvironmentMap::height)
        .def("_get", [](const EnvironmentMap &e, const std::string &name, bool tangent) {
            if (name == "radiance") {
                farr a({(py::ssize_t) e.height, (py::ssize_t) e.width, (py::ssize_t) 3});
                const std::vector<float> &src = tangent ? e.d_data : e.data;
                if (src.size() == (size_t) a.size()) std::memcpy(a.mutable_data(), src.data(), sizeof(float) * src.size()); else std::memset(a.mutable_data(), 0, sizeof(float) * a.size());
                return a;
            }
            if (name == "scale") { farr a(1); a.mutable_data()[0] = tangent ? e.d_scale : e.scale; return a; }
            if (name == "radiance_uv_xf") { farr a(4); std::memcpy(a.mutable_data(), tangent ? e.d_uv_xf : e.uv_xf, 16); return a; }
            farr a({4, 4});
            if (name == "to_world_left") std::memcpy(a.mutable_data(), (tangent ? e.d_to_world_left : e.to_world_left).data(), 64);
            else if (tangent) std::memset(a.mutable_data(), 0, 64);
            else std::memcpy(a.mutable_data(), e.to_world_raw.data(), 64);
            return a; })
        .def("_set", [](EnvironmentMap &e, const std::string &name, const farr &v, const farr &t) {
            if (name == "radiance") {
                if (v.ndim() != 3 || v.shape(2) != 3) throw Exception("EnvironmentMap: radiance must be a [height, width, 3] array");
                const bool same = e.height == (int) v.shape(0) && e.width == (int) v.shape(1) && e.data.size() == (size_t) v.size() &&
                                  std::memcmp(e.data.data(), v.data(), sizeof(float) * e.data.size()) == 0;
                e.height = (int) v.shape(0); e.width = (int) v.shape(1);
                if (!same) { e.data.assign(v.data(), v.data() + v.size()); e.m_cells_dirty = true; }      // (a tangent-only update keeps the cell distribution)
                if (t.size() == v.size()) e.d_data.assign(t.data(), t.data() + t.size()); else e.d_data.clear();
            } else if (name == "scale") { e.scale = v.data()[0]; e.d_scale = t.size() ? t.data()[0] : 0.f; }
            else if (name == "radiance_uv_xf") {          // m_radiance.rotate / scale / translate: the cell masses are those of the transformed map
                if (v.size() != 4) throw Exception("EnvironmentMap: a uv transform is [rotate, scale, translate.x, translate.y]");
                if (std::memcmp(e.uv_xf, v.data(), 16) != 0) { std::memcpy(e.uv_xf, v.data(), 16); e.m_cells_dirty = true; }
                if (t.size() == 4) std::memcpy(e.d_uv_xf, t.data(), 16); else std::memset(e.d_uv_xf, 0, 16);
            }
            else if (name == "to_world_left") { e.to_world_left = to_m16(v); e.d_to_world_left = t.size() == 16 ? to_m16(t) : zeros16(); }
            else e.to_world_raw = to_m16(v);
            e.m_ready = false; });

    py::class_<Scene, Object>(m, "Scene", py::dynamic_attr())
        .def(py::init<>())
        .def("_add_EnvironmentMap", &Scene::add_EnvironmentMap)
        .def("add_Sensor", &Scene::add_Sensor, "Add Sensor")
        .def("_add_Mesh_file", [](Scene &s, const std::string &f, const farr &t, const std::string &b, const Emitter *e) { s.add_Mesh(f, to_m16(t), b, e); })
        .def("_add_Mesh_obj", [](Scene &s, const Mesh *mesh, const std::string &b, const Emitter *e) { s.add_Mesh(mesh, b, e); })
        .def("add_BSDF", &Scene::add_BSDF, "Add BSDF", "bsdf"_a, "name"_a, "twoSide"_a = false)
        .def("_add_normalmap_BSDF", &Scene::add_normalmap_BSDF, "bsdf1"_a, "bsdf2"_a, "name"_a, "twoSide"_a = false)
        .def("_configure", &Scene::configure, "active_sensor"_a = std::vector<int>(), py::call_guard<py::gil_scoped_release>())
        .def("_configure_host", &Scene::configure_host, "active_sensor"_a = std::vector<int>())
        .def("is_ready", &Scene::is_ready)
        .def_readwrite("opts", &Scene::m_opts, "Render options")
        .def_readwrite("seed", &Scene::seed, "Sample seed")
        .def_readonly("num_sensors", &Scene::m_num_sensors)
        .def_readonly("num_meshes", &Scene::m_num_meshes)
        .def("get_num_emitters", &Scene::get_num_emitters)
        .def_property_readonly("aabb", [](const Scene &s) {      // Scene::m_lower / m_upper, what configure() logs as "AABB"
            farr a({2, 3});
            for (int k = 0; k < 3; ++k) { a.mutable_data()[k] = s.m_lower[k]; a.mutable_data()[3 + k] = s.m_upper[k]; }
            return a; })
        .def_property_readonly("param_map", [](Scene &s) {
            py::dict d;
            // The wrappers keep their scene alive through a PYTHON attribute (a reference the garbage collector can see) instead
            // of pybind11's hidden keep-alive list: the Python layer stores the wrappers in the scene's __dict__, and a hidden
            // back reference made every populated Scene (host arrays, device blob, BVH) immortal.
            py::object self = py::cast(&s, py::return_value_policy::reference);
            for (auto &kv : s.m_param_map) {
                py::object o = py::cast(kv.second, py::return_value_policy::reference);
                o.attr("_psdr_scene") = self;
                d[py::str(kv.first)] = o;
            }
            return d; }, "Parameter map")
        .def("_sampler_state", [](const Scene &s, int k) { return py::make_tuple(s.m_samplers[k].ready, s.m_samplers[k].sample_count, s.m_samplers[k].seed, s.m_samplers[k].skip); })
        .def("_set_sampler_state", [](Scene &s, int k, bool ready, int64_t count, uint64_t seed, uint64_t skip) {
            if (k < 0 || k > 2) throw Exception("sampler index");
            s.m_samplers[k] = SamplerState{ready, count, seed, skip}; })
        .def("_bvh_stats", [](const Scene &s) {
            int32_t n = 0, l = 0, d = 0, b = 0;
            if (!s.m_hip || psdr_hip_scene_stats(s.m_hip, &n, &l, &d, &b)) throw Exception("scene not configured");
            return py::make_tuple(n, l, d, b); })
        // Scene::chain_geometry: float32 arrays in (rows of the snapshot), per wanted mesh (index, g_to_world [4, 4], g_vertices [nv, 3]) and g_world_to_sample out
        .def("_chain_geometry", [](const Scene &s, int sensor_id, const farr &g_tri, const farr &g_sec, const farr &g_prim, const std::vector<int> &want_mesh, bool want_camera,
                                  const py::array_t<double, py::array::c_style | py::array::forcecast> &w2s) {
            std::vector<uint8_t> want(s.m_meshes.size(), 0);
            if (w2s.size() != 0 && w2s.size() != 16) throw Exception("_chain_geometry: world_to_sample is 4 x 4");
            for (int i : want_mesh) if (i >= 0 && (size_t) i < want.size()) want[(size_t) i] = 1;
            if ((size_t) g_tri.size() != 22 * s.snap.area.size()) throw Exception("_chain_geometry: g_tri does not match the snapshot");
            if ((size_t) g_sec.size() < 6 * (size_t) s.snap.n_sec_edges) throw Exception("_chain_geometry: g_sec does not match the snapshot");
            Scene::GeometryAdjoint ga;
            {
                py::gil_scoped_release rel;
                ga = s.chain_geometry(sensor_id, g_tri.data(), g_sec.data(), g_prim.data(), want, want_camera, w2s.size() == 16 ? w2s.data() : nullptr);
            }
            py::list meshes;
            for (const Scene::MeshAdjoint &m : ga.meshes) {
                py::array_t<double> gm({4, 4}), gv({(ssize_t) m.g_vertices.size() / 3, (ssize_t) 3});
                std::memcpy(gm.mutable_data(), m.g_to_world, sizeof(double) * 16);
                if (!m.g_vertices.empty()) std::memcpy(gv.mutable_data(), m.g_vertices.data(), sizeof(double) * m.g_vertices.size());
                meshes.append(py::make_tuple(m.mesh, gm, gv));
            }
            py::array_t<double> gw({4, 4});
            std::memcpy(gw.mutable_data(), ga.g_world_to_sample, sizeof(double) * 16);
            return py::make_tuple(meshes, gw); })
        // (triangles, secondary edges) of the configured snapshot: what reverse mode sizes its adjoint buffers with
        .def("_snapshot_counts", [](const Scene &s) { return py::make_tuple((int64_t) s.snap.area.size(), (int64_t) s.snap.n_sec_edges); })
        .def("_hip_handle", [](const Scene &s) { return (uintptr_t) s.m_hip; })
        .def("_check_device_rows", &Scene::check_device_rows)
        // what the last configure() did to the device copy (psdr_update_info, include/psdr_hip.h)
        .def("_last_update", [](const Scene &s) {
            py::dict d;
            const psdr_update_info &u = s.m_last_update;
            d["tree"] = u.tree == 2 ? "built" : (u.tree == 1 ? "refitted" : "kept");
            d["reallocated"] = u.reallocated; d["bytes_uploaded"] = u.bytes_uploaded; d["sah_cost"] = u.sah_cost; d["sah_cost_built"] = u.sah_cost_built;
            d["ms_host"] = s.m_ms_host; d["ms_tree"] = u.ms_tree; d["ms_fill"] = u.ms_fill; d["ms_upload"] = u.ms_upload; d["ms_total"] = u.ms_total;
            return d; })
        .def_readwrite("_always_rebuild", &Scene::m_always_rebuild)
        .def("_snapshot", [](Scene &s) {
            // configured host arrays in the oracle's row formats (tests compare them with the CPU restatement)
            s.ensure_full_snapshot();                 // (a lean configure leaves the rows to whoever asks)
            const Scene::Snapshot &S = s.snap;
            const size_t n = S.area.size();
            farr tri({(ssize_t) n, (ssize_t) 22}), dtri({(ssize_t) n, (ssize_t) 22});
            const std::vector<float> *src[7] = {&S.p0, &S.e1, &S.e2, &S.n0, &S.n1, &S.n2, &S.fn};
            const std::vector<float> *dsrc[7] = {&S.d_p0, &S.d_e1, &S.d_e2, &S.d_n0, &S.d_n1, &S.d_n2, &S.d_fn};
            for (size_t i = 0; i < n; ++i) {
                for (int k = 0; k < 7; ++k) for (int c = 0; c < 3; ++c) { tri.mutable_data()[22 * i + 3 * k + c] = (*src[k])[3 * i + c]; dtri.mutable_data()[22 * i + 3 * k + c] = (*dsrc[k])[3 * i + c]; }
                tri.mutable_data()[22 * i + 21] = S.area[i]; dtri.mutable_data()[22 * i + 21] = S.d_area[i];
            }
            const size_t ne = (size_t) S.n_sec_edges;
            farr se({(ssize_t) ne, (ssize_t) 16}), dse({(ssize_t) ne, (ssize_t) 16});
            for (size_t i = 0; i < ne; ++i) {
                float *o = se.mutable_data() + 16 * i, *d = dse.mutable_data() + 16 * i;
                for (int c = 0; c < 3; ++c) {
                    o[c] = S.se_p0[3 * i + c]; o[3 + c] = S.se_e1[3 * i + c]; o[6 + c] = S.se_n0[3 * i + c]; o[9 + c] = S.se_n1[3 * i + c]; o[12 + c] = S.se_p2[3 * i + c];
                    d[c] = S.se_d_p0[3 * i + c]; d[3 + c] = S.se_d_e1[3 * i + c]; d[6 + c] = d[9 + c] = d[12 + c] = 0.f;
                }
                o[15] = S.se_boundary[i] ? 1.f : 0.f; d[15] = 0.f;
            }
            py::dict out;
            out["triangles"] = tri; out["d_triangles"] = dtri; out["sec_edges"] = se; out["d_sec_edges"] = dse;
            out["mesh_id"] = py::array_t<int32_t>(S.mesh_id.size(), S.mesh_id.data());
            out["uv"] = from_vec(S.uv, 6);
            out["sec_edge_cmf"] = from_vec(S.sec_edge_distrb.cmf, 1);
            out["face_cmf"] = from_vec(S.face_cmf, 1);
            std::vector<float> ew;
            for (const psdr_emitter_rec &e : S.emitters) ew.push_back(e.sampling_weight);
            out["emitter_weights"] = from_vec(ew, 1);
            {   // BSDF rows as the kernels see them: (type, nested row, bitmap widths of the three slots, per-vertex count)
                py::list rows;
                for (const psdr_bsdf_rec &b : S.bsdfs)
                    rows.append(py::make_tuple(b.type, b.type == 5 ? b.nested_bsdf : -1, b.tex_data ? b.tex_width : 0, b.spec_tex_data ? b.spec_tex_width : 0,
                                               b.rough_tex_data ? b.rough_tex_width : 0, b.type == 4 ? b.pv_count : 0));
                out["bsdf_rows"] = rows;
                out["face_indices"] = py::array_t<int32_t>(S.face_indices.size(), S.face_indices.data());
            }
            if (S.has_envmap && s.m_emitter_env) {
                const EnvironmentMap &E = *s.m_emitter_env;
                out["env_cell_pmf"] = from_vec(E.cell_distrb.pmf, 1); out["env_cell_cmf"] = from_vec(E.cell_distrb.cmf, 1);
                out["env_cell_sum"] = E.cell_distrb.sum;
                std::vector<float> b = {E.lower[0], E.lower[1], E.lower[2], E.upper[0], E.upper[1], E.upper[2]};
                out["env_bounds"] = from_vec(b, 1);
                out["env_reso"] = py::make_tuple(E.reso[0], E.reso[1]);
            }
            return out; });

    py::class_<Integrator, Object>(m, "Integrator", py::dynamic_attr())
        .def("_renderC", &Integrator::renderC, "scene"_a, "sensor_id"_a, "seed"_a, "pix_ids"_a, "n_pix"_a, "out"_a, "stream"_a, "shard_rank"_a, "shard_count"_a,
             py::call_guard<py::gil_scoped_release>())
        .def("_renderD", &Integrator::renderD, "scene"_a, "sensor_id"_a, "seed"_a, "pix_ids"_a, "n_pix"_a, "out"_a, "dout"_a, "stream"_a, "shard_rank"_a,
             "shard_count"_a, "terms"_a, py::call_guard<py::gil_scoped_release>())
        .def_readwrite("trace_static_edges", &Integrator::m_trace_static_edges)
        .def_readwrite("_shard_mode", &Integrator::m_shard_mode);

    m.def("_render_d_bwd", [](const Integrator &it, const Scene &scene, int sensor_id, const std::vector<uint64_t> &seeds, const std::vector<uint64_t> &skips,
                              uintptr_t d_rgb, uintptr_t g_tri, uintptr_t g_bsdf, uintptr_t g_emitter, uintptr_t g_sec, uintptr_t g_prim, uintptr_t stream,
                              int rank, int count, int terms, uintptr_t mesh_filter, bool skip_bsdf, bool skip_emitter, uintptr_t g_tex, uintptr_t g_cam, uintptr_t g_env, uintptr_t g_env_scale, uintptr_t g_mat, uintptr_t g_env_xf, uintptr_t pix_ids, int n_pix, uintptr_t g_uv_xf, uintptr_t prim_filter) {
        if (!scene.is_ready()) throw Exception("Input scene must be configured!");
        psdr_render_args a;
        std::memset(&a, 0, sizeof(a));
        a.sensor_id = sensor_id; a.max_depth = it.max_depth(); a.hide_emitters = it.hide_emitters() ? 1 : 0;
        for (int k = 0; k < 3; ++k) { a.samplers[k].seed = seeds[k]; a.samplers[k].skip = skips[k]; }
        a.shard_rank = rank; a.shard_count = count; a.shard_mode = it.m_shard_mode; a.zero_output = 1; a.terms = terms; a.guiding = it.guiding(sensor_id);
        a.direct_mode = it.direct_mis() + 1;
        a.pix_ids = reinterpret_cast<const int32_t *>(pix_ids); a.n_pix = pix_ids ? n_pix : 0;       // batch rendering: interior term only
        a.field_mode = it.field() + 1; a.field_object = field_object_index(scene, it); a.intensity = it.intensity(false);
        psdr_grads g{reinterpret_cast<float *>(g_tri), reinterpret_cast<float *>(g_bsdf), reinterpret_cast<float *>(g_emitter),
                     reinterpret_cast<float *>(g_sec), reinterpret_cast<float *>(g_prim),
                     reinterpret_cast<const uint8_t *>(mesh_filter), skip_bsdf ? 1 : 0, skip_emitter ? 1 : 0, reinterpret_cast<float *>(g_tex), reinterpret_cast<float *>(g_cam),
                     reinterpret_cast<float *>(g_env), reinterpret_cast<float *>(g_env_scale), reinterpret_cast<float *>(g_mat), reinterpret_cast<float *>(g_env_xf), reinterpret_cast<float *>(g_uv_xf),
                     reinterpret_cast<const uint8_t *>(prim_filter)};
        if (psdr_hip_render_d_bwd(scene.m_hip, &a, reinterpret_cast<const float *>(d_rgb), &g, reinterpret_cast<void *>(stream)))
            throw Exception(std::string("libpsdr_hip: ") + psdr_hip_last_error());
    });

    // one PIZ chunk of an EXR file -> its uncompressed 16-bit words (exr_piz.cpp; used by psdr_jit_amd/exr.py)
    m.def("_piz_decode", [](const py::bytes &chunk, int nx, int ny, const std::vector<int> &words_per_sample) {
        const std::string buf = chunk;
        size_t total = 0;
        for (int w : words_per_sample) total += (size_t) nx * ny * w;
        py::array_t<uint16_t> out((py::ssize_t) total);
        try {
            piz_decode_chunk(reinterpret_cast<const uint8_t *>(buf.data()), buf.size(), nx, ny, words_per_sample, out.mutable_data());
        } catch (const std::runtime_error &e) { throw Exception(e.what()); }
        return out;
    });

    // Scene::ray_intersect<false> for device arrays of rays (psdr_hip_ray_intersect)
    m.def("_ray_intersect", [](const Scene &scene, int n, uintptr_t o, uintptr_t d, uintptr_t out, uintptr_t stream) {
        if (!scene.is_ready()) throw Exception("Input scene must be configured!");
        if (psdr_hip_ray_intersect(scene.m_hip, n, reinterpret_cast<const float *>(o), reinterpret_cast<const float *>(d), reinterpret_cast<float *>(out),
                                   reinterpret_cast<void *>(stream)))
            throw Exception(std::string("libpsdr_hip: ") + psdr_hip_last_error());
    });

    // (offsets[3*n_bsdfs], total) of psdr_hip_scene_tex_layout
    m.def("_tex_layout", [](const Scene &scene) {
        if (!scene.is_ready()) throw Exception("Input scene must be configured!");
        // rows = the snapshot's BSDFs: the scene's own, then the ones nested in normal maps
        const size_t nb = scene.snap.bsdfs.size();
        std::vector<int64_t> off((size_t) 3 * std::max<size_t>(1, nb), -1);
        int64_t total = 0;
        if (psdr_hip_scene_tex_layout(scene.m_hip, off.data(), &total)) throw Exception(std::string("libpsdr_hip: ") + psdr_hip_last_error());
        off.resize((size_t) 3 * nb);
        return py::make_tuple(off, total);
    });

    py::class_<PathTracer, Integrator>(m, "PathTracer", py::dynamic_attr())
        .def(py::init<int>(), "max_depth"_a = 1)
        .def("preprocess_secondary_edges", &PathTracer::preprocess_secondary_edges, "scene"_a, "sensor_id"_a, "resolution"_a, "nrounds"_a = 1, "seed"_a = 0)
        .def("_guiding_mass", [](const PathTracer &p, int sid) { auto v = p.guiding_mass(sid); return from_vec(v, 1); })
        .def("_guiding_handle", [](const PathTracer &p, int sid) { return reinterpret_cast<uintptr_t>(p.guiding(sid)); })    // psdr_render_args.guiding for direct C-ABI calls
        .def_readwrite("hide_emitters", &PathTracer::m_hide_emitters)
        .def_readonly("max_depth", &PathTracer::m_max_depth);
    py::class_<FieldExtractionIntegrator, Integrator>(m, "FieldExtractionIntegrator", py::dynamic_attr())
        .def(py::init<const std::string &>())
        .def_readonly("field", &FieldExtractionIntegrator::m_field_name)
        .def_readonly("object", &FieldExtractionIntegrator::m_object);
    py::class_<CollocatedIntegrator, Integrator>(m, "CollocatedIntegrator", py::dynamic_attr())
        .def(py::init<float>())
        .def("_get", [](const CollocatedIntegrator &c, const std::string &, bool tangent) { farr a(1); a.mutable_data()[0] = tangent ? c.d_intensity : c.m_intensity; return a; })
        .def("_set", [](CollocatedIntegrator &c, const std::string &, const farr &v, const farr &t) { c.m_intensity = v.data()[0]; c.d_intensity = t.size() ? t.data()[0] : 0.f; });
    py::class_<DirectIntegrator, PathTracer>(m, "Direct", py::dynamic_attr())
        .def(py::init<int>(), "mis"_a = 2)
        .def_readonly("mis", &DirectIntegrator::m_mis);
}
