"""Builds the PRODUCT scene (psdr_jit_amd, host C++ + HIP) from the same neutral SceneSpec the
oracle consumes, through the reference-style Python surface."""
import numpy as np


def build_scene(spec, configure=True, active=(0,), host_only=False, log_level=0):
    import psdr_jit_amd as psdr
    sc = psdr.Scene()
    sc.opts.width, sc.opts.height = spec.width, spec.height
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse = spec.spp, spec.sppe, spec.sppse
    sc.opts.log_level = log_level
    for c in spec.cameras:
        cam = psdr.OrthographicCamera(c.near, c.far) if getattr(c, "orthographic", False) else psdr.PerspectiveCamera(c.fov_x, c.near, c.far)
        cam._set("to_world", c.to_world_raw, c.d_to_world_raw)
        cam._set("to_world_left", c.to_world_left, c.d_to_world_left)
        cam._set("to_world_right", c.to_world_right, c.d_to_world_right)
        sc.add_Sensor(cam)
    def plain_bsdf(b):
        """a stand-alone psdr BSDF object for one BsdfSpec of type 0-3 (used for the BSDF nested in a NormalMap)"""
        f1 = lambda x: np.asarray([x], np.float32)
        t = getattr(b, "type", 0)
        if t == 1:
            bs = psdr.MicrofacetBSDF(list(b.specular), list(b.reflectance), float(b.roughness))
            bs._set("specularReflectance", np.asarray(b.specular, np.float32), np.asarray(b.d_specular, np.float32))
            bs._set("diffuseReflectance", np.asarray(b.reflectance, np.float32), np.asarray(b.d_reflectance, np.float32))
            bs._set("roughness", f1(b.roughness), f1(b.d_roughness))
        elif t == 2:
            bs = psdr.RoughConductorBSDF()
            bs._set("alpha_u", f1(b.alpha_u), f1(b.d_alpha_u)); bs._set("alpha_v", f1(b.alpha_v), f1(b.d_alpha_v))
            bs._set("eta", np.asarray(b.eta, np.float32), np.asarray(b.d_eta, np.float32))
            bs._set("k", np.asarray(b.k, np.float32), np.asarray(b.d_k, np.float32))
            bs._set("specular_reflectance", np.asarray(b.specular, np.float32), np.asarray(b.d_specular, np.float32))
        elif t == 0:
            bs = psdr.DiffuseBSDF(list(b.reflectance))
            bs._set("reflectance", np.asarray(b.reflectance, np.float32), np.asarray(b.d_reflectance, np.float32))
        else:
            raise ValueError("nested BSDF type %d" % t)
        bs.twoSide = bool(b.two_sided)
        return bs

    def put_xf(bs, b):
        # rotate / scale / translate of the three bitmap slots (BsdfSpec.tex_xf) and their forward tangents
        if getattr(b, "tex_xf", None) is None and getattr(b, "d_tex_xf", None) is None:
            return
        xf = np.asarray(b.tex_xf if getattr(b, "tex_xf", None) is not None else [[0, 1, 0, 0]] * 3, np.float32).reshape(3, 4)
        dxf = np.asarray(b.d_tex_xf if getattr(b, "d_tex_xf", None) is not None else np.zeros((3, 4)), np.float32).reshape(3, 4)
        for k in range(3):
            bs._set_uv_xf(k, np.ascontiguousarray(xf[k]), np.ascontiguousarray(dxf[k]))

    for i, b in enumerate(spec.bsdfs):
        if getattr(b, "type", 0) == 5:              # NormalMapBSDF over spec.bsdfs[b.nested] (which the scene also holds as its own entry)
            nm = psdr.NormalMapBSDF(list(b.reflectance))
            nm._set("normal_map", np.asarray(b.reflectance, np.float32), np.asarray(b.d_reflectance, np.float32))
            if getattr(b, "texture", None) is not None:
                tex = np.ascontiguousarray(np.asarray(b.texture, np.float32))
                dtex = np.ascontiguousarray(np.asarray(b.d_texture, np.float32)) if getattr(b, "d_texture", None) is not None else np.zeros_like(tex)
                nm._set("normal_map", tex, dtex)
            nm.nested_bsdf = plain_bsdf(spec.bsdfs[b.nested])
            put_xf(nm, b)
            sc.add_BSDF(nm, b.name or ("bsdf%d" % i), b.two_sided)
            continue
        if getattr(b, "type", 0) == 1:
            bs = psdr.MicrofacetBSDF(list(b.specular), list(b.reflectance), float(b.roughness))
            bs._set("specularReflectance", np.asarray(b.specular, np.float32), np.asarray(b.d_specular, np.float32))
            bs._set("diffuseReflectance", np.asarray(b.reflectance, np.float32), np.asarray(b.d_reflectance, np.float32))
            bs._set("roughness", np.asarray([b.roughness], np.float32), np.asarray([b.d_roughness], np.float32))
            for attr, name in (("texture", "diffuseReflectance"), ("spec_texture", "specularReflectance"), ("rough_texture", "roughness")):
                t = getattr(b, attr, None)
                if t is not None:
                    tex = np.ascontiguousarray(np.asarray(t, np.float32))
                    dt = getattr(b, "d_" + attr, None)
                    bs._set(name, tex, np.ascontiguousarray(np.asarray(dt, np.float32)) if dt is not None else np.zeros_like(tex))
            put_xf(bs, b)
            sc.add_BSDF(bs, b.name or ("bsdf%d" % i), b.two_sided)
            continue
        if getattr(b, "type", 0) == 2:
            bs = psdr.RoughConductorBSDF()
            f1 = lambda x: np.asarray([x], np.float32)
            bs._set("alpha_u", f1(b.alpha_u), f1(b.d_alpha_u)); bs._set("alpha_v", f1(b.alpha_v), f1(b.d_alpha_v))
            bs._set("eta", np.asarray(b.eta, np.float32), np.asarray(b.d_eta, np.float32))
            bs._set("k", np.asarray(b.k, np.float32), np.asarray(b.d_k, np.float32))
            bs._set("specular_reflectance", np.asarray(b.specular, np.float32), np.asarray(b.d_specular, np.float32))
            for attr, name in (("texture", "eta"), ("spec_texture", "k"), ("rough_texture", "alpha_u")):       # bitmap parameters
                t = getattr(b, attr, None)
                if t is not None:
                    tex = np.ascontiguousarray(np.asarray(t, np.float32))
                    dt = getattr(b, "d_" + attr, None)
                    bs._set(name, tex, np.ascontiguousarray(np.asarray(dt, np.float32)) if dt is not None else np.zeros_like(tex))
            put_xf(bs, b)
            sc.add_BSDF(bs, b.name or ("bsdf%d" % i), b.two_sided)
            continue
        if getattr(b, "type", 0) == 4:
            f2 = lambda x, w: np.ascontiguousarray(np.asarray(x, np.float32).reshape(-1, w) if w == 3 else np.asarray(x, np.float32).reshape(-1))
            bs = psdr.MicrofacetBSDFPerVertex(f2(b.pv_specular, 3), f2(b.pv_diffuse, 3), f2(b.pv_roughness, 1))
            for attr, name, w in (("pv_specular", "specularReflectance", 3), ("pv_diffuse", "diffuseReflectance", 3), ("pv_roughness", "roughness", 1)):
                d = getattr(b, "d_" + attr, None)
                bs._set(name, f2(getattr(b, attr), w), f2(d, w) if d is not None else np.zeros_like(f2(getattr(b, attr), w)))
            put_xf(bs, b)
            sc.add_BSDF(bs, b.name or ("bsdf%d" % i), b.two_sided)
            continue
        if getattr(b, "type", 0) == 3:
            bs = psdr.RoughDielectricBSDF()
            f1 = lambda x: np.asarray([x], np.float32)
            bs._set("alpha_u", f1(b.alpha_u), f1(b.d_alpha_u)); bs._set("alpha_v", f1(b.alpha_v), f1(b.d_alpha_v))
            bs._set("eta", f1(b.eta[0]), f1(b.d_eta[0])); bs._set("inv_eta", f1(b.eta[1]), f1(b.d_eta[1]))
            if getattr(b, "rough_texture", None) is not None:
                tex = np.ascontiguousarray(np.asarray(b.rough_texture, np.float32))
                dt = getattr(b, "d_rough_texture", None)
                bs._set("alpha_u", tex, np.ascontiguousarray(np.asarray(dt, np.float32)) if dt is not None else np.zeros_like(tex))
            put_xf(bs, b)
            sc.add_BSDF(bs, b.name or ("bsdf%d" % i), b.two_sided)
            continue
        bs = psdr.DiffuseBSDF(list(b.reflectance))
        bs._set("reflectance", np.asarray(b.reflectance, np.float32), np.asarray(b.d_reflectance, np.float32))
        if getattr(b, "texture", None) is not None:
            tex = np.ascontiguousarray(np.asarray(b.texture, np.float32))
            dtex = np.ascontiguousarray(np.asarray(b.d_texture, np.float32)) if getattr(b, "d_texture", None) is not None else np.zeros_like(tex)
            bs._set("reflectance", tex, dtex)
        put_xf(bs, b)
        sc.add_BSDF(bs, b.name or ("bsdf%d" % i), b.two_sided)
    def add_envs_before(k):
        # emitters are numbered in the order they are added: keep the spec's numbering
        for i, e in enumerate(spec.emitters):
            if getattr(e, "type", 0) == 1 and i < k and i not in added_env:
                env = psdr.EnvironmentMap(np.asarray(e.env_data, np.float32))
                env.scale = float(e.env_scale)
                env.to_world = np.asarray(e.env_to_world_raw, np.float32)
                env.to_world_left = np.asarray(e.env_to_world_left, np.float32)
                # forward tangents of the differentiable members (texels, scale, to_world_left)
                if getattr(e, "d_env_data", None) is not None:
                    env._set("radiance", np.ascontiguousarray(np.asarray(e.env_data, np.float32)), np.ascontiguousarray(np.asarray(e.d_env_data, np.float32)))
                env._set("scale", np.asarray([e.env_scale], np.float32), np.asarray([getattr(e, "d_env_scale", 0.0)], np.float32))
                env._set("radiance_uv_xf", np.asarray(getattr(e, "env_uv_xf", (0, 1, 0, 0)), np.float32), np.asarray(getattr(e, "d_env_uv_xf", (0, 0, 0, 0)), np.float32))
                env._set("to_world_left", np.asarray(e.env_to_world_left, np.float32), np.asarray(getattr(e, "d_env_to_world_left", np.zeros((4, 4))), np.float32))
                sc.add_EnvironmentMap(env)
                added_env.add(i)

    added_env = set()
    for m in spec.meshes:
        if m.emitter >= 0:
            add_envs_before(m.emitter)
        mesh = psdr.Mesh()
        mesh.enable_edges = m.enable_edges
        if m.path is not None:
            mesh.load(m.path)
        else:
            mesh.load_raw(m.vertices, m.faces, m.uvs if m.uvs is not None else np.zeros((0, 2), np.float32),
                          m.face_uvs if m.face_uvs is not None else np.zeros((0, 3), np.int32))
        mesh.use_face_normal = m.use_face_normals
        mesh._set("to_world", m.to_world_raw, m.d_to_world_raw)
        mesh._set("to_world_left", m.to_world_left, m.d_to_world_left)
        mesh._set("to_world_right", m.to_world_right, m.d_to_world_right)
        if m.d_vertices is not None:
            mesh._set("vertex_positions", m.vertices, m.d_vertices)
        em = None
        if m.emitter >= 0:
            e = spec.emitters[m.emitter]
            em = psdr.AreaLight(list(e.radiance))
            em._set("radiance", np.asarray(e.radiance, np.float32), np.asarray(e.d_radiance, np.float32))
        b = spec.bsdfs[m.bsdf]
        sc.add_Mesh(mesh, b.name or ("bsdf%d" % m.bsdf), em)
    add_envs_before(len(spec.emitters))
    if configure:
        if host_only:
            sc._configure_host(list(active))
        else:
            sc._configure(list(active))
            sc.__dict__["_psdr_active"] = list(active)
    return sc


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
