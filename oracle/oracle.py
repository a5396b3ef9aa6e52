"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes driver for oracle/_build/liboracle.so (the CPU restatement of psdr-jit's
PathTracer.renderC / renderD hot path, see oracle/oracle.h).  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product package.

Pinned against the figures and log lines the reference's tutorial notebooks embed (tests/test_oracle_notebooks.py); the reference
itself cannot be built or imported here, see oracle/README.md.
"""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

TERM_INTERIOR, TERM_PRIMARY, TERM_SECONDARY = 1, 2, 4
TERM_ALL = 7


def build(force: bool = False, native: bool = False) -> str:
    """Compile the oracle with g++ (a few seconds).  native=True builds and SELECTS the -march=native variant
    (bit-identical results, BASELINE.md §3's CPU-baseline protocol): call it before the first OracleScene."""
    global _LIB_PATH, _lib
    args = ["make", "-C", _HERE] + (["-B"] if force else []) + (["native"] if native else [])
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    if native:
        path = os.path.join(_HERE, "_build", "liboracle_native.so")
        if path != _LIB_PATH:
            _LIB_PATH, _lib = path, None
    return _LIB_PATH


# ------------------------------------------------------------------ neutral scene description
def _eye():
    return np.eye(4, dtype=np.float32)


def _zero4():
    return np.zeros((4, 4), dtype=np.float32)


@dataclass
class MeshSpec:
    vertices: np.ndarray                  # (n,3) float32, object space
    faces: np.ndarray                     # (m,3) int32
    uvs: Optional[np.ndarray] = None      # (k,2) float32
    face_uvs: Optional[np.ndarray] = None  # (m,3) int32
    to_world_left: np.ndarray = field(default_factory=_eye)
    to_world_raw: np.ndarray = field(default_factory=_eye)
    to_world_right: np.ndarray = field(default_factory=_eye)
    d_to_world_left: np.ndarray = field(default_factory=_zero4)
    d_to_world_raw: np.ndarray = field(default_factory=_zero4)
    d_to_world_right: np.ndarray = field(default_factory=_zero4)
    d_vertices: Optional[np.ndarray] = None
    bsdf: int = 0
    emitter: int = -1
    use_face_normals: bool = False
    enable_edges: bool = True
    path: Optional[str] = None            # OBJ file the arrays came from (for the product API)


@dataclass
class BsdfSpec:
    reflectance: tuple = (0.5, 0.5, 0.5)
    d_reflectance: tuple = (0.0, 0.0, 0.0)
    two_sided: bool = False
    name: str = ""
    texture: Optional[np.ndarray] = None      # [H, W, 3] reflectance texture (Bitmap3fD), overrides `reflectance`
    d_texture: Optional[np.ndarray] = None    # tangent of the texels
    type: int = 0                             # 0 = DiffuseBSDF, 1 = MicrofacetBSDF(specular, diffuse = reflectance, roughness), 2 = RoughConductorBSDF
    specular: tuple = (0.04, 0.04, 0.04)
    d_specular: tuple = (0.0, 0.0, 0.0)
    roughness: float = 0.5
    d_roughness: float = 0.0
    alpha_u: float = 0.1                      # type 2 = RoughConductorBSDF(alpha_u, alpha_v, eta, k, specular = specular_reflectance)
    alpha_v: float = 0.1
    d_alpha_u: float = 0.0
    d_alpha_v: float = 0.0
    eta: tuple = (0.0, 0.0, 0.0)
    d_eta: tuple = (0.0, 0.0, 0.0)
    k: tuple = (1.0, 1.0, 1.0)
    d_k: tuple = (0.0, 0.0, 0.0)
    spec_texture: Optional[np.ndarray] = None   # MicrofacetBSDF bitmap parameters: [H, W, 3] specular reflectance,
    d_spec_texture: Optional[np.ndarray] = None
    rough_texture: Optional[np.ndarray] = None  # [H, W] roughness (`texture` is then its diffuse reflectance map)
    d_rough_texture: Optional[np.ndarray] = None
    pv_specular: Optional[np.ndarray] = None    # type 4 = MicrofacetBSDFPerVertex: [n, 3], [n, 3], [n] per mesh-local vertex
    pv_diffuse: Optional[np.ndarray] = None
    pv_roughness: Optional[np.ndarray] = None
    d_pv_specular: Optional[np.ndarray] = None
    d_pv_diffuse: Optional[np.ndarray] = None
    d_pv_roughness: Optional[np.ndarray] = None
    nested: int = -1                            # type 5 = NormalMapBSDF: index of the nested BSDF; `reflectance` / `texture` = the normal map
    # uv transform of the bitmaps `texture`, `spec_texture`, `rough_texture`: [3][rotate, scale, translate.x, translate.y] (Bitmap::m_rot,
    # m_scale, m_trans, reference bitmap.h:37-39) and its tangent; None = identity / zero
    tex_xf: Optional[np.ndarray] = None
    d_tex_xf: Optional[np.ndarray] = None


@dataclass
class EmitterSpec:
    """AreaLight (type 0, attached to a mesh through MeshSpec.emitter_id) or EnvironmentMap (type 1: env_data is the
    lat-long radiance image [H, W, 3]; the bounding cube is added by configure, as in the reference)."""
    radiance: tuple = (1.0, 1.0, 1.0)
    d_radiance: tuple = (0.0, 0.0, 0.0)
    type: int = 0
    env_data: Optional[np.ndarray] = None
    env_scale: float = 1.0
    env_to_world_left: np.ndarray = field(default_factory=_eye)
    env_to_world_raw: np.ndarray = field(default_factory=_eye)
    d_env_data: Optional[np.ndarray] = None        # tangents of the texels / scale / to_world_left
    d_env_scale: float = 0.0
    d_env_to_world_left: np.ndarray = field(default_factory=lambda: np.zeros((4, 4), dtype=np.float32))
    env_uv_xf: tuple = (0.0, 1.0, 0.0, 0.0)        # m_radiance's rotate, scale, translate.x, translate.y (+ tangent)
    d_env_uv_xf: tuple = (0.0, 0.0, 0.0, 0.0)


@dataclass
class CameraSpec:
    fov_x: float = 60.0
    near: float = 1e-6
    far: float = 1e7
    to_world_left: np.ndarray = field(default_factory=_eye)
    to_world_raw: np.ndarray = field(default_factory=_eye)
    to_world_right: np.ndarray = field(default_factory=_eye)
    d_to_world_left: np.ndarray = field(default_factory=_zero4)
    d_to_world_raw: np.ndarray = field(default_factory=_zero4)
    d_to_world_right: np.ndarray = field(default_factory=_zero4)
    orthographic: bool = False            # OrthographicCamera(near, far): fov_x unused


@dataclass
class SceneSpec:
    meshes: List[MeshSpec]
    bsdfs: List[BsdfSpec]
    emitters: List[EmitterSpec]
    cameras: List[CameraSpec]
    width: int = 128
    height: int = 128
    spp: int = 1
    sppe: int = 0
    sppse: int = 0


# ------------------------------------------------------------------ ctypes mirror of oracle.h
_F16 = C.c_float * 16
_F3 = C.c_float * 3


class _Mesh(C.Structure):
    _fields_ = [("n_vertices", C.c_int), ("n_faces", C.c_int), ("n_uvs", C.c_int),
                ("vertices", C.POINTER(C.c_float)), ("d_vertices", C.POINTER(C.c_float)),
                ("faces", C.POINTER(C.c_int)), ("uvs", C.POINTER(C.c_float)), ("face_uvs", C.POINTER(C.c_int)),
                ("to_world_left", _F16), ("to_world_raw", _F16), ("to_world_right", _F16),
                ("d_to_world_left", _F16), ("d_to_world_raw", _F16), ("d_to_world_right", _F16),
                ("bsdf_id", C.c_int), ("emitter_id", C.c_int), ("use_face_normals", C.c_int), ("enable_edges", C.c_int)]


class _Bsdf(C.Structure):
    _fields_ = [("type", C.c_int), ("reflectance", _F3), ("d_reflectance", _F3), ("two_sided", C.c_int),
                ("tex_width", C.c_int), ("tex_height", C.c_int), ("tex_data", C.POINTER(C.c_float)), ("d_tex_data", C.POINTER(C.c_float)),
                ("specular", _F3), ("d_specular", _F3), ("roughness", C.c_float), ("d_roughness", C.c_float),
                ("alpha_u", C.c_float), ("alpha_v", C.c_float), ("d_alpha_u", C.c_float), ("d_alpha_v", C.c_float),
                ("eta", _F3), ("d_eta", _F3), ("k", _F3), ("d_k", _F3),
                ("spec_tex_width", C.c_int), ("spec_tex_height", C.c_int), ("spec_tex_data", C.POINTER(C.c_float)), ("d_spec_tex_data", C.POINTER(C.c_float)),
                ("rough_tex_width", C.c_int), ("rough_tex_height", C.c_int), ("rough_tex_data", C.POINTER(C.c_float)), ("d_rough_tex_data", C.POINTER(C.c_float)),
                ("pv_count", C.c_int), ("pv_specular", C.POINTER(C.c_float)), ("pv_diffuse", C.POINTER(C.c_float)), ("pv_roughness", C.POINTER(C.c_float)),
                ("d_pv_specular", C.POINTER(C.c_float)), ("d_pv_diffuse", C.POINTER(C.c_float)), ("d_pv_roughness", C.POINTER(C.c_float)),
                ("nested_bsdf", C.c_int), ("tex_xf", (C.c_float * 4) * 3), ("d_tex_xf", (C.c_float * 4) * 3)]


class _Emitter(C.Structure):
    _fields_ = [("radiance", _F3), ("d_radiance", _F3), ("type", C.c_int), ("env_width", C.c_int), ("env_height", C.c_int),
                ("env_data", C.POINTER(C.c_float)), ("env_scale", C.c_float), ("env_to_world_left", _F16), ("env_to_world_raw", _F16),
                ("d_env_data", C.POINTER(C.c_float)), ("d_env_scale", C.c_float), ("d_env_to_world_left", _F16),
                ("env_uv_xf", C.c_float * 4), ("d_env_uv_xf", C.c_float * 4)]


class _Camera(C.Structure):
    _fields_ = [("fov_x", C.c_float), ("near_clip", C.c_float), ("far_clip", C.c_float),
                ("to_world_left", _F16), ("to_world_raw", _F16), ("to_world_right", _F16),
                ("d_to_world_left", _F16), ("d_to_world_raw", _F16), ("d_to_world_right", _F16), ("orthographic", C.c_int)]


class _Desc(C.Structure):
    _fields_ = [("n_meshes", C.c_int), ("meshes", C.POINTER(_Mesh)),
                ("n_bsdfs", C.c_int), ("bsdfs", C.POINTER(_Bsdf)),
                ("n_emitters", C.c_int), ("emitters", C.POINTER(_Emitter)),
                ("n_cameras", C.c_int), ("cameras", C.POINTER(_Camera)),
                ("width", C.c_int), ("height", C.c_int), ("spp", C.c_int), ("sppe", C.c_int), ("sppse", C.c_int)]


class _Sampler(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("skip", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.POINTER(_Desc), C.POINTER(C.c_int), C.c_int]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_get_num_threads.restype = C.c_int
        for name in ("orc_num_triangles", "orc_num_sec_edges"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_int
        L.orc_num_primary_edges.argtypes = [C.c_void_p, C.c_int]
        L.orc_num_primary_edges.restype = C.c_int
        L.orc_get_triangle_info.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_get_sec_edges.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_get_primary_edges.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_get_mesh_edges.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_get_mesh_edges.restype = C.c_int
        L.orc_emitter_sampling_weight.argtypes = [C.c_void_p, C.c_int]
        L.orc_emitter_sampling_weight.restype = C.c_float
        L.orc_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_render_c.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _Sampler, C.c_void_p, C.c_int,
                                   C.c_int64, C.c_int64, C.c_void_p]
        L.orc_render_c.restype = C.c_int
        L.orc_render_d.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(_Sampler), C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_render_d.restype = C.c_int
        L.orc_li_lanes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _Sampler, C.c_int64, C.c_int64, C.c_void_p]
        L.orc_li_lanes.restype = C.c_int
        L.orc_guiding_build.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]
        L.orc_guiding_build.restype = C.c_void_p
        L.orc_guiding_num_cells.argtypes = [C.c_void_p]
        L.orc_guiding_num_cells.restype = C.c_int
        L.orc_guiding_get_mass.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_guiding_destroy.argtypes = [C.c_void_p]
        L.orc_tea64.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_tea64.restype = C.c_uint64
        L.orc_pcg32_raw.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        L.orc_sampler_floats.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        L.orc_square_to_cosine_hemisphere.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_square_to_uniform_triangle.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_coordinate_system.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_distrb_sample_reuse.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_distrb_sample_reuse.restype = C.c_int
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int)) if a is not None else None


def _m16(m):
    return _F16(*[float(x) for x in np.asarray(m, dtype=np.float32).reshape(16)])


def set_num_threads(n: int):
    lib().orc_set_num_threads(int(n))


def get_num_threads() -> int:
    return lib().orc_get_num_threads()


FIELDS = ["silhouette", "position", "depth", "geoNormal", "shNormal", "uv", "bsdf", "segmentation", "collocated"]


class OracleScene:
    """A configured scene (= Scene.configure(active_sensors) of the reference)."""

    def __init__(self, spec: SceneSpec, active_sensors=(0,)):
        L = lib()
        self.spec = spec
        self._keep = []
        meshes = (_Mesh * len(spec.meshes))()
        for i, m in enumerate(spec.meshes):
            v, f = _f32(m.vertices), _i32(m.faces)
            dv = _f32(m.d_vertices) if m.d_vertices is not None else None
            uv = _f32(m.uvs) if m.uvs is not None else None
            fuv = _i32(m.face_uvs) if m.face_uvs is not None else None
            self._keep += [v, f, dv, uv, fuv]
            mm = meshes[i]
            mm.n_vertices, mm.n_faces = v.shape[0], f.shape[0]
            mm.n_uvs = uv.shape[0] if uv is not None else 0
            mm.vertices, mm.d_vertices, mm.faces = _fp(v), _fp(dv), _ip(f)
            mm.uvs, mm.face_uvs = _fp(uv), _ip(fuv)
            mm.to_world_left, mm.to_world_raw, mm.to_world_right = _m16(m.to_world_left), _m16(m.to_world_raw), _m16(m.to_world_right)
            mm.d_to_world_left, mm.d_to_world_raw, mm.d_to_world_right = _m16(m.d_to_world_left), _m16(m.d_to_world_raw), _m16(m.d_to_world_right)
            mm.bsdf_id, mm.emitter_id = m.bsdf, m.emitter
            mm.use_face_normals, mm.enable_edges = int(m.use_face_normals), int(m.enable_edges)
        bsdfs = (_Bsdf * max(1, len(spec.bsdfs)))()
        for i, b in enumerate(spec.bsdfs):
            bsdfs[i].type = int(getattr(b, "type", 0))
            bsdfs[i].nested_bsdf = int(getattr(b, "nested", -1))
            xf = np.asarray(getattr(b, "tex_xf", None) if getattr(b, "tex_xf", None) is not None else [[0, 1, 0, 0]] * 3, dtype=np.float32).reshape(3, 4)
            dxf = np.asarray(getattr(b, "d_tex_xf", None) if getattr(b, "d_tex_xf", None) is not None else np.zeros((3, 4)), dtype=np.float32).reshape(3, 4)
            for k in range(3):
                for q in range(4):
                    bsdfs[i].tex_xf[k][q], bsdfs[i].d_tex_xf[k][q] = float(xf[k, q]), float(dxf[k, q])
            bsdfs[i].specular = _F3(*getattr(b, "specular", (0.04, 0.04, 0.04)))
            bsdfs[i].d_specular = _F3(*getattr(b, "d_specular", (0.0, 0.0, 0.0)))
            bsdfs[i].roughness = float(getattr(b, "roughness", 0.5))
            bsdfs[i].d_roughness = float(getattr(b, "d_roughness", 0.0))
            bsdfs[i].alpha_u, bsdfs[i].alpha_v = float(b.alpha_u), float(b.alpha_v)
            bsdfs[i].d_alpha_u, bsdfs[i].d_alpha_v = float(b.d_alpha_u), float(b.d_alpha_v)
            bsdfs[i].eta, bsdfs[i].d_eta, bsdfs[i].k, bsdfs[i].d_k = _F3(*b.eta), _F3(*b.d_eta), _F3(*b.k), _F3(*b.d_k)
            bsdfs[i].reflectance = _F3(*b.reflectance)
            bsdfs[i].d_reflectance = _F3(*b.d_reflectance)
            bsdfs[i].two_sided = int(b.two_sided)
            if getattr(b, "texture", None) is not None:
                tex = np.ascontiguousarray(np.asarray(b.texture, dtype=np.float32))
                assert tex.ndim == 3 and tex.shape[2] == 3
                self._keep.append(tex)
                bsdfs[i].tex_height, bsdfs[i].tex_width = tex.shape[0], tex.shape[1]
                bsdfs[i].tex_data = tex.ctypes.data_as(C.POINTER(C.c_float))
                if getattr(b, "d_texture", None) is not None:
                    dt = np.ascontiguousarray(np.asarray(b.d_texture, dtype=np.float32))
                    assert dt.shape == tex.shape
                    self._keep.append(dt)
                    bsdfs[i].d_tex_data = dt.ctypes.data_as(C.POINTER(C.c_float))
            for name, ch in (("spec", 3), ("rough", 1)):      # Microfacet bitmap parameters
                t = getattr(b, name + "_texture", None)
                if t is None:
                    continue
                tex = np.ascontiguousarray(np.asarray(t, dtype=np.float32))
                tex = tex.reshape(tex.shape[0], tex.shape[1], ch)
                self._keep.append(tex)
                setattr(bsdfs[i], name + "_tex_height", tex.shape[0]); setattr(bsdfs[i], name + "_tex_width", tex.shape[1])
                setattr(bsdfs[i], name + "_tex_data", tex.ctypes.data_as(C.POINTER(C.c_float)))
                dt = getattr(b, "d_" + name + "_texture", None)
                if dt is not None:
                    dt = np.ascontiguousarray(np.asarray(dt, dtype=np.float32)).reshape(tex.shape)
                    self._keep.append(dt)
                    setattr(bsdfs[i], "d_" + name + "_tex_data", dt.ctypes.data_as(C.POINTER(C.c_float)))
            if int(getattr(b, "type", 0)) == 4:               # MicrofacetBSDFPerVertex
                n = len(np.asarray(b.pv_roughness).reshape(-1))
                bsdfs[i].pv_count = n
                for name, width in (("pv_specular", 3), ("pv_diffuse", 3), ("pv_roughness", 1)):
                    for pre in ("", "d_"):
                        a = getattr(b, pre + name, None)
                        if a is None:
                            continue
                        a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(n, width))
                        self._keep.append(a)
                        setattr(bsdfs[i], pre + name, a.ctypes.data_as(C.POINTER(C.c_float)))
        emitters = (_Emitter * max(1, len(spec.emitters)))()
        for i, e in enumerate(spec.emitters):
            emitters[i].radiance = _F3(*e.radiance)
            emitters[i].d_radiance = _F3(*e.d_radiance)
            emitters[i].type = int(e.type)
            emitters[i].env_to_world_left, emitters[i].env_to_world_raw = _m16(e.env_to_world_left), _m16(e.env_to_world_raw)
            emitters[i].env_scale = float(e.env_scale)
            emitters[i].d_env_scale = float(getattr(e, "d_env_scale", 0.0))
            emitters[i].env_uv_xf = (C.c_float * 4)(*[float(q) for q in getattr(e, "env_uv_xf", (0.0, 1.0, 0.0, 0.0))])
            emitters[i].d_env_uv_xf = (C.c_float * 4)(*[float(q) for q in getattr(e, "d_env_uv_xf", (0.0, 0.0, 0.0, 0.0))])
            emitters[i].d_env_to_world_left = _m16(getattr(e, "d_env_to_world_left", np.zeros((4, 4), np.float32)))
            if e.type == 1 and getattr(e, "d_env_data", None) is not None:
                dimg = np.ascontiguousarray(np.asarray(e.d_env_data, dtype=np.float32))
                assert dimg.shape == np.asarray(e.env_data).shape
                self._keep.append(dimg)
                emitters[i].d_env_data = dimg.ctypes.data_as(C.POINTER(C.c_float))
            if e.type == 1:
                img = np.ascontiguousarray(np.asarray(e.env_data, dtype=np.float32))
                assert img.ndim == 3 and img.shape[2] == 3
                self._keep.append(img)
                emitters[i].env_height, emitters[i].env_width = img.shape[0], img.shape[1]
                emitters[i].env_data = img.ctypes.data_as(C.POINTER(C.c_float))
        cams = (_Camera * len(spec.cameras))()
        for i, c in enumerate(spec.cameras):
            cams[i].fov_x, cams[i].near_clip, cams[i].far_clip = c.fov_x, c.near, c.far
            cams[i].orthographic = int(getattr(c, "orthographic", False))
            cams[i].to_world_left, cams[i].to_world_raw, cams[i].to_world_right = _m16(c.to_world_left), _m16(c.to_world_raw), _m16(c.to_world_right)
            cams[i].d_to_world_left, cams[i].d_to_world_raw, cams[i].d_to_world_right = _m16(c.d_to_world_left), _m16(c.d_to_world_raw), _m16(c.d_to_world_right)
        desc = _Desc(len(spec.meshes), meshes, len(spec.bsdfs), bsdfs, len(spec.emitters), emitters, len(spec.cameras), cams,
                     spec.width, spec.height, spec.spp, spec.sppe, spec.sppse)
        act = (C.c_int * max(1, len(active_sensors)))(*active_sensors)
        self._h = L.orc_scene_create(C.byref(desc), act, len(active_sensors))
        if not self._h:
            raise RuntimeError(L.orc_last_error().decode())

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_scene_destroy(self._h)
                self._h = None
        except Exception:      # interpreter shutdown
            pass

    # ---- snapshot introspection
    @property
    def num_triangles(self):
        return lib().orc_num_triangles(self._h)

    @property
    def num_sec_edges(self):
        return lib().orc_num_sec_edges(self._h)

    def num_primary_edges(self, sensor=0):
        return lib().orc_num_primary_edges(self._h, sensor)

    def aabb(self):
        """(lower[3], upper[3]) of Scene::m_lower / m_upper"""
        L = lib()
        L.orc_scene_aabb.restype = None
        L.orc_scene_aabb.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        b = (C.c_float * 6)()
        L.orc_scene_aabb(self._h, b)
        return np.array(b[:3], np.float32), np.array(b[3:], np.float32)

    def envmap_info(self):
        """(bounds[6], reso[2], cell_sum, pmf, cmf) of the configured EnvironmentMap, or None"""
        b = (C.c_float * 6)(); r = (C.c_int * 2)(); cs = C.c_float()
        L = lib()
        L.orc_envmap_info.restype = C.c_int
        L.orc_envmap_info.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L.orc_envmap_cells.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_envmap_cells.restype = None
        if not L.orc_envmap_info(self._h, b, r, C.byref(cs)):
            return None
        n = r[0] * r[1]
        pmf, cmf = np.empty(n, np.float32), np.empty(n, np.float32)
        lib().orc_envmap_cells(self._h, pmf.ctypes.data_as(C.POINTER(C.c_float)), cmf.ctypes.data_as(C.POINTER(C.c_float)))
        return np.array(list(b), np.float32), (r[0], r[1]), float(cs.value), pmf, cmf

    def env_sample(self, ref_p, s2):
        L = lib()
        ref_p = np.ascontiguousarray(ref_p, np.float32); s2 = np.ascontiguousarray(s2, np.float32)
        n = len(ref_p)
        p, nn, pdf = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32), np.empty(n, np.float32)
        fp = C.POINTER(C.c_float)
        L.orc_env_sample.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, fp, fp]; L.orc_env_sample.restype = None
        L.orc_env_sample(self._h, n, ref_p.ctypes.data_as(fp), s2.ctypes.data_as(fp), p.ctypes.data_as(fp), nn.ctypes.data_as(fp), pdf.ctypes.data_as(fp))
        return p, nn, pdf

    def env_pdf(self, ref_p, p, nrm):
        L = lib()
        ref_p = np.ascontiguousarray(ref_p, np.float32); p = np.ascontiguousarray(p, np.float32); nrm = np.ascontiguousarray(nrm, np.float32)
        n = len(ref_p)
        pdf = np.empty(n, np.float32)
        fp = C.POINTER(C.c_float)
        L.orc_env_pdf.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, fp]; L.orc_env_pdf.restype = None
        L.orc_env_pdf(self._h, n, ref_p.ctypes.data_as(fp), p.ctypes.data_as(fp), nrm.ctypes.data_as(fp), pdf.ctypes.data_as(fp))
        return pdf

    def set_direct_mis(self, mis):
        """-1: PathTracer (default); 0/1/2: DirectIntegrator(mis) for the render calls that follow"""
        L = lib()
        L.orc_set_direct_mis.argtypes = [C.c_void_p, C.c_int]; L.orc_set_direct_mis.restype = None
        L.orc_set_direct_mis(self._h, int(mis))

    def set_field(self, field=-1, obj=-1, intensity=1.0, d_intensity=0.0):
        """first-hit integrators: see orc_set_field (field names: FIELDS)"""
        L = lib()
        L.orc_set_field.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]; L.orc_set_field.restype = None
        if isinstance(field, str):
            field = FIELDS.index(field)
        L.orc_set_field(self._h, int(field), int(obj), float(intensity), float(d_intensity))

    def emitter_weight(self, i):
        L = lib()
        L.orc_emitter_sampling_weight.restype = C.c_float
        L.orc_emitter_sampling_weight.argtypes = [C.c_void_p, C.c_int]
        return float(L.orc_emitter_sampling_weight(self._h, int(i)))

    def triangle_info(self, tangent=False):
        out = np.zeros((self.num_triangles, 25), dtype=np.float32)
        lib().orc_get_triangle_info(self._h, int(tangent), out.ctypes.data)
        return out

    def sec_edges(self, tangent=False):
        out = np.zeros((self.num_sec_edges, 16), dtype=np.float32)
        lib().orc_get_sec_edges(self._h, int(tangent), out.ctypes.data)
        return out

    def primary_edges(self, sensor=0, tangent=False):
        out = np.zeros((self.num_primary_edges(sensor), 7), dtype=np.float32)
        lib().orc_get_primary_edges(self._h, sensor, int(tangent), out.ctypes.data)
        return out

    def mesh_edges(self, mesh):
        n = lib().orc_get_mesh_edges(self._h, mesh, None, 0)
        out = np.zeros((n, 5), dtype=np.int32)
        lib().orc_get_mesh_edges(self._h, mesh, out.ctypes.data, n)
        return out

    def emitter_sampling_weight(self, e=0):
        return lib().orc_emitter_sampling_weight(self._h, e)

    def trace(self, o, d, use_bvh=False):
        o, d = _f32(o).reshape(-1, 3), _f32(d).reshape(-1, 3)
        n = o.shape[0]
        tri = np.zeros(n, dtype=np.int32)
        uv = np.zeros((n, 2), dtype=np.float32)
        t = np.zeros(n, dtype=np.float32)
        lib().orc_trace(self._h, n, o.ctypes.data, d.ctypes.data, int(use_bvh), tri.ctypes.data, uv.ctypes.data, t.ctypes.data)
        return tri, uv, t

    # ---- rendering
    def _npix(self, pix_ids):
        return len(pix_ids) if pix_ids is not None else self.spec.width * self.spec.height

    def render_c(self, sensor=0, max_depth=1, hide_emitters=False, seed=0, skip=0, pix_ids=None, lane_begin=0, lane_end=-1):
        pid = _i32(pix_ids) if pix_ids is not None else None
        n = self._npix(pix_ids)
        out = np.zeros((n, 3), dtype=np.float32)
        rc = lib().orc_render_c(self._h, sensor, max_depth, int(hide_emitters), _Sampler(seed, skip),
                                pid.ctypes.data if pid is not None else None, n, lane_begin, lane_end, out.ctypes.data)
        if rc:
            raise RuntimeError(lib().orc_last_error().decode())
        return out

    def render_d(self, sensor=0, max_depth=1, hide_emitters=False, seeds=(0, 0, 0), skips=(0, 0, 0), pix_ids=None,
                 guiding=None, terms=TERM_ALL, shard_rank=0, shard_count=1, shard_mode=0):
        pid = _i32(pix_ids) if pix_ids is not None else None
        n = self._npix(pix_ids)
        out = np.zeros((n, 3), dtype=np.float32)
        dout = np.zeros((n, 3), dtype=np.float32)
        sm = (_Sampler * 3)(*[_Sampler(int(seeds[i]), int(skips[i])) for i in range(3)])
        rc = lib().orc_render_d(self._h, sensor, max_depth, int(hide_emitters), sm,
                                pid.ctypes.data if pid is not None else None, n,
                                guiding._h if guiding is not None else None, terms, shard_rank, shard_count, shard_mode,
                                out.ctypes.data, dout.ctypes.data)
        if rc:
            raise RuntimeError(lib().orc_last_error().decode())
        return out, dout

    def li_lanes(self, lane_begin, lane_end, sensor=0, max_depth=1, hide_emitters=False, seed=0, skip=0):
        out = np.zeros((lane_end - lane_begin, 3), dtype=np.float32)
        lib().orc_li_lanes(self._h, sensor, max_depth, int(hide_emitters), _Sampler(seed, skip), lane_begin, lane_end, out.ctypes.data)
        return out

    def guiding_build(self, sensor, reso, nrounds=1, seed=0, max_depth=1):
        return OracleGuiding(self, sensor, reso, nrounds, seed, max_depth)


class OracleGuiding:
    def __init__(self, scene, sensor, reso, nrounds, seed, max_depth):
        r = (C.c_int * 4)(*reso)
        self._scene = scene
        self._h = lib().orc_guiding_build(scene._h, sensor, max_depth, r, nrounds, seed)
        if not self._h:
            raise RuntimeError(lib().orc_last_error().decode())

    def mass(self):
        n = lib().orc_guiding_num_cells(self._h)
        out = np.zeros(n, dtype=np.float32)
        lib().orc_guiding_get_mass(self._h, out.ctypes.data)
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_guiding_destroy(self._h)
                self._h = None
        except Exception:
            pass


# ------------------------------------------------------------------ KAT helpers
def tea64(v0, v1):
    return int(lib().orc_tea64(v0, v1))


def pcg32_raw(initstate, initseq, n):
    out = np.zeros(n, dtype=np.uint32)
    lib().orc_pcg32_raw(initstate, initseq, n, out.ctypes.data)
    return out


def sampler_floats(seed_value, lane, n, skip=0):
    out = np.zeros(n, dtype=np.float32)
    lib().orc_sampler_floats(seed_value, lane, skip, n, out.ctypes.data)
    return out


def square_to_cosine_hemisphere(uv):
    uv = _f32(uv).reshape(-1, 2)
    out = np.zeros((uv.shape[0], 3), dtype=np.float32)
    lib().orc_square_to_cosine_hemisphere(uv.shape[0], uv.ctypes.data, out.ctypes.data)
    return out


def square_to_uniform_triangle(uv):
    uv = _f32(uv).reshape(-1, 2)
    out = np.zeros((uv.shape[0], 2), dtype=np.float32)
    lib().orc_square_to_uniform_triangle(uv.shape[0], uv.ctypes.data, out.ctypes.data)
    return out


def coordinate_system(n):
    n = _f32(n)
    s, t = np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().orc_coordinate_system(n.ctypes.data, s.ctypes.data, t.ctypes.data)
    return s, t


def distrb_sample_reuse(pmf, sample):
    pmf = _f32(pmf)
    s = np.array([sample], dtype=np.float32)
    pdf = np.zeros(1, dtype=np.float32)
    idx = lib().orc_distrb_sample_reuse(len(pmf), pmf.ctypes.data, s.ctypes.data, pdf.ctypes.data)
    return idx, float(s[0]), float(pdf[0])
