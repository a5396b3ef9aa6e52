"""psdr_jit_amd — MI355X-native drop-in for the psdr_jit PathTracer hot path.

Keeps the reference's Python surface (reference src/psdr.cpp:120-439) for the in-scope subset:
    Scene, RenderOption, Mesh, DiffuseBSDF, AreaLight, PerspectiveCamera, PathTracer
    scene.opts.{width,height,spp,sppe,sppse,log_level}, scene.seed, scene.param_map[...],
    scene.add_Sensor / add_BSDF / add_Mesh / configure, PathTracer.renderC / renderD /
    preprocess_secondary_edges / hide_emitters.
drjit arrays are replaced by torch tensors (device memory, streams, autograd = plumbing):
    Matrix4fD / Matrix4fC / FloatD build tensors, images come back as float32 [n_pixels, 3]
    CUDA(ROCm) tensors, `forward_grad(img, P)` stands in for drjit.set_grad/forward_to/grad and
    `loss.backward()` for drjit.backward.

All rendering happens in libpsdr_hip.so (hand-written gfx950 kernels) behind the C ABI of
include/psdr_hip.h.  There is NO CPU fallback: importing this package fails if the native
libraries are missing, and rendering raises if no GPU is visible.
"""
import os as _os

import numpy as _np
import torch as _torch

from . import build as _build

_HERE = _os.path.dirname(_os.path.abspath(__file__))
if not (_os.path.exists(_build.HIP_LIB) and _os.path.exists(_build.CORE_LIB)):
    raise ImportError(
        "psdr_jit_amd: native libraries not built (expected %s and %s). Run `python -m psdr_jit_amd.build` "
        "or __graft_entry__.build(); there is no CPU fallback." % (_build.HIP_LIB, _build.CORE_LIB))

from . import _psdr_core as _core  # noqa: E402

Object = _core.Object
RenderOption = _core.RenderOption
BSDF = _core.BSDF
DiffuseBSDF = _core.DiffuseBSDF
Emitter = _core.Emitter
AreaLight = _core.AreaLight
MicrofacetBSDF = _core.MicrofacetBSDF
RoughConductorBSDF = _core.RoughConductorBSDF
RoughDielectricBSDF = _core.RoughDielectricBSDF
MicrofacetBSDFPerVertex = _core.MicrofacetBSDFPerVertex
NormalMapBSDF = _core.NormalMapBSDF
EnvironmentMap = _core.EnvironmentMap
Sensor = _core.Sensor
PerspectiveCamera = _core.PerspectiveCamera
Mesh = _core.Mesh
Scene = _core.Scene
Integrator = _core.Integrator


def OrthographicCamera(near, far):
    """OrthographicCamera(near, far) (reference psdr.cpp:375-383, src/sensor/orthographic.cpp): the view volume is
    [-1, 1] x [-1/aspect, 1/aspect] camera units wide; returns a camera object with .orthographic == True."""
    return _core._make_orthographic(float(near), float(far))


PathTracer = _core.PathTracer
Direct = _core.Direct          # DirectIntegrator(mis), reference psdr.cpp:436-439
PsdrException = _core.PsdrException

TERM_INTERIOR, TERM_PRIMARY, TERM_SECONDARY, TERM_ALL = 1, 2, 4, 7


# ------------------------------------------------------------------ drjit type stand-ins
def FloatD(x=0.0):
    """drjit.cuda.ad.Float stand-in: a float32 scalar tensor (set requires_grad_() to differentiate)."""
    return x.to(_torch.float32) if isinstance(x, _torch.Tensor) else _torch.tensor(float(x), dtype=_torch.float32)


FloatC = FloatD


def Matrix4fD(rows):
    """drjit Matrix4f stand-in: 4x4 float32 tensor; entries may be tensors (keeps the autograd graph)."""
    if isinstance(rows, _torch.Tensor):
        return rows.to(_torch.float32).reshape(4, 4)
    flat = [e for r in rows for e in r]
    if any(isinstance(e, _torch.Tensor) for e in flat):
        ref = next(e for e in flat if isinstance(e, _torch.Tensor))
        flat = [e.to(_torch.float32).reshape(()) if isinstance(e, _torch.Tensor) else _torch.tensor(float(e), dtype=_torch.float32, device=ref.device) for e in flat]
        return _torch.stack(flat).reshape(4, 4)
    return _torch.tensor(_np.asarray(rows, dtype=_np.float32).reshape(4, 4))


Matrix4fC = Matrix4fD


def _split(x, shape):
    """tensor/array -> (float32 numpy value, tensor-or-None kept for autograd)."""
    if isinstance(x, _torch.Tensor):
        return x.detach().to("cpu", _torch.float32).numpy().reshape(shape), x
    return _np.asarray(x, dtype=_np.float32).reshape(shape), None


def _zeros_like(v):
    return _np.zeros_like(v, dtype=_np.float32)


def _params(obj):
    d = obj.__dict__.get("_psdr_params")
    if d is None:
        d = {}
        obj.__dict__["_psdr_params"] = d
    return d


def _make_param_property(name, shape_of):
    def getter(self):
        t = _params(self).get(name)
        if t is not None:
            return t
        return _torch.from_numpy(_np.array(self._get(name, False)))

    def setter(self, value):
        v, t = _split(value, shape_of(self, value))
        self._set(name, v, _zeros_like(v))
        if t is not None:
            _params(self)[name] = t
        else:
            _params(self).pop(name, None)

    return property(getter, setter)


def _m44(self, value):
    return (4, 4)


def _v3(self, value):
    return (3,) if _np.size(value.detach().cpu().numpy() if isinstance(value, _torch.Tensor) else value) == 3 else (1,)


def _refl_shape(self, value):
    """a colour (1 or 3 numbers) or a reflectance texture [H, W, 3] (the reference's Bitmap3fD)"""
    shp = tuple(value.shape) if hasattr(value, "shape") else _np.shape(value)
    if len(shp) == 3:
        return shp
    return _v3(self, value)


class Bitmap3fD:
    """Stand-in for the reference's Bitmap3fD (bitmap.h): an [H, W, 3] array.  Bitmap3fD(), Bitmap3fD(file_name),
    Bitmap3fD(value) or Bitmap3fD(width, height, data[H*W, 3])."""

    def __init__(self, *args):
        if len(args) == 0:
            self.data = _np.zeros((1, 1, 3), _np.float32)
        elif len(args) == 1 and isinstance(args[0], (str, bytes, _os.PathLike)):
            self.data = None
            self.load_openexr(args[0])
        elif len(args) == 1:
            a = args[0].detach().cpu().numpy() if isinstance(args[0], _torch.Tensor) else _np.asarray(args[0], _np.float32)
            self.data = a.reshape(1, 1, 3) if a.size == 3 else _np.ascontiguousarray(a, _np.float32)
        else:
            w, h, d = args
            d = d.detach().cpu().numpy() if isinstance(d, _torch.Tensor) else _np.asarray(d, _np.float32)
            self.data = _np.ascontiguousarray(d, _np.float32).reshape(int(h), int(w), 3)

    def load_openexr(self, file_name):
        self.data = load_radiance_image(file_name)

    @property
    def resolution(self):
        return (self.data.shape[1], self.data.shape[0])

    # m_rot, m_scale, m_trans (reference bitmap.h:37-39; bound as rotate / scale / translate, psdr.cpp:217-219)
    rotate, scale, translate = 0.0, 1.0, (0.0, 0.0)


def _bitmap_eval(self, uv, flip_v=True, envmap_mode=False):
    """Bitmap::eval(uv[N, 2]) (reference bitmap.cpp:47-128) -> [N, channels] numpy array"""
    from .host_utils import bitmap_eval
    uv = uv.detach().cpu().numpy() if isinstance(uv, _torch.Tensor) else uv
    xf = [float(_np.asarray(q.detach().cpu() if isinstance(q, _torch.Tensor) else q).reshape(-1)[k]) for q, k in
          ((self.rotate, 0), (self.scale, 0), (self.translate, 0), (self.translate, 1))]
    return bitmap_eval(self.data, uv, flip_v, envmap_mode, xf)




Bitmap3fD.eval = _bitmap_eval


class Bitmap1fD:
    """Stand-in for the reference's Bitmap1fD: an [H, W] array (Bitmap1fD(), Bitmap1fD(value), Bitmap1fD(width, height, data))."""

    def __init__(self, *args):
        if len(args) == 0:
            self.data = _np.zeros((1, 1), _np.float32)
        elif len(args) == 1:
            a = args[0].detach().cpu().numpy() if isinstance(args[0], _torch.Tensor) else _np.asarray(args[0], _np.float32)
            self.data = a.reshape(1, 1) if a.size == 1 else _np.ascontiguousarray(a, _np.float32)
        else:
            w, h, d = args
            d = d.detach().cpu().numpy() if isinstance(d, _torch.Tensor) else _np.asarray(d, _np.float32)
            self.data = _np.ascontiguousarray(d, _np.float32).reshape(int(h), int(w))

    @property
    def resolution(self):
        return (self.data.shape[1], self.data.shape[0])

    rotate, scale, translate = 0.0, 1.0, (0.0, 0.0)       # psdr.cpp:204-206


Bitmap1fD.eval = _bitmap_eval
from .host_utils import Sampler, DiscreteDistribution  # noqa: E402,F401


def _const_of(x, n):
    """a constant parameter given as a number/vector or as a 1x1 Bitmap (larger bitmaps: DiffuseBSDF and MicrofacetBSDF only)"""
    if isinstance(x, (Bitmap3fD, Bitmap1fD)):
        if x.data.size != n:
            raise RuntimeError("bitmap parameters larger than 1x1 are only built for DiffuseBSDF and MicrofacetBSDF")
        return x.data.reshape(n)
    return x


def _vtx(self, value):
    return (self.num_vertices, 3)


for _cls in (Mesh, Sensor, PerspectiveCamera):
    for _n in ("to_world", "to_world_left", "to_world_right"):
        setattr(_cls, _n, _make_param_property(_n, _m44))
Mesh.vertex_positions = _make_param_property("vertex_positions", _vtx)
DiffuseBSDF.reflectance = _make_param_property("reflectance", _refl_shape)
def _rough_shape(self, value):
    """a number or a roughness map [H, W] (the reference's Bitmap1fD)"""
    shp = tuple(value.shape) if hasattr(value, "shape") else _np.shape(value)
    return shp if len(shp) == 2 and shp[0] > 1 else (1,)


MicrofacetBSDF.specularReflectance = _make_param_property("specularReflectance", _refl_shape)
MicrofacetBSDF.diffuseReflectance = _make_param_property("diffuseReflectance", _refl_shape)
MicrofacetBSDF.roughness = _make_param_property("roughness", _rough_shape)
# alpha / eta / k of RoughConductor and alpha of RoughDielectric may be bitmaps above 1x1 (an alpha map serves both axes)
for _n in ("alpha_u", "alpha_v"):
    setattr(RoughConductorBSDF, _n, _make_param_property(_n, _rough_shape))
    setattr(RoughDielectricBSDF, _n, _make_param_property(_n, _rough_shape))
for _n in ("eta", "k"):
    setattr(RoughConductorBSDF, _n, _make_param_property(_n, _refl_shape))
RoughConductorBSDF.specular_reflectance = _make_param_property("specular_reflectance", _v3)
RoughDielectricBSDF.eta = _make_param_property("eta", lambda self, value: (1,))
for _n in ("specularReflectance", "diffuseReflectance"):
    setattr(MicrofacetBSDFPerVertex, _n, _make_param_property(_n, lambda self, value: (-1, 3)))
MicrofacetBSDFPerVertex.roughness = _make_param_property("roughness", lambda self, value: (-1,))
NormalMapBSDF.normal_map = _make_param_property("normal_map", _refl_shape)
AreaLight.radiance = _make_param_property("radiance", _v3)


# ------------------------------------------------------------------ uv transform of a bitmap parameter
# The reference's Bitmap1fD / Bitmap3fD carry three differentiable members besides their texels - translate (Vector2fD), rotate and
# scale (FloatD), psdr.cpp:204-206, 217-219, applied to every lookup by bitmap.cpp:64-86.  Here a bitmap parameter of a BSDF (or the
# environment map's radiance) is a tensor property, so its transform is reached through obj.uv_transform(name):
#     t = scene.param_map["BSDF[0]"].uv_transform("reflectance");  t.rotate = torch.tensor(0.3, requires_grad=True)
# (the reference spelling: scene.param_map["BSDF[0]"].reflectance.rotate = FloatD(0.3)).  The three members travel to the host
# object as one leaf [rotate, scale, translate.x, translate.y] named "<parameter>@uv".
_UV_SLOT = {"reflectance": 0, "diffuseReflectance": 0, "specularReflectance": 1, "roughness": 2, "normal_map": 0,
            "eta": 0, "k": 1, "alpha_u": 2, "alpha_v": 2}
_UV_SUFFIX = "@uv"


def _uv_push(obj, name, v, d):
    """[rotate, scale, tx, ty] (+ tangent) of bitmap parameter `name` -> host object"""
    v = _np.ascontiguousarray(_np.asarray(v, dtype=_np.float32).reshape(4))
    d = _np.ascontiguousarray(_np.asarray(d, dtype=_np.float32).reshape(4))
    if isinstance(obj, EnvironmentMap):
        obj._set("radiance_uv_xf", v, d)
    else:
        obj._set_uv_xf(_UV_SLOT[name], v, d)


class BitmapTransform:
    """rotate / scale / translate of one bitmap parameter (reference Bitmap::m_rot / m_scale / m_trans)"""

    def __init__(self, obj, name):
        if isinstance(obj, EnvironmentMap):
            if name != "radiance":
                raise RuntimeError("EnvironmentMap: the bitmap parameter is 'radiance'")
        elif name not in _UV_SLOT or not hasattr(type(obj), name):
            raise RuntimeError("%s has no bitmap parameter %r" % (type(obj).__name__, name))
        if name == "alpha_v":
            # ONE roughness map serves both axes here (the kernels look alpha up once, slot 2: shade.h "alpha (both axes)"), so the two names
            # are one transform: uv_transform("alpha_v") IS uv_transform("alpha_u") - reading either shows what was set through the other.
            # (The reference keeps m_rot / m_scale / m_trans per Bitmap, bitmap.h:37-39; two different alpha maps are not representable here.)
            self.__dict__["_asked"] = "alpha_v"
            name = "alpha_u"
        self.__dict__.setdefault("_asked", name)
        self.__dict__["_obj"], self.__dict__["_name"] = obj, name
        parts = obj.__dict__.setdefault("_psdr_uv_parts", {})
        if name not in parts:
            cur = self._host()
            parts[name] = {"rotate": float(cur[0]), "scale": float(cur[1]), "translate": (float(cur[2]), float(cur[3]))}

    def _host(self):
        o = self._obj
        return _np.asarray(o._get("radiance_uv_xf", False) if isinstance(o, EnvironmentMap) else o._get_uv_xf(_UV_SLOT[self._name], False))

    def __getattr__(self, key):
        if key in ("rotate", "scale", "translate"):
            return self._obj.__dict__["_psdr_uv_parts"][self._name][key]
        raise AttributeError(key)

    def __setattr__(self, key, value):
        if key not in ("rotate", "scale", "translate"):
            raise AttributeError("a bitmap transform has rotate, scale and translate")
        parts = self._obj.__dict__["_psdr_uv_parts"][self._name]
        if self._name == "alpha_u":
            # the aliased pair: a value set through one name and then changed through the other is the case the reference would keep apart
            by = parts.setdefault("_set_by", {})
            differs = not _np.array_equal(_np.asarray(_torch.as_tensor(parts[key]).detach().cpu()), _np.asarray(_torch.as_tensor(value).detach().cpu()))
            if by.get(key, self._asked) != self._asked and differs:
                import warnings
                warnings.warn("uv_transform(%r).%s overrides the value set through uv_transform(%r): alpha_u and alpha_v share ONE roughness map and ONE "
                              "transform here (the reference keeps a Bitmap per axis)" % (self._asked, key, by[key]), RuntimeWarning, stacklevel=2)
            if differs or key not in by:
                by[key] = self._asked
        parts[key] = value
        seq = (parts["rotate"], parts["scale"], parts["translate"])
        if any(isinstance(q, _torch.Tensor) for q in seq):
            ref = next(q for q in seq if isinstance(q, _torch.Tensor))
            flat = [(q.to(ref.device, _torch.float32) if isinstance(q, _torch.Tensor) else _torch.as_tensor(q, dtype=_torch.float32, device=ref.device)).reshape(-1) for q in seq]
            t = _torch.cat(flat)
            if t.numel() != 4:
                raise RuntimeError("bitmap transform: rotate and scale are scalars, translate has two components")
            _params(self._obj)[self._name + _UV_SUFFIX] = t
            v = t.detach().cpu().numpy()
        else:
            _params(self._obj).pop(self._name + _UV_SUFFIX, None)
            v = _np.concatenate([_np.asarray(q, dtype=_np.float32).reshape(-1) for q in seq])
        _uv_push(self._obj, self._name, v, _np.zeros(4, _np.float32))


def _uv_transform(self, name="radiance"):
    return BitmapTransform(self, name)


for _cls in (DiffuseBSDF, MicrofacetBSDF, RoughConductorBSDF, RoughDielectricBSDF, NormalMapBSDF, EnvironmentMap):
    _cls.uv_transform = _uv_transform


def _copy_uv_parts(src, dst):
    """the Python half of a bitmap transform follows the host object's copy (the host copy carries the values)"""
    if src is not None and "_psdr_uv_parts" in src.__dict__:
        dst.__dict__["_psdr_uv_parts"] = {k: dict(v) for k, v in src.__dict__["_psdr_uv_parts"].items()}


def _adopt_bitmap_transform(obj, name, bitmap):
    """a Bitmap1fD / Bitmap3fD handed to a constructor brings its rotate / scale / translate along"""
    if isinstance(bitmap, (Bitmap3fD, Bitmap1fD)) and bitmap.data.size > 3:
        for key in ("rotate", "scale", "translate"):
            val = getattr(bitmap, key)
            if isinstance(val, _torch.Tensor) or val != getattr(type(bitmap), key):
                setattr(obj.uv_transform(name), key, val)


def _set_transform(self, mat, set_left=True):
    """Mesh/Sensor.set_transform (reference mesh.h:26-33, sensor.h:34-40)."""
    name = "to_world_left" if set_left else "to_world_right"
    setattr(self, name, mat)


def _append_transform(self, mat, append_left=True):
    name = "to_world_left" if append_left else "to_world_right"
    cur = getattr(self, name)
    m = Matrix4fD(mat)
    cur = cur.to(m.device) if isinstance(cur, _torch.Tensor) else cur
    setattr(self, name, (m @ cur) if append_left else (cur @ m))


for _cls in (Mesh, Sensor, PerspectiveCamera):
    _cls.set_transform = _set_transform
    _cls.append_transform = _append_transform

_DiffuseBSDF_init = DiffuseBSDF.__init__


def _diffuse_init(self, reflectance=None):
    if reflectance is None:
        _DiffuseBSDF_init(self)
        return
    src = reflectance
    if isinstance(reflectance, Bitmap3fD):
        reflectance = reflectance.data.reshape(3) if reflectance.data.size == 3 else reflectance.data
    shp = tuple(reflectance.shape) if hasattr(reflectance, "shape") else _np.shape(reflectance)
    if len(shp) == 3:                       # a reflectance texture
        _DiffuseBSDF_init(self)
        self.reflectance = reflectance
        _adopt_bitmap_transform(self, "reflectance", src)
        return
    v, t = _split(reflectance, (-1,))
    _DiffuseBSDF_init(self, v)
    if t is not None:
        _params(self)["reflectance"] = t


DiffuseBSDF.__init__ = _diffuse_init
_MicrofacetBSDF_init = MicrofacetBSDF.__init__


def _microfacet_init(self, specular=None, diffuse=None, roughness=None):
    """MicrofacetBSDF() or MicrofacetBSDF(specularReflectance, diffuseReflectance, roughness) (reference psdr.cpp:298-304); each
    parameter a constant, a Bitmap3fD / Bitmap1fD, or an array [H, W, 3] / [H, W] (a bitmap with a resolution above 1x1)"""
    _MicrofacetBSDF_init(self)
    if specular is None:
        return
    for name, val, n in (("specularReflectance", specular, 3), ("diffuseReflectance", diffuse, 3), ("roughness", roughness, 1)):
        _adopt_bitmap_transform(self, name, val)
        if isinstance(val, (Bitmap3fD, Bitmap1fD)):
            val = val.data.reshape(n) if val.data.size == n else val.data
        shp = tuple(val.shape) if hasattr(val, "shape") else _np.shape(val)
        if len(shp) < 2:
            v = _split(val, (-1,))[0]
            if n == 3 and v.size == 1 and not isinstance(val, _torch.Tensor):
                val = v * _np.ones(3, _np.float32)
        setattr(self, name, val)


MicrofacetBSDF.__init__ = _microfacet_init
_RoughConductorBSDF_init = RoughConductorBSDF.__init__


def _roughconductor_init(self, *args):
    """RoughConductorBSDF(), (alpha, eta, k[, specular_reflectance]) or (alpha_u, alpha_v, eta, k[, specular_reflectance])
    (reference roughconductor.h:10-26; constants instead of bitmaps)"""
    _RoughConductorBSDF_init(self)
    if not args:
        return
    # 1x1 Bitmap1fD / Bitmap3fD arguments (tutorials/batch_render.ipynb) are the constants they hold
    orig = args
    args = tuple((_const_of(a, a.data.size) if a.data.size in (1, 3) else a.data) if isinstance(a, (Bitmap3fD, Bitmap1fD)) else a for a in args)
    scal = lambda x: float(_np.ravel(_split(x, (-1,))[0])[0])
    # the reference tells its two overloads apart by Bitmap1fD / Bitmap3fD types; plain numbers cannot: five arguments or a
    # Bitmap1fD in second place select (alpha_u, alpha_v, eta, k[, specular]), anything else is (alpha, eta, k[, specular]),
    # and a four-argument call whose second argument is a bare scalar could be either and is refused
    if len(args) == 5 or (len(args) == 4 and isinstance(orig[1], Bitmap1fD)):
        anisotropic = True
    elif len(args) == 4 and not isinstance(orig[1], Bitmap3fD) and _np.size(_split(args[1], (-1,))[0]) == 1:
        raise ValueError("RoughConductorBSDF: (a, b, c, d) with a scalar b is ambiguous between (alpha, eta, k, specular_reflectance) and "
                         "(alpha_u, alpha_v, eta, k): pass Bitmap1fD / Bitmap3fD arguments as the reference does, or all five parameters")
    else:
        anisotropic = False
    if len(args) < 3:
        raise ValueError("RoughConductorBSDF takes (alpha, eta, k[, specular_reflectance]) or (alpha_u, alpha_v, eta, k[, specular_reflectance])")
    if anisotropic:
        au, av, rest = args[0], args[1], args[2:]
    else:
        au, av, rest = args[0], args[0], args[1:]
    keep = lambda x: (x if x.dim() == 2 else x.reshape(1)) if isinstance(x, _torch.Tensor) else (x if _np.ndim(x) == 2 else [scal(x)])      # a tensor stays a leaf, a map a map
    self.alpha_u, self.alpha_v = keep(au), keep(av)
    self.eta, self.k = rest[0], rest[1]
    if len(rest) > 2:
        self.specular_reflectance = rest[2]


RoughConductorBSDF.__init__ = _roughconductor_init
_MicrofacetBSDFPerVertex_init = MicrofacetBSDFPerVertex.__init__


def _microfacet_pv_init(self, specular, diffuse, roughness):
    """MicrofacetBSDFPerVertex(specularReflectance[n, 3], diffuseReflectance[n, 3], roughness[n]) (reference psdr.cpp:306-310,
    microfacet_pv.h): one value per vertex of the mesh the BSDF is used on"""
    _MicrofacetBSDFPerVertex_init(self)
    self.specularReflectance, self.diffuseReflectance, self.roughness = specular, diffuse, roughness


MicrofacetBSDFPerVertex.__init__ = _microfacet_pv_init
_NormalMapBSDF_init = NormalMapBSDF.__init__


def _normalmap_init(self, normal_map=None):
    """NormalMapBSDF() or NormalMapBSDF(vec3 | Bitmap3fD | array[H, W, 3]) (reference psdr.cpp:273-277, normalmap.h); the BSDF it
    perturbs is assigned through `nested_bsdf`"""
    _NormalMapBSDF_init(self)
    if normal_map is not None:
        src = normal_map
        if isinstance(normal_map, Bitmap3fD):
            normal_map = normal_map.data.reshape(3) if normal_map.data.size == 3 else normal_map.data
        self.normal_map = normal_map
        _adopt_bitmap_transform(self, "normal_map", src)


def _nm_get_nested(self):
    n = self._nested
    if n is not None:
        src = self.__dict__.get("_psdr_nested_src")
        if src is not None and "_psdr_params" in src.__dict__ and "_psdr_params" not in n.__dict__:
            n.__dict__["_psdr_params"] = dict(src.__dict__["_psdr_params"])
            _copy_uv_parts(src, n)
    return n


def _nm_set_nested(self, bsdf):
    self._set_nested(bsdf)                       # the map keeps its own copy, as the scene does with everything it is given
    self.__dict__["_psdr_nested_src"] = bsdf


NormalMapBSDF.__init__ = _normalmap_init
NormalMapBSDF.nested_bsdf = property(_nm_get_nested, _nm_set_nested)
_RoughDielectricBSDF_init = RoughDielectricBSDF.__init__


def _roughdielectric_init(self, *args):
    """RoughDielectricBSDF(), (intIOR, extIOR) or (alpha, intIOR, extIOR) (reference roughdielectric.h:10-27; alpha a constant
    or a 1x1 Bitmap1fD).  The reference binds the class without a constructor (psdr.cpp:295) and only builds it from XML."""
    if len(args) == 3:
        _RoughDielectricBSDF_init(self, float(args[1]), float(args[2]))
        al = args[0]
        if isinstance(al, Bitmap1fD):
            al = al.data.reshape(1) if al.data.size == 1 else al.data
        if _np.ndim(al.detach().cpu().numpy() if isinstance(al, _torch.Tensor) else al) == 2:
            self.alpha_u = al                                   # an alpha map (serves both axes)
        else:
            a = float(_np.ravel(_split(al, (-1,))[0])[0])
            self.alpha_u, self.alpha_v = [a], [a]
    elif len(args) == 2:
        _RoughDielectricBSDF_init(self, float(args[0]), float(args[1]))
    else:
        _RoughDielectricBSDF_init(self)


RoughDielectricBSDF.__init__ = _roughdielectric_init
_AreaLight_init = AreaLight.__init__


def _arealight_init(self, radiance):
    v, t = _split(radiance, (-1,))
    _AreaLight_init(self, v)
    if t is not None:
        _params(self)["radiance"] = t


AreaLight.__init__ = _arealight_init


# ------------------------------------------------------------------ Scene
def _keep(scene, key, src):
    """The scene works on its own copies (reference scene.cpp:107-126,148-309): carry the torch leaves over."""
    obj = scene.param_map[key]
    scene.__dict__.setdefault("_psdr_objs", {})[key] = obj
    if src is not None and "_psdr_params" in src.__dict__:
        obj.__dict__["_psdr_params"] = dict(src.__dict__["_psdr_params"])
    _copy_uv_parts(src, obj)
    return obj


_Scene_add_Sensor = Scene.add_Sensor
_Scene_add_BSDF = Scene.add_BSDF


def _add_Sensor(self, sensor):
    _Scene_add_Sensor(self, sensor)
    _keep(self, "Sensor[%d]" % (self.num_sensors - 1), sensor)


def _add_BSDF(self, bsdf, name, twoSide=False):
    _Scene_add_BSDF(self, bsdf, name, twoSide)
    live = _keep(self, "BSDF[id=%s]" % name, bsdf)
    if isinstance(bsdf, NormalMapBSDF):          # the nested BSDF's leaves travel with it
        src = bsdf.__dict__.get("_psdr_nested_src")
        nested = live._nested
        if nested is not None:
            self.__dict__.setdefault("_psdr_objs", {})["BSDF[id=%s].nested" % name] = nested
            if src is not None and "_psdr_params" in src.__dict__:
                nested.__dict__["_psdr_params"] = dict(src.__dict__["_psdr_params"])
            _copy_uv_parts(src, nested)


def _add_normalmap_BSDF(self, bsdf1, bsdf2, name, twoSide=False):
    """Scene.add_normalmap_BSDF(NormalMapBSDF, MicrofacetBSDF, name, twoSide) (reference scene.cpp:128-145)"""
    nm = NormalMapBSDF(bsdf1.normal_map)
    if "_psdr_params" in bsdf1.__dict__:
        nm.__dict__["_psdr_params"] = dict(bsdf1.__dict__["_psdr_params"])
    _copy_uv_parts(bsdf1, nm)
    nm.nested_bsdf = bsdf2
    _add_BSDF(self, nm, name, twoSide)


def load_radiance_image(path):
    """Lat-long radiance image [H, W, 3] float32 from a .npy file or an OpenEXR file (the reference reads EXR through
    tinyexr, bitmap_loader.cpp; here: psdr_jit_amd.exr, a small reader for scanline float/half images)."""
    path = _os.fspath(path)
    if path.lower().endswith(".npy"):
        img = _np.load(path)
    else:
        from . import exr as _exr
        img = _exr.read_rgb(path)
    img = _np.ascontiguousarray(_np.asarray(img, dtype=_np.float32))
    if img.ndim != 3 or img.shape[2] < 3:
        raise RuntimeError("EnvironmentMap: expected an [H, W, 3] image")
    return img[:, :, :3].copy()


_EnvironmentMap_init = EnvironmentMap.__init__


def _envmap_init(self, radiance=None):
    """EnvironmentMap(file_name) as in the reference (envmap.h:13-15), or EnvironmentMap(array[H, W, 3])."""
    if radiance is None:
        _EnvironmentMap_init(self)
    elif isinstance(radiance, (str, bytes, _os.PathLike)):
        _EnvironmentMap_init(self, load_radiance_image(radiance))
    else:
        r = radiance.detach().cpu().numpy() if isinstance(radiance, _torch.Tensor) else radiance
        _EnvironmentMap_init(self, _np.ascontiguousarray(_np.asarray(r, dtype=_np.float32)))
        if isinstance(radiance, _torch.Tensor):
            _params(self)["radiance"] = radiance


EnvironmentMap.__init__ = _envmap_init
# m_radiance (texels), m_scale and m_to_world_left are differentiable members (envmap.h:40-45): tensors assigned here are leaves
EnvironmentMap.radiance = _make_param_property("radiance", lambda self, value: tuple(value.shape) if hasattr(value, "shape") else _np.shape(value))


def _env_scale_get(self):
    t = _params(self).get("scale")
    return t if t is not None else float(self._get("scale", False)[0])


def _env_scale_set(self, value):
    v, t = _split(value, (1,))
    self._set("scale", v, _zeros_like(v))
    if t is not None:
        _params(self)["scale"] = t
    else:
        _params(self).pop("scale", None)


EnvironmentMap.scale = property(_env_scale_get, _env_scale_set)
EnvironmentMap.to_world = property(lambda self: self._get("to_world_raw", False), lambda self, v: self._set("to_world_raw", _np.ascontiguousarray(_np.asarray(_split(v, (4, 4))[0], dtype=_np.float32)), _np.zeros((4, 4), _np.float32)))
EnvironmentMap.to_world_left = _make_param_property("to_world_left", _m44)
EnvironmentMap.set_transform = lambda self, mat: setattr(self, "to_world_left", mat)


def _add_EnvironmentMap(self, emitter_or_path, to_world=None, scale=1.0):
    """add_EnvironmentMap(fname, to_world, scale) or add_EnvironmentMap(emitter), reference scene.cpp:85-105."""
    n_em = self.get_num_emitters()
    if isinstance(emitter_or_path, EnvironmentMap):
        e = emitter_or_path
    else:
        e = EnvironmentMap(emitter_or_path)
        e.scale = scale
        if to_world is not None:
            e.to_world = to_world
    self._add_EnvironmentMap(e)
    _keep(self, "Emitter[%d]" % n_em, e)


def _add_Mesh(self, mesh_or_path, *args):
    """add_Mesh(path, Matrix4, bsdf_id, emitter|None) or add_Mesh(mesh, bsdf_id, emitter=None)."""
    n_em = self.get_num_emitters()
    if isinstance(mesh_or_path, (str, bytes, _os.PathLike)):
        transform, bsdf_id, emitter = args[0], args[1], (args[2] if len(args) > 2 else None)
        v, t = _split(transform, (4, 4))
        self._add_Mesh_file(_os.fspath(mesh_or_path), v, bsdf_id, emitter)
        obj = _keep(self, "Mesh[%d]" % (self.num_meshes - 1), None)
        if t is not None:
            _params(obj)["to_world"] = t
    else:
        bsdf_id, emitter = args[0], (args[1] if len(args) > 1 else None)
        self._add_Mesh_obj(mesh_or_path, bsdf_id, emitter)
        _keep(self, "Mesh[%d]" % (self.num_meshes - 1), mesh_or_path)
    if emitter is not None:
        _keep(self, "Emitter[%d]" % n_em, emitter)


def _configure(self, active_sensor=()):
    _sync_params(self)
    self._configure(list(active_sensor))
    self.__dict__["_psdr_active"] = list(active_sensor)


Scene.add_Sensor = _add_Sensor
Scene.add_BSDF = _add_BSDF
Scene.add_normalmap_BSDF = _add_normalmap_BSDF
Scene.add_Mesh = _add_Mesh


class RayC:
    """Stand-in for the reference's RayC / RayD (ray.h): origins and directions as [N, 3] tensors."""

    def __init__(self, o=None, d=None):
        self.o, self.d = o, d

    def reversed(self):
        return RayC(self.o, -self.d)


RayD = RayC


class FrameC:
    def __init__(self, s, t, n):
        self.s, self.t, self.n = s, t, n


class IntersectionC:
    """What Scene.unit_ray_intersect returns (reference intersection.h:24-60): wi, p, t, n, sh_frame, uv, J, shape as tensors
    over the rays; `shape` holds mesh indices (-1 = miss) where the reference holds mesh pointers."""

    def __init__(self, rec):
        self._valid = rec[:, 0] > 0
        self.shape = rec[:, 1].to(_torch.int32)
        self.t, self.J = rec[:, 2], rec[:, 3]
        self.p, self.n = rec[:, 4:7], rec[:, 7:10]
        self.sh_frame = FrameC(rec[:, 10:13], rec[:, 13:16], rec[:, 16:19])
        self.wi, self.uv = rec[:, 19:22], rec[:, 22:24]

    def is_valid(self):
        return self._valid


IntersectionD = IntersectionC


class InteractionC:
    """reference interaction.h / psdr.cpp:223-233: the base of IntersectionC - wi, p, t and is_valid()"""

    def __init__(self, wi, p, t, valid):
        self.wi, self.p, self.t, self._valid = wi, p, t, valid

    def is_valid(self):
        return self._valid


InteractionD = InteractionC


class SampleRecordC:
    """reference records.h / psdr.cpp:251-257: pdf, is_valid"""

    def __init__(self, pdf, is_valid):
        self.pdf, self.is_valid = pdf, is_valid


class PositionSampleC(SampleRecordC):
    """reference records.h / psdr.cpp:259-265: p, J (+ pdf, is_valid); what Mesh.sample_position returns.  C: numpy arrays, D: torch tensors"""

    def __init__(self, p, J, pdf, is_valid, n=None):
        SampleRecordC.__init__(self, pdf, is_valid)
        self.p, self.J, self.n = p, J, n


SampleRecordD, PositionSampleD = SampleRecordC, PositionSampleC


def _mesh_sample_position(self, sample2, active=True):
    """Mesh.sample_position(sample2, active) (reference psdr.cpp:321-322, mesh.cpp:403-454): a face by area (DiscreteDistribution::sample_reuse on
    sample2.x), a point on it by warp::square_to_uniform_triangle; pdf = 1 / total area.  A numpy sample2 gives the C instantiation (numpy float32
    arrays, J = 1), a torch tensor the D one: torch float64 tensors whose graph reaches sample2 and the mesh's leaves (vertex_positions, to_world*),
    J = area / detach(area).  Host-side numpy / torch - scripts call it on a handful of samples; the kernels sample emitters themselves (shade.h)."""
    from . import chain as _chain
    if self.num_faces <= 0 or _np.asarray(self.vertex_positions_T).size == 0:
        raise PsdrException("Mesh::sample_position: the mesh is not configured")
    is_d = isinstance(sample2, _torch.Tensor)
    F = _torch.as_tensor(_np.asarray(self.face_indices, dtype=_np.int64))
    if is_d:
        def leaf(name):
            t = _params(self).get(name)
            return t.to(_torch.float64) if t is not None else _chain._t(self._get(name, False))
        M = leaf("to_world_left").reshape(4, 4) @ leaf("to_world").reshape(4, 4) @ leaf("to_world_right").reshape(4, 4)
        Vw = _chain._xform_pos(M, leaf("vertex_positions").reshape(-1, 3))
        s2 = sample2.to(_torch.float64).reshape(-1, 2)
    else:
        Vw = _chain._t(self.vertex_positions_T).reshape(-1, 3)
        s2 = _torch.as_tensor(_np.asarray(sample2, dtype=_np.float32).reshape(-1, 2)).to(_torch.float64)
    rows = _chain._process_mesh(Vw, F)
    area = rows[:, 21]
    dist = DiscreteDistribution()
    dist.init(area.detach().cpu().numpy().astype(_np.float32))
    sx = s2[:, 0].detach().cpu().numpy().astype(_np.float32)
    if dist.m_size == 1:
        idx, reused = _np.zeros(sx.size, _np.int64), s2[:, 0]
    else:
        # sample_reuse (pmf.cpp:26-45): the index, and the sample stretched back over its bucket
        t = sx * _np.float32(dist.m_sum)
        idx = _np.searchsorted(dist.m_cmf[:dist.m_size - 1], t, side="left").astype(_np.int64)
        lo = _np.where(idx > 0, dist.m_cmf[_np.maximum(idx - 1, 0)], _np.float32(0.0)).astype(_np.float32)
        reused = (s2[:, 0] * float(dist.m_sum) - _torch.as_tensor(lo.astype(_np.float64))) / _torch.as_tensor(dist.m_pmf[idx].astype(_np.float64))
    it = _torch.as_tensor(idx)
    tt = _torch.sqrt(_torch.clamp(1.0 - reused, min=0.0))                      # warp.h:79-82
    u, v = 1.0 - tt, tt * s2[:, 1]
    p = rows[it, 0:3] + rows[it, 3:6] * u[:, None] + rows[it, 6:9] * v[:, None]
    a = area[it]
    ok = _torch.as_tensor(_np.broadcast_to(_np.asarray(active, dtype=bool), (s2.shape[0],)).copy())
    inv_total = 1.0 / float(area.detach().sum())
    if is_d:
        return PositionSampleD(p, a / a.detach(), _torch.full((s2.shape[0],), inv_total, dtype=_torch.float64), ok, n=rows[it, 18:21])
    f = lambda x: x.detach().cpu().numpy().astype(_np.float32)
    return PositionSampleC(f(p), _np.ones(s2.shape[0], _np.float32), _np.full(s2.shape[0], inv_total, _np.float32), ok.numpy(), n=f(rows[it, 18:21]))


Mesh.sample_position = _mesh_sample_position


def _get_valid_edge_indices(self):
    # Mesh::m_valid_edge_indices (mesh.h:110, bound read-write at psdr.cpp:335): a member the reference declares and never fills or reads -
    # kept so that scripts touching it keep running; empty [0, 2] until a script stores something
    return self.__dict__.get("_psdr_valid_edge_indices", _np.zeros((0, 2), _np.int32))


def _set_valid_edge_indices(self, value):
    self.__dict__["_psdr_valid_edge_indices"] = _np.asarray(value, dtype=_np.int32).reshape(-1, 2)


Mesh.valid_edge_indices = property(_get_valid_edge_indices, _set_valid_edge_indices)


def _unit_ray_intersect(self, ray, active=None):
    """Scene.unit_ray_intersect (reference psdr.cpp:404, scene.cpp:809-...): closest hits of a batch of rays on the GPU.
    The AD variant returns the same detached record (derivatives of intersections are taken inside renderD)."""
    dev = _device()
    o = _torch.as_tensor(ray.o, dtype=_torch.float32).to(dev).reshape(-1, 3).contiguous()
    d = _torch.as_tensor(ray.d, dtype=_torch.float32).to(dev).reshape(-1, 3).contiguous()
    if o.shape != d.shape:
        raise RuntimeError("unit_ray_intersect: origins and directions differ in size")
    n = int(o.shape[0])
    rec = _torch.zeros((n, 24), dtype=_torch.float32, device=dev)
    if n:
        _core._ray_intersect(self, n, o.data_ptr(), d.data_ptr(), rec.data_ptr(), _stream_ptr())
    its = IntersectionC(rec)
    if active is not None:
        its._valid = its._valid & _torch.as_tensor(active, dtype=_torch.bool, device=dev).reshape(-1)
    return its


class SensorDirectSample:
    """reference include/psdr/sensor/sensor.h SensorDirectSample_: q (sample-space position), pixel_idx, sensor_val, is_valid"""

    def __init__(self, q, pixel_idx, sensor_val, is_valid):
        self.q, self.pixel_idx, self.sensor_val, self.is_valid = q, pixel_idx, sensor_val, is_valid


def _sample_direct(self, p):
    """PerspectiveCamera.sample_direct(points[N, 3]) (reference perspective.cpp:181-197): where world points land on the film of the
    configured camera and the importance they carry.  Host-side utility in torch (the kernels have their own copy, edges.h)."""
    W, H, px, py, pz, dx, dy, dz, inv_area = self._direct_params()
    if W <= 0:
        raise RuntimeError("sample_direct: the camera has not been configured by a scene")
    p = _torch.as_tensor(p, dtype=_torch.float32).reshape(-1, 3)
    M = _torch.as_tensor(_np.asarray(self.world_to_sample, dtype=_np.float32)).to(p.device)
    h = p @ M[:3, :3].T + M[:3, 3]
    wq = p @ M[3, :3] + M[3, 3]
    q = (h / wq[:, None])[:, :2]
    ix, iy = _torch.floor(q[:, 0] * W).to(_torch.int64), _torch.floor(q[:, 1] * H).to(_torch.int64)
    valid = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)
    idx = _torch.where(valid, iy * W + ix, _torch.full_like(ix, -1))
    d = p - _torch.tensor([px, py, pz], dtype=_torch.float32, device=p.device)
    dist2 = (d * d).sum(dim=1)
    d = d / _torch.sqrt(dist2)[:, None]
    cos_t = d @ _torch.tensor([dx, dy, dz], dtype=_torch.float32, device=p.device)
    val = (1.0 / dist2) * (1.0 / cos_t) ** 3 * inv_area
    return SensorDirectSample(q, idx.to(_torch.int32), val, valid)


PerspectiveCamera.sample_direct = _sample_direct
Scene.unit_ray_intersect = _unit_ray_intersect
Scene.unit_ray_intersectAD = _unit_ray_intersect
Scene.add_EnvironmentMap = _add_EnvironmentMap


def _load_file(self, file_name, auto_configure=True):
    """Scene::load_file (reference scene.cpp:75-78, scene_loader.cpp)"""
    import sys as _sys
    from . import scene_loader as _sl
    _sl.load_file(self, _os.fspath(file_name), _sys.modules[__name__], auto_configure)


def _load_string(self, scene_xml, auto_configure=True):
    """Scene::load_string (reference scene.cpp:80-83)"""
    import sys as _sys
    from . import scene_loader as _sl
    _sl.load_string(self, scene_xml, _sys.modules[__name__], auto_configure)


Scene.load_file = _load_file
Scene.load_string = _load_string
Scene.configure = _configure


def _leaves(scene, integ=None):
    """[(object, name, tensor)] for every torch-valued scene parameter (and the integrator's own, e.g. CollocatedIntegrator.m_intensity)."""
    out = []
    if integ is not None:
        for name, t in sorted(integ.__dict__.get("_psdr_params", {}).items()):
            out.append((integ, name, t))
    for key in sorted(scene.__dict__.get("_psdr_objs", {})):
        obj = scene.__dict__["_psdr_objs"][key]
        for name, t in sorted(obj.__dict__.get("_psdr_params", {}).items()):
            out.append((obj, name, t))
    return out


def _sync_params(scene, tangents=None, integ=None):
    """Push current tensor values (and optional tangents {id(tensor): array}) into the host objects."""
    for obj, name, t in _leaves(scene, integ):
        v = t.detach().to("cpu", _torch.float32).numpy()
        if name.endswith(_UV_SUFFIX):                        # [rotate, scale, translate] of a bitmap parameter
            d = tangents[id(t)] if tangents is not None and id(t) in tangents else _np.zeros(4, _np.float32)
            _uv_push(obj, name[:-len(_UV_SUFFIX)], v, d)
            continue
        shape = (4, 4) if name.startswith("to_world") else ((obj.num_vertices, 3) if name == "vertex_positions" else (-1,))
        if isinstance(obj, (_core.BSDF, _core.Emitter)) and t.dim() >= 2 and not name.startswith("to_world"):
            shape = tuple(t.shape)                           # a bitmap parameter keeps its [H, W(, 3)] shape
        if isinstance(obj, MicrofacetBSDFPerVertex):
            shape = (-1,) if name == "roughness" else (-1, 3)
        v = v.reshape(shape)
        d = _zeros_like(v)
        if tangents is not None and id(t) in tangents:
            d = _np.asarray(tangents[id(t)], dtype=_np.float32).reshape(v.shape)
        obj._set(name, v, d)


# ------------------------------------------------------------------ distributed sharding
def _shard():
    """(rank, world) when torch.distributed is initialised, else (0, 1)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _device():
    if not _torch.cuda.is_available():
        raise RuntimeError("psdr_jit_amd: no ROCm GPU visible — the renderer has no CPU fallback")
    return _torch.device("cuda", _torch.cuda.current_device())


def _stream_ptr():
    return int(_torch.cuda.current_stream().cuda_stream)


def _pix(batch_pix, dev):
    if batch_pix is None or (isinstance(batch_pix, int) and batch_pix == -1):
        return None
    return _torch.as_tensor(batch_pix, dtype=_torch.int32).to(dev).contiguous()


def _all_reduce(t, distributed):
    if distributed:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM)     # RCCL over xGMI ("nccl" backend on ROCm)
    return t


def _shard_mode():
    """the partition of every sampler's lanes over the ranks (psdr_render_args.shard_mode, include/psdr_hip.h): PSDR_SHARD=rows = contiguous runs - pixel-row tiles of the
    interior term, which an all-gather then assembles (BASELINE north_star's partition); anything else = interleaved 256-lane chunks (balanced whatever the image)"""
    return 1 if _os.environ.get("PSDR_SHARD", "interleaved").lower() == "rows" else 0


def _row_tile(n, width, world, batch):
    """pixels per rank of the contiguous partition (api.hip::shard_run): whole pixel rows of a full frame (`width` pixels each), whole pixels of a batch list"""
    unit = 1 if batch else width
    n_units = (n + unit - 1) // unit
    return ((n_units + world - 1) // world) * unit


def _gather_tiles(buf, n, tile, world, async_op=False):
    """[2, n, 3] buffers whose rank-r rows [r tile, (r + 1) tile) are final -> the whole buffers on every rank: all_gather_into_tensor of 1 / world each (the interior
    term under PSDR_SHARD=rows; an all_reduce would move the full buffer for data that is disjoint by construction).  async_op: returns a function that waits and
    returns the assembled [2, n, 3] buffer"""
    import torch.distributed as dist
    rank = dist.get_rank()
    pad = tile * world
    mine = _torch.zeros((2, tile, 3), dtype=buf.dtype, device=buf.device)
    lo, hi = min(n, rank * tile), min(n, (rank + 1) * tile)
    mine[:, :hi - lo] = buf[:, lo:hi]
    full = _torch.empty((2, pad, 3), dtype=buf.dtype, device=buf.device)
    works = [dist.all_gather_into_tensor(full[k], mine[k], async_op=True) for k in range(2)]          # (image and derivative are separate slabs)

    def finish():
        for w in works:
            w.wait()
        return full[:, :n]
    return finish if async_op else finish()


# ------------------------------------------------------------------ Integrator.renderC / renderD
def _renderC(self, scene, sensor_id=0, seed=-1, batch_pix=-1, distributed=None):
    """Integrator.renderC (reference integrator.cpp:12-48): float32 [n_pixels, 3], pixel = y*W + x."""
    dev = _device()
    pix = _pix(batch_pix, dev)
    n = int(pix.numel()) if pix is not None else scene.opts.width * scene.opts.height
    rank, world = _shard() if distributed in (None, True) else (0, 1)
    out = _torch.empty((n, 3), dtype=_torch.float32, device=dev)
    self._shard_mode = _shard_mode() if world > 1 else 0
    self._renderC(scene, sensor_id, seed, pix.data_ptr() if pix is not None else 0, n, out.data_ptr(), _stream_ptr(), rank, world)
    return _all_reduce(out, world > 1)


def _render_terms(render, n, dev, world, terms, split_ok=True, tile=0):
    """image and forward derivative of one renderD from `render(launch_terms, continue_streams, image, derivative)`, summed over the ranks.

    On several ranks (torch.distributed initialised: one process per GPU, RCCL) every rank evaluates its 256-lane chunks of the three samplers.  The interior
    term's samples belong to disjoint pixels per rank and it is the first to finish, so its [image | derivative] goes into a collective of its own as soon as
    it is launched - RCCL's stream runs it while the edge kernels, the bulk of a renderD, are still busy - and only the edge terms' derivative is reduced after
    them: the exposed collective moves a third of the bytes.  PSDR_SINGLE_COLLECTIVE=1 keeps rounds 1-4's form: all three terms, then one all_reduce of the
    stacked buffer.  (device-agnostic: tests/test_distributed_cpu.py drives both forms over gloo with the oracle as `render`)"""
    launch = terms & 7
    split = split_ok and world > 1 and (launch & TERM_INTERIOR) and (launch & (TERM_PRIMARY | TERM_SECONDARY)) and (terms >> 4) == 0 \
        and _os.environ.get("PSDR_SINGLE_COLLECTIVE", "0") != "1"
    if not split:
        buf = (_torch.empty if launch else _torch.zeros)((2, n, 3), dtype=_torch.float32, device=dev)
        render(terms, False, buf[0], buf[1])
        _all_reduce(buf, world > 1)      # one collective for image + derivative
        return buf[0], buf[1]
    import torch.distributed as dist
    buf = _torch.empty((2, n, 3), dtype=_torch.float32, device=dev)
    render(TERM_INTERIOR, False, buf[0], buf[1])
    if tile > 0:
        # PSDR_SHARD=rows: this rank's interior samples all lie in its own tile of pixel rows, so the term is ASSEMBLED, not summed - 1 / world of the bytes per rank.
        # (the gather is issued after the edge kernels are in the queue: torch's collectives order themselves behind the work of the current stream at call time)
        works = _gather_tiles(buf, n, tile, world, async_op=True)               # (waits for the interior kernel only; overlaps the launches below)
        edge = _torch.empty((2, n, 3), dtype=_torch.float32, device=dev)
        render(launch & (TERM_PRIMARY | TERM_SECONDARY), True, edge[0], edge[1])
        dist.all_reduce(edge[1], op=dist.ReduceOp.SUM)
        full = works()
        return full[0], full[1] + edge[1]
    work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)          # (waits for the interior kernel only; overlaps the launches below)
    edge = _torch.empty((2, n, 3), dtype=_torch.float32, device=dev)            # edge[0]: the edge terms' primal, identically zero (integrator.cpp:192, path.cpp:265) - not reduced
    # the edge terms continue from the sampler state the first call left (it has seeded all three streams and advanced the interior one)
    render(launch & (TERM_PRIMARY | TERM_SECONDARY), True, edge[0], edge[1])
    dist.all_reduce(edge[1], op=dist.ReduceOp.SUM)
    work.wait()
    return buf[0], buf[1] + edge[1]


def _render_d_raw(self, scene, sensor_id, seed, batch_pix, terms, distributed=None):
    dev = _device()
    pix = _pix(batch_pix, dev)
    n = int(pix.numel()) if pix is not None else scene.opts.width * scene.opts.height
    rank, world = _shard() if distributed in (None, True) else (0, 1)
    pp = pix.data_ptr() if pix is not None else 0
    self._shard_mode = _shard_mode() if world > 1 else 0
    tile = _row_tile(n, scene.opts.width, world, pix is not None) if self._shard_mode == 1 else 0

    def render(launch_terms, continue_streams, image, derivative):
        self._renderD(scene, sensor_id, -1 if continue_streams else seed, pp, n, image.data_ptr(), derivative.data_ptr(), _stream_ptr(), rank, world, launch_terms)
    return _render_terms(render, n, dev, world, terms, split_ok=pix is None, tile=tile)


def _bsdf_index(scene, obj):
    """row of a BSDF in the configured snapshot: the scene's own BSDFs in order, then the BSDFs nested in normal maps"""
    pm = scene.param_map
    nb = sum(1 for k in pm if k.startswith("BSDF[") and not k.startswith("BSDF[id="))
    hidden = nb
    for k in range(nb):
        b = pm["BSDF[%d]" % k]
        if b is obj:
            return k
        if isinstance(b, NormalMapBSDF):
            if b._nested is obj:
                return hidden
            hidden += 1
    raise RuntimeError("BSDF is not part of the scene")


def _mesh_index(scene, mesh):
    for i in range(scene.num_meshes):
        if scene.param_map.get("Mesh[%d]" % i) is mesh:
            return i
    raise RuntimeError("mesh is not part of the scene")


class _intra_op_threads:
    """caps torch's intra-op thread pool for a block of small CPU tensor work and restores it on every way out"""

    def __init__(self, cap):
        self.cap = cap

    def __enter__(self):
        self.n = _torch.get_num_threads()
        if self.n > self.cap:
            _torch.set_num_threads(self.cap)

    def __exit__(self, *exc):
        if self.n > self.cap:
            _torch.set_num_threads(self.n)
        return False


class _RenderDFn(_torch.autograd.Function):
    """Autograd node of renderD.  forward = primal image; backward = the reverse-mode kernels
    (psdr_hip_render_d_bwd: adjoints of the configured snapshot) followed by the host chain rule of
    chain.py down to the leaf tensors (mesh transforms, vertex positions, reflectances, radiances)."""

    @staticmethod
    def forward(ctx, state, *leaf_tensors):
        ctx.state = state
        return state["img"]

    @staticmethod
    def backward(ctx, grad_img):
        from . import chain
        st = ctx.state
        integ, scene = st["integrator"], st["scene"]
        scene.__dict__["_psdr_fwd_hint"] = None          # this scene's images go to reverse mode: renderD makes no forward-mode launch ahead of time
        leaves = st["leaves"]
        needs = ctx.needs_input_grad[1:]
        want_cam = any(need and isinstance(obj, Sensor) for (obj, name, t), need in zip(leaves, needs))
        dev = grad_img.device
        g_img = grad_img.contiguous().to(_torch.float32)
        n_tris, n_sec = (int(x) for x in scene._snapshot_counts())
        cam = scene.param_map["Sensor[%d]" % st["sensor_id"]]
        n_prim = int(_np.asarray(cam._primary_edge_ids()).reshape(-1, 3).shape[0])
        pm = scene.param_map
        nb = sum(1 for k in pm if k.startswith("BSDF[") and not k.startswith("BSDF[id="))
        ne = sum(1 for k in pm if k.startswith("Emitter[") and not k.startswith("Emitter[id="))
        # BSDF rows of the snapshot: the scene's own, then the ones nested in normal maps
        n_hidden = sum(1 for k in range(nb) if isinstance(pm["BSDF[%d]" % k], NormalMapBSDF))
        sizes = [n_tris * 22, max(1, nb + n_hidden) * 3, max(1, ne) * 3, max(1, n_sec) * 6, max(1, n_prim) * 4]
        flat = _torch.zeros(sum(sizes), dtype=_torch.float32, device=dev)
        offs = [0]
        for z in sizes:
            offs.append(offs[-1] + z)
        ptr = [flat.data_ptr() + 4 * o for o in offs[:-1]]
        if st["seed"] != -1:
            seeds, skips = [st["seed"]] * 3, [0, 0, 0]
        else:
            seeds, skips = [s[2] for s in st["samplers"]], [s[3] for s in st["samplers"]]
        rank, world = _shard()
        # requires_grad at the level of the configured snapshot: which meshes' triangle rows, and whether any BSDF colour /
        # emitter radiance, are wanted.  The interior adjoint probes only those (psdr_grads.mesh_filter / skip_*).
        want_mesh = _np.zeros(max(1, scene.num_meshes), dtype=_np.uint8)
        want_bsdf = want_em = False
        for (obj, name, t), need in zip(leaves, needs):
            if not need:
                continue
            if isinstance(obj, Mesh):
                want_mesh[_mesh_index(scene, obj)] = 1
            elif isinstance(obj, _core.BSDF):
                want_bsdf = want_bsdf or (t.dim() < 2 and name in ("reflectance", "diffuseReflectance"))     # (the rest: g_tex / g_mat)
            elif isinstance(obj, _core.Emitter):
                want_em = want_em or not isinstance(obj, EnvironmentMap)     # (the map's adjoints come back in g_env / g_env_scale)
        mesh_filter = _torch.from_numpy(want_mesh).to(dev)
        # bitmap parameters (a leaf of 2 or 3 dimensions on a BSDF) and per-vertex values: their adjoints come back in one flat buffer
        tex_leaves = [i for i, ((obj, name, t), need) in enumerate(zip(leaves, needs)) if need and isinstance(obj, _core.BSDF) and (t.dim() >= 2 or isinstance(obj, MicrofacetBSDFPerVertex))]
        # rotate / scale / translate of a bitmap: four scalars per bitmap, filled by the same pass (psdr_grads.g_uv_xf beside g_tex / g_env)
        uv_leaves = [i for i, ((obj, name, t), need) in enumerate(zip(leaves, needs)) if need and name.endswith(_UV_SUFFIX)]
        g_uv = _torch.zeros(4 * (3 * (nb + n_hidden) + 1), dtype=_torch.float32, device=dev) if uv_leaves else None
        g_tex = None
        if tex_leaves or any(isinstance(leaves[i][0], _core.BSDF) for i in uv_leaves):
            tex_off, tex_total = _core._tex_layout(scene)
            g_tex = _torch.zeros(max(1, int(tex_total)), dtype=_torch.float32, device=dev)
        g_cam = _torch.zeros(16, dtype=_torch.float32, device=dev) if want_cam else None
        bpix = _pix(st["batch_pix"], dev)             # batch rendering: the adjoint image is [len(batch_pix), 3]
        # constant parameters of the GGX BSDFs: (row offset, length) inside the BSDF's 16-float row of psdr_grads.g_mat
        mat_rows = {"MicrofacetBSDF": {"specularReflectance": (0, 3), "roughness": (3, 1)},
                    "RoughConductorBSDF": {"alpha_u": (0, 1), "alpha_v": (1, 1), "eta": (2, 3), "k": (5, 3), "specular_reflectance": (8, 3)},
                    "RoughDielectricBSDF": {"alpha_u": (0, 1), "alpha_v": (1, 1), "eta": (2, 1)}}
        mat_leaves = [i for i, ((obj, name, t), need) in enumerate(zip(leaves, needs))
                      if need and t.dim() < 2 and name in mat_rows.get(type(obj).__name__, {})]
        g_mat = _torch.zeros(16 * max(1, nb + n_hidden), dtype=_torch.float32, device=dev) if mat_leaves else None
        env_leaves = [i for i, ((obj, name, t), need) in enumerate(zip(leaves, needs)) if need and isinstance(obj, EnvironmentMap) and not name.endswith(_UV_SUFFIX)]
        g_env = g_env_scale = g_env_xf = None
        for i in env_leaves:
            obj, name, t = leaves[i]
            if name == "radiance":
                g_env = _torch.zeros(t.numel(), dtype=_torch.float32, device=dev)
            elif name == "scale":
                g_env_scale = _torch.zeros(1, dtype=_torch.float32, device=dev)
            else:               # to_world_left: through the adjoint of from_world = (to_world_left . to_world_raw)^-1
                g_env_xf = _torch.zeros(16, dtype=_torch.float32, device=dev)
        if g_env is None and any(isinstance(leaves[i][0], EnvironmentMap) for i in uv_leaves):
            g_env = _torch.zeros(int(_np.prod(leaves[uv_leaves[0]][0]._get("radiance", False).shape)), dtype=_torch.float32, device=dev)
        if g_env_scale is not None and g_env is None:        # the scale adjoint is assembled from the texel probes
            g_env = _torch.zeros(int(_np.prod(leaves[env_leaves[0]][0]._get("radiance", False).shape)), dtype=_torch.float32, device=dev)
        # the boundary terms' adjoints are rows of edges (g_sec / g_prim): nobody reads them unless a mesh or the camera is differentiated
        bwd_terms = _derivative_terms(st["terms"], leaves, {id(t) for (_o, _n, t), need in zip(leaves, needs) if need}) & 7
        # ... and the primary-edge samples add to the row of THEIR edge only: with some meshes differentiated and the camera (and the integrator) not, the samples
        # on the other meshes' edges are not traced (psdr_grads.prim_edge_filter)
        prim_filter = None
        if (bwd_terms & TERM_PRIMARY) and n_prim > 0 and not any(need and not isinstance(obj, Mesh) and _moves_edges(obj, name) for (obj, name, t), need in zip(leaves, needs)):
            edge_mesh = _np.asarray(cam._primary_edge_ids(), dtype=_np.int64).reshape(-1, 3)[:, 0]
            prim_filter = _torch.from_numpy(_np.ascontiguousarray(want_mesh[edge_mesh])).to(dev)
        integ._shard_mode = _shard_mode() if world > 1 else 0            # (the partition of the forward pass: the adjoints of all ranks are summed below, whatever the partition)
        _core._render_d_bwd(integ, scene, st["sensor_id"], seeds, skips, g_img.data_ptr(), ptr[0], ptr[1], ptr[2], ptr[3], ptr[4],
                            _stream_ptr(), rank, world, bwd_terms, mesh_filter.data_ptr(), not want_bsdf, not want_em,
                            g_tex.data_ptr() if g_tex is not None else 0, g_cam.data_ptr() if g_cam is not None else 0,
                            g_env.data_ptr() if g_env is not None else 0, g_env_scale.data_ptr() if g_env_scale is not None else 0,
                            g_mat.data_ptr() if g_mat is not None else 0, g_env_xf.data_ptr() if g_env_xf is not None else 0,
                            bpix.data_ptr() if bpix is not None else 0, int(bpix.numel()) if bpix is not None else 0,
                            g_uv.data_ptr() if g_uv is not None else 0, prim_filter.data_ptr() if prim_filter is not None else 0)
        _all_reduce(flat, world > 1)
        for extra in (g_env, g_env_scale, g_mat, g_env_xf, g_uv):
            if extra is not None:
                _all_reduce(extra, world > 1)
        if g_cam is not None:
            _all_reduce(g_cam, world > 1)
        if g_tex is not None:
            _all_reduce(g_tex, world > 1)
        g = flat.to("cpu", _torch.float64)
        g_bsdf_all = g[offs[1]:offs[2]].reshape(-1, 3)
        g_tri, g_bsdf, g_em = g[offs[0]:offs[1]].reshape(n_tris, 22), g_bsdf_all[:nb], g[offs[2]:offs[3]].reshape(-1, 3)[:ne]
        g_sec, g_prim = g[offs[3]:offs[4]].reshape(-1, 6)[:n_sec], g[offs[4]:offs[5]].reshape(-1, 4)[:n_prim]

        # From the adjoints of the snapshot rows to the leaves: the per-triangle / per-edge part of the chain rule runs in the host core (Scene::chain_geometry through
        # chain.native_geometry_grads: double arithmetic on host threads; rounds 1-4 differentiated a torch restatement of configure() here, 25 ms per step on config 5),
        # colours and radiances are rows of g_bsdf / g_emitter.
        grads = [None] * len(leaves)
        g_np = g.numpy()
        geo_wanted = [(obj, name) for (obj, name, t), need in zip(leaves, needs)
                      if need and ((isinstance(obj, Mesh) and name in ("vertex_positions", "to_world_left", "to_world", "to_world_right")) or
                                   (isinstance(obj, Sensor) and name in ("to_world_left", "to_world", "to_world_right")))]
        if geo_wanted:
            geo = chain.native_geometry_grads(scene, st["sensor_id"], geo_wanted, g_np[offs[0]:offs[1]], g_np[offs[3]:offs[3] + 6 * n_sec], g_np[offs[4]:offs[4] + 4 * n_prim],
                                              g_cam.to("cpu", _torch.float64).numpy().reshape(4, 4) if g_cam is not None else None)
        for i, ((obj, name, t), need) in enumerate(zip(leaves, needs)):
            if not need:
                continue
            if (obj, name) in [(o, n) for o, n in geo_wanted] and (id(obj), name) in geo:
                grads[i] = _torch.as_tensor(_np.ascontiguousarray(geo[(id(obj), name)])).reshape(t.shape).to(t.device, t.dtype)
            elif isinstance(obj, _core.BSDF) and t.dim() < 2 and name == ("diffuseReflectance" if isinstance(obj, MicrofacetBSDF) else "reflectance") \
                    and isinstance(obj, (DiffuseBSDF, MicrofacetBSDF)):
                b = _bsdf_index(scene, obj)
                if b < nb:                      # (a BSDF nested in a normal map: below)
                    row = g_bsdf[b].to(_torch.float32)
                    grads[i] = (row if t.numel() == 3 else row.sum().reshape(1)).reshape(t.shape).to(t.device, t.dtype)
            elif isinstance(obj, AreaLight) and name == "radiance":
                e = [k for k in range(ne) if pm.get("Emitter[%d]" % k) is obj]
                if e:
                    row = g_em[e[0]].to(_torch.float32)
                    grads[i] = (row if t.numel() == 3 else row.sum().reshape(1)).reshape(t.shape).to(t.device, t.dtype)
        for i in tex_leaves:          # the leaf IS the texel array: its gradient is its block of g_tex
            obj, name, t = leaves[i]
            b = _bsdf_index(scene, obj)
            slot = {"reflectance": 0, "diffuseReflectance": 0, "specularReflectance": 1, "roughness": 2, "normal_map": 0,
                    "eta": 0, "k": 1, "alpha_u": 2, "alpha_v": 2}[name]       # the three bitmap / per-vertex slots of a BSDF
            off = int(tex_off[3 * b + slot])
            grads[i] = _torch.zeros_like(t) if off < 0 else g_tex[off:off + t.numel()].reshape(t.shape).to(t.device, t.dtype)
        for i, ((obj, name, t), need) in enumerate(zip(leaves, needs)):      # the colour of a BSDF nested in a normal map: its own g_bsdf row
            if need and isinstance(obj, _core.BSDF) and t.dim() < 2 and name in ("reflectance", "diffuseReflectance"):
                b = _bsdf_index(scene, obj)
                if b >= nb:
                    row = g_bsdf_all[b].to(_torch.float32)
                    grads[i] = (row if t.numel() == 3 else row.sum().reshape(1)).reshape(t.shape).to(t.device, t.dtype)
        for i in mat_leaves:          # a colour given as one number is the sum of its three components
            obj, name, t = leaves[i]
            b = _bsdf_index(scene, obj)
            off, n = mat_rows[type(obj).__name__][name]
            row = g_mat[16 * b + off:16 * b + off + n]
            grads[i] = (row if t.numel() == n else row.sum().reshape(1)).reshape(t.shape).to(t.device, t.dtype)
        for i, ((obj, name, t), need) in enumerate(zip(leaves, needs)):      # CollocatedIntegrator.m_intensity: the image is linear in it
            if need and obj is integ and name == "m_intensity":
                grads[i] = ((g_img * st["img"]).sum() / t.detach().to(dev).reshape(())).reshape(t.shape).to(t.device, t.dtype)
        for i in env_leaves:
            obj, name, t = leaves[i]
            if name in ("radiance", "scale"):
                src = g_env if name == "radiance" else g_env_scale
                grads[i] = src.reshape(t.shape).to(t.device, t.dtype)
            else:               # from_world = inverse(to_world_left . to_world_raw): d from = -from . d to . from
                with _torch.enable_grad():
                    L = t.detach().to("cpu", _torch.float64).reshape(4, 4).clone().requires_grad_(True)
                    raw = _torch.as_tensor(_np.asarray(obj._get("to_world_raw", False), dtype=_np.float64))
                    fw = _torch.linalg.inv(L @ raw)
                    (gl,) = _torch.autograd.grad(fw, L, g_env_xf.to("cpu", _torch.float64).reshape(4, 4))
                grads[i] = gl.reshape(t.shape).to(t.device, t.dtype)
        for i in uv_leaves:           # [rotate, scale, translate.x, translate.y]: the bitmap's row of g_uv_xf (the last row is the environment map's)
            obj, name, t = leaves[i]
            row = 3 * (nb + n_hidden) if isinstance(obj, EnvironmentMap) else 3 * _bsdf_index(scene, obj) + _UV_SLOT[name[:-len(_UV_SUFFIX)]]
            grads[i] = g_uv[4 * row:4 * row + 4].reshape(t.shape).to(t.device, t.dtype)
        return (None,) + tuple(grads)


def _moves_edges(obj, name):
    """Does this leaf reach the boundary terms?  Their integrand is detach(radiance difference) x the normal velocity of the edge point (reference
    integrator.cpp:179-198, path.cpp:171-294, scene.cpp:1027-1068): the only differentiable factor is the edge's geometry - mesh vertices and transforms, the
    camera pose - so colours, material constants, bitmaps, radiances and the environment map's texels / scale get exactly zero from both edge terms.  (The
    map's to_world and the integrators' own parameters are kept on the full path.)"""
    return isinstance(obj, (Mesh, Sensor)) or isinstance(obj, Integrator) or (isinstance(obj, EnvironmentMap) and name.startswith("to_world"))


def _derivative_terms(terms, leaves, moving):
    """the terms a derivative pass has to LAUNCH for the leaves in `moving` (ids of leaf tensors with a tangent / that need an adjoint): without a leaf that
    moves an edge the two boundary terms contribute zeros and only their samplers advance (bits 4-6, as in renderD's primal pass) - the images, derivatives
    and sampler streams are what the full launch gives, bit for bit, at the cost of the interior term alone (BASELINE config 5: a fifth)"""
    if any(id(t) in moving and _moves_edges(obj, name) for (obj, name, t) in leaves):
        return terms
    advance = ((terms >> 4) & 7) or (terms & 7)
    return (terms & TERM_INTERIOR) | (advance << 4)


def _replay_forward(integ, scene, st, tangents):
    """Re-render with the sampler state of the recorded call and the given leaf tangents."""
    _sync_params(scene, tangents, integ)
    scene._configure(st["active"])
    after = [scene._sampler_state(k) for k in range(3)]
    for k, s in enumerate(st["samplers"]):
        scene._set_sampler_state(k, *s)
    _, dimg = _render_d_raw(integ, scene, st["sensor_id"], st["seed"], st["batch_pix"], _derivative_terms(st["terms"], st["leaves"], tangents))
    for k, s in enumerate(after):
        scene._set_sampler_state(k, *s)
    return dimg


def _jvp_tangents(leaves, param, direction=None):
    """{id(leaf tensor): d leaf / d param . direction} for the scene's torch parameters (direction default: ones): the tangents
    drjit.set_grad(param, direction); drjit.forward_to(...) would carry into the scene (reference README.md:102-104)"""
    tangents = {}
    v = _torch.ones_like(param) if direction is None else direction
    for (obj, name, t) in leaves:
        if not t.requires_grad:
            continue
        if t is param:
            tangents[id(t)] = v.detach().cpu().numpy()
            continue
        w = _torch.zeros_like(t, requires_grad=True)
        (g,) = _torch.autograd.grad(t, param, w, create_graph=True, allow_unused=True)
        if g is None:
            continue
        (jv,) = _torch.autograd.grad(g, w, v, allow_unused=True)
        if jv is not None:
            tangents[id(t)] = jv.detach().cpu().numpy()
    return tangents


def _renderD(self, scene, sensor_id=0, seed=-1, batch_pix=-1, terms=TERM_ALL):
    """Integrator.renderD (reference integrator.cpp:51-100).  Returns the image as a tensor attached
    to the autograd graph of the scene's torch parameters."""
    import weakref
    leaves = _leaves(scene, self)
    for _name, _t in self.__dict__.get("_psdr_params", {}).items():       # the integrator's own tensor parameters (m_intensity)
        _v = _t.detach().to("cpu", _torch.float32).numpy().reshape(-1)
        self._set(_name, _v, _zeros_like(_v))
    state = {"integrator": self, "scene": scene, "sensor_id": sensor_id, "batch_pix": batch_pix, "terms": terms,
             "active": scene.__dict__.get("_psdr_active", []), "leaves": leaves, "seed": seed,
             "samplers": [scene._sampler_state(k) for k in range(3)]}      # the streams this call starts from
    tens = [t for (_, _, t) in leaves]
    if any(t.requires_grad for t in tens):
        # The derivative is computed later (forward_grad / backward) from the recorded sampler state: in general only the primal image is
        # needed now, and both edge terms have zero primal - launch the interior term, advance all three samplers.
        # When the previous image of this scene went to forward_grad(img, P) for a P that is still alive, this call is expected to go the
        # same way (an optimisation loop repeats its step): the tangents of P are installed and the ONE launch that forward mode needs is
        # made now - image and derivative from the same pass, as the reference gets them from one recorded renderD - instead of a primal pass
        # here and a full replay in forward_grad.  A wrong guess costs the edge terms of one call, never a result: forward_grad and backward
        # check what they find.
        hint = scene.__dict__.get("_psdr_fwd_hint")
        param = hint["param"]() if hint else None
        if param is not None and param.requires_grad and (terms & 7) != 0:
            tangents = _jvp_tangents(leaves, param)
            if tangents:
                _sync_params(scene, tangents, self)
                scene._configure(state["active"])
                img, dimg = _render_d_raw(self, scene, sensor_id, seed, batch_pix, _derivative_terms(terms, leaves, tangents))
                _sync_params(scene, None, self)
                scene._configure(state["active"])
                state["img"], state["dimg"], state["dimg_param"] = img, dimg, weakref.ref(param)
                return _RenderDFn.apply(state, *tens)
        img, _ = _render_d_raw(self, scene, sensor_id, seed, batch_pix, (terms & TERM_INTERIOR) | ((terms & 7) << 4))
        state["img"] = img
        return _RenderDFn.apply(state, *tens)
    img, dimg = _render_d_raw(self, scene, sensor_id, seed, batch_pix, terms)
    return img


def forward_grad(img, param, direction=None):
    """d img / d param along `direction` (default: ones) — the torch spelling of
    drjit.set_grad(P, 1); drjit.forward_to(img); drjit.grad(img) (reference README.md:102-104)."""
    import weakref
    fn = img.grad_fn
    if fn is None or not hasattr(fn, "state"):
        raise RuntimeError("forward_grad: img does not come from renderD with differentiable scene parameters")
    st = fn.state
    scene = st["scene"]
    scene.__dict__["_psdr_fwd_hint"] = {"param": weakref.ref(param)}       # (the next renderD of this scene: see _renderD)
    if direction is None and st.get("dimg") is not None and st["dimg_param"]() is param:
        return st["dimg"]                                                   # renderD made the forward-mode launch already
    tangents = _jvp_tangents(st["leaves"], param, direction)
    dimg = _replay_forward(st["integrator"], scene, st, tangents)
    _sync_params(scene, None, st["integrator"])
    scene._configure(st["active"])
    return dimg


Integrator.renderC = _renderC
Integrator.renderD = _renderD
PathTracer.renderC = _renderC
PathTracer.renderD = _renderD
Direct.renderC = _renderC
Direct.renderD = _renderD
FieldExtractionIntegrator = _core.FieldExtractionIntegrator
CollocatedIntegrator = _core.CollocatedIntegrator
for _cls in (FieldExtractionIntegrator, CollocatedIntegrator):
    _cls.renderC = _renderC
    _cls.renderD = _renderD
CollocatedIntegrator.m_intensity = _make_param_property("m_intensity", lambda self, value: (1,))


def render_d_fwd(integrator, scene, sensor_id=0, seed=-1, batch_pix=-1, terms=TERM_ALL, tangents=None):
    """One-shot renderD + forward derivative: returns (img, d_img).  `tangents` maps scene leaf tensors
    (by identity) to their tangent arrays; host objects whose tangents were set through `_set` keep them."""
    if tangents is not None:
        _sync_params(scene, {id(k): v for k, v in tangents.items()})
        scene._configure(scene.__dict__.get("_psdr_active", []))
    return _render_d_raw(integrator, scene, sensor_id, seed, batch_pix, terms)
