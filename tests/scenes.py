"""Scene descriptions shared by the tests: the README Cornell box (reference README.md:54-85,
8 meshes / 36 triangles / 66 mesh edges) and the tutorial sphere box (tutorials/Forward_AD.ipynb
cells 2-5, 652 triangles).  Geometry comes from the OBJ data files under examples/data/cbox.

The OBJ reader here is the tests' own (fan triangulation = what an ear-clipping triangulator
yields for convex polygons, which is what the reference gets from tinyobj)."""
import os
import sys

import numpy as np

from oracle.oracle import BsdfSpec, CameraSpec, EmitterSpec, MeshSpec, SceneSpec

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import synth                                                # noqa: E402  (file-less inputs shared with bench.py)
from synth import icosphere, synthetic_envmap                # noqa: E402,F401

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "data", "cbox")


def load_obj(path):
    v, vt, f, fuv = [], [], [], []
    with open(path) as fh:
        for line in fh:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                v.append([float(x) for x in p[1:4]])
            elif p[0] == "vt":
                vt.append([float(x) for x in p[1:3]])
            elif p[0] == "f":
                idx = [q.split("/") for q in p[1:]]
                vi = [int(q[0]) - 1 for q in idx]
                ti = [int(q[1]) - 1 if len(q) > 1 and q[1] != "" else -1 for q in idx]
                for k in range(1, len(vi) - 1):
                    f.append([vi[0], vi[k], vi[k + 1]])
                    fuv.append([ti[0], ti[k], ti[k + 1]])
    v = np.asarray(v, dtype=np.float32)
    f = np.asarray(f, dtype=np.int32)
    if vt:
        return v, f, np.asarray(vt, dtype=np.float32), np.asarray(fuv, dtype=np.int32)
    return v, f, None, None


def translate(x, y, z):
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = [x, y, z]
    return m


def _mesh(name, bsdf, emitter=-1, raw=None):
    path = os.path.join(DATA, name)
    stem = name[len("cbox_"):-len(".obj")] if name.startswith("cbox_") and name.endswith(".obj") else None
    if stem in synth.CORNELL_BOX:                 # the README's eight meshes come from coordinates (examples/synth.py), not from shipped files
        path = synth.write_cornell_box()[stem]
    v, f, uv, fuv = load_obj(path)
    m = MeshSpec(vertices=v, faces=f, uvs=uv, face_uvs=fuv, bsdf=bsdf, emitter=emitter, path=path)
    if raw is not None:
        m.to_world_raw = raw
    return m


def cbox_scene(width=128, height=128, spp=4, sppe=0, sppse=0, param="light_x"):
    """README scene.  param selects the scalar the tangent data refers to:
       'light_x'  : Mesh[0].to_world_left = T(100*P, 0, 0), P = 0   (README.md:87-90)
       'box_x'    : same for Mesh[1] (small box)
       'albedo'   : d red-wall reflectance / dP = (1,1,1)
       'radiance' : d light radiance / dP = (1,1,1)
       'camera_x' : camera to_world_left = T(100*P, 0, 0)
       None       : no tangent."""
    bsdfs = [BsdfSpec((0.0, 0.0, 0.0), name="light"), BsdfSpec((0.5, 0.5, 0.5), name="cat"),
             BsdfSpec((0.95, 0.95, 0.95), name="white"), BsdfSpec((0.20, 0.90, 0.20), name="green"),
             BsdfSpec((0.90, 0.20, 0.20), name="red")]
    emitters = [EmitterSpec((20.0, 20.0, 8.0))]
    meshes = [
        _mesh("cbox_luminaire.obj", 0, emitter=0, raw=translate(0.0, -0.5, 0.0)),
        _mesh("cbox_smallbox.obj", 1), _mesh("cbox_largebox.obj", 1),
        _mesh("cbox_floor.obj", 2), _mesh("cbox_ceiling.obj", 2), _mesh("cbox_back.obj", 2),
        _mesh("cbox_greenwall.obj", 3), _mesh("cbox_redwall.obj", 4),
    ]
    cam = CameraSpec(60.0, 0.000001, 10000000.0, to_world_raw=translate(208.0, 273.0, -800.0))
    dT = np.zeros((4, 4), dtype=np.float32)
    dT[0, 3] = 100.0
    if param == "light_x":
        meshes[0].d_to_world_left = dT
    elif param == "box_x":
        meshes[1].d_to_world_left = dT
    elif param == "albedo":
        bsdfs[4].d_reflectance = (1.0, 1.0, 1.0)
    elif param == "radiance":
        emitters[0].d_radiance = (1.0, 1.0, 1.0)
    elif param == "camera_x":
        cam.d_to_world_left = dT
    elif param is not None:
        raise ValueError(param)
    return SceneSpec(meshes, bsdfs, emitters, [cam], width, height, spp, sppe, sppse)


def set_param_value(spec, param, P):
    """Return the spec with the scalar parameter moved to value P (for finite differences)."""
    import copy
    s = copy.deepcopy(spec)
    if param == "light_x":
        s.meshes[0].to_world_left = translate(100.0 * P, 0, 0)
    elif param == "box_x":
        s.meshes[1].to_world_left = translate(100.0 * P, 0, 0)
    elif param == "albedo":
        r = s.bsdfs[4].reflectance
        s.bsdfs[4].reflectance = (r[0] + P, r[1] + P, r[2] + P)
    elif param == "radiance":
        r = s.emitters[0].radiance
        s.emitters[0].radiance = (r[0] + P, r[1] + P, r[2] + P)
    elif param == "camera_x":
        s.cameras[0].to_world_left = translate(100.0 * P, 0, 0)
    else:
        raise ValueError(param)
    return s


def sphere_scene(width=512, height=512, spp=32, sppe=32, sppse=32):
    """tutorials/Forward_AD.ipynb cells 2-5: 8 meshes, 652 triangles; the notebook logs
    '(79) primary edges initialized' and '990 secondary edges initialized'."""
    bsdfs = [BsdfSpec((0.2, 0.9, 0.9), name="sphere_large"), BsdfSpec((0.5, 0.5, 0.5), name="back"),
             BsdfSpec((0.5, 0.5, 0.5), name="light"), BsdfSpec((0.9, 0.6, 0.1), name="sphere_small"),
             BsdfSpec((0.95, 0.95, 0.95), name="white"), BsdfSpec((0.2, 0.9, 0.2), name="green"),
             BsdfSpec((0.9, 0.2, 0.2), name="red")]
    emitters = [EmitterSpec((20.0, 20.0, 8.0))]
    meshes = [
        _mesh("cbox_luminaire.obj", 2, emitter=0, raw=translate(0.0, -0.5, 0.0)),
        _mesh("cbox_smallball.obj", 3), _mesh("cbox_largeball.obj", 0),
        _mesh("cbox_floor.obj", 4), _mesh("cbox_ceiling.obj", 4), _mesh("cbox_back.obj", 1),
        _mesh("cbox_greenwall.obj", 5), _mesh("cbox_redwall.obj", 6),
    ]
    dT = np.zeros((4, 4), dtype=np.float32)
    dT[0, 3] = 100.0
    meshes[0].d_to_world_left = dT
    meshes[1].d_to_world_left = dT
    cam = CameraSpec(60.0, 0.000001, 10000000.0, to_world_raw=translate(278.0, 273.0, -500.0))
    return SceneSpec(meshes, bsdfs, emitters, [cam], width, height, spp, sppe, sppse)


def envmap_scene(width=64, height=64, spp=4, sppe=0, sppse=0, param="albedo", env=None, area_light=False, floor_only=False, balls=False):
    """Cornell-box furniture (floor + the two boxes) under an environment map - the Forward_AD_envmap layout in
    small.  param: 'albedo' (d box reflectance / dP = (1,1,1)), 'box_x' (small box translated by 100*P in x), None."""
    bsdfs = [BsdfSpec((0.5, 0.5, 0.5), name="cat"), BsdfSpec((0.8, 0.8, 0.8), name="white"), BsdfSpec((0.0, 0.0, 0.0), name="light")]
    emitters = [EmitterSpec(type=1, env_data=env if env is not None else synthetic_envmap(), env_scale=1.0)]
    meshes = [_mesh("cbox_floor.obj", 1)]
    if balls:      # the tutorial spheres (304 + 304 triangles): more than 64 triangles, so the BVH path is taken
        meshes = [_mesh("cbox_smallball.obj", 0), _mesh("cbox_largeball.obj", 0)] + meshes
    elif not floor_only:
        meshes = [_mesh("cbox_smallbox.obj", 0), _mesh("cbox_largebox.obj", 0)] + meshes
    if area_light:
        emitters.append(EmitterSpec((20.0, 20.0, 8.0)))
        meshes.append(_mesh("cbox_luminaire.obj", 2, emitter=1, raw=translate(0.0, -100.0, 0.0)))
    cam = CameraSpec(60.0, 0.000001, 10000000.0, to_world_raw=translate(278.0, 400.0, -700.0) @ _rot_x(np.radians(25.0)))
    if param == "albedo":
        bsdfs[0].d_reflectance = (1.0, 1.0, 1.0)
    elif param == "box_x" and not floor_only:
        dT = np.zeros((4, 4), dtype=np.float32)
        dT[0, 3] = 100.0
        meshes[0].d_to_world_left = dT
    elif param == "box_rot" and not floor_only:
        # the first mesh turns about the vertical axis through (185, 0, 169): positions AND the interpolated vertex normals carry a tangent
        dT = np.zeros((4, 4), dtype=np.float32)
        dT[0, 2], dT[2, 0] = 1.0, -1.0
        dT[0, 3], dT[2, 3] = -169.0, 185.0
        meshes[0].d_to_world_left = dT
    elif param in ("box_rot_x", "box_rot_z") and not floor_only:
        # ... about a horizontal axis through (185, 80, 169)
        dT = np.zeros((4, 4), dtype=np.float32)
        if param == "box_rot_x":
            dT[1, 2], dT[2, 1] = -1.0, 1.0
            dT[1, 3], dT[2, 3] = 169.0, -80.0
        else:
            dT[0, 1], dT[1, 0] = -1.0, 1.0
            dT[0, 3], dT[1, 3] = 80.0, -185.0
        meshes[0].d_to_world_left = dT
    elif param is not None:
        raise ValueError(param)
    return SceneSpec(meshes, bsdfs, emitters, [cam], width, height, spp, sppe, sppse)


def _rot_x(a):
    m = np.eye(4, dtype=np.float32)
    c, s = np.cos(a), np.sin(a)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    return m


def config5_scene(width=128, height=128, spp=4, sppe=4, sppse=4, level=6, env_res=(1024, 512), param="albedo"):
    """BASELINE config 5: a ~100k-triangle mesh on a floor under a synthetic 1024x512 lat-long map
    (constant sky + one Gaussian sun), DiffuseBSDF albedo of the mesh as the parameter."""
    v, f = icosphere(level, radius=150.0, noise=0.01, seed=0)
    bsdfs = [BsdfSpec((0.5, 0.5, 0.5), name="blob"), BsdfSpec((0.8, 0.8, 0.8), name="white")]
    emitters = [EmitterSpec(type=1, env_data=synthetic_envmap(env_res[0], env_res[1]), env_scale=1.0)]
    blob = MeshSpec(vertices=v, faces=f, uvs=None, face_uvs=None, bsdf=0, emitter=-1)
    blob.to_world_raw = translate(278.0, 160.0, 280.0)
    meshes = [blob, _mesh("cbox_floor.obj", 1)]
    cam = CameraSpec(60.0, 0.000001, 10000000.0, to_world_raw=translate(278.0, 400.0, -700.0) @ _rot_x(np.radians(25.0)))
    if param == "albedo":
        bsdfs[0].d_reflectance = (1.0, 1.0, 1.0)
    elif param == "blob_x":              # the mesh itself moves: every vertex, every edge (the shape-optimisation case; both boundary terms are non-zero)
        dT = np.zeros((4, 4), dtype=np.float32)
        dT[0, 3] = 100.0
        blob.d_to_world_left = dT
    elif param is not None:
        raise ValueError(param)
    return SceneSpec(meshes, bsdfs, emitters, [cam], width, height, spp, sppe, sppse)


def ramp_texture(width=16, height=8):
    """reflectance texture that is linear in (u, v): bilinear interpolation reproduces it exactly"""
    u = np.arange(width, dtype=np.float64) / (width - 1)
    v = np.arange(height, dtype=np.float64) / (height - 1)
    uu, vv = np.meshgrid(u, v)
    return np.stack([0.2 + 0.6 * uu, 0.8 - 0.5 * vv, 0.3 + 0.2 * uu + 0.3 * vv], axis=-1).astype(np.float32)


def checker_texture(width=32, height=32, cells=4):
    yy, xx = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    c = ((xx * cells // width) + (yy * cells // height)) % 2
    return np.where(c[..., None] == 1, np.array([0.85, 0.8, 0.2]), np.array([0.15, 0.25, 0.7])).astype(np.float32)


def textured_scene(width=48, height=48, spp=4, sppe=0, sppse=0, texture=None, param=None, env=True, box=True):
    """A uv-mapped floor quad (uv = (x, z)/560) with a textured DiffuseBSDF, the small box on it, under a constant
    environment map (env=True) or the Cornell luminaire.  param: 'texture' (d texel / dP = 1), 'box_x', None."""
    tex = texture if texture is not None else ramp_texture()
    v = np.array([[0, 0, 0], [560, 0, 0], [560, 0, 560], [0, 0, 560]], dtype=np.float32)
    f = np.array([[0, 2, 1], [0, 3, 2]], dtype=np.int32)             # normal +y
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32)
    floor = MeshSpec(vertices=v, faces=f, uvs=uv, face_uvs=f.copy(), bsdf=0, emitter=-1)
    bsdfs = [BsdfSpec((0.5, 0.5, 0.5), name="tex", texture=tex), BsdfSpec((0.5, 0.5, 0.5), name="cat"), BsdfSpec((0.0, 0.0, 0.0), name="light")]
    meshes = [floor]
    if box:
        meshes.append(_mesh("cbox_smallbox.obj", 1))
    if env:
        emitters = [EmitterSpec(type=1, env_data=synthetic_envmap(64, 32, sun=False), env_scale=1.0)]
    else:
        emitters = [EmitterSpec((20.0, 20.0, 8.0))]
        meshes.append(_mesh("cbox_luminaire.obj", 2, emitter=0, raw=translate(0.0, -100.0, 0.0)))
    cam = CameraSpec(60.0, 0.000001, 10000000.0, to_world_raw=translate(278.0, 400.0, -700.0) @ _rot_x(np.radians(25.0)))
    if param == "texture":
        bsdfs[0].d_texture = np.ones_like(tex)
    elif param == "box_x" and box:
        dT = np.zeros((4, 4), dtype=np.float32)
        dT[0, 3] = 100.0
        meshes[1].d_to_world_left = dT
    elif param is not None:
        raise ValueError(param)
    return SceneSpec(meshes, bsdfs, emitters, [cam], width, height, spp, sppe, sppse)


def _scale_m(s):
    m = np.eye(4, dtype=np.float32)
    m[0, 0] = m[1, 1] = m[2, 2] = s
    return m


def ortho_cbox_scene(width=48, height=48, spp=4, sppe=0, sppse=0, param="box_x"):
    """The README Cornell box shrunk into the orthographic view volume ([-1, 1] x [-1/aspect, 1/aspect] camera units:
    sensor transforms may not scale, reference sensor.cpp:7-14) and seen through OrthographicCamera(near, far)."""
    spec = cbox_scene(width, height, spp, sppe, sppse, param=None)
    S = _scale_m(1.0 / 300.0) @ translate(-278.0, -273.0, -280.0)
    for m in spec.meshes:
        m.to_world_raw = (S @ np.asarray(m.to_world_raw, np.float32)).astype(np.float32)
    spec.cameras[0] = CameraSpec(0.0, 0.1, 100.0, to_world_raw=translate(0.0, 0.0, -5.0), orthographic=True)
    dT = np.zeros((4, 4), dtype=np.float32)
    dT[0, 3] = 0.3
    if param == "box_x":
        spec.meshes[1].d_to_world_left = dT
    elif param == "light_x":
        spec.meshes[0].d_to_world_left = dT
    elif param == "camera_x":
        spec.cameras[0].d_to_world_left = dT
    elif param is not None:
        raise ValueError(param)
    return spec


def microfacet_cbox_scene(width=48, height=48, spp=8, sppe=0, sppse=0, param="roughness", two_sided=False):
    """The README Cornell box with Microfacet boxes and floor (reference src/bsdf/microfacet.cpp).
    param: 'roughness' | 'specular' | 'diffuse' (d/dP = 1 on the boxes' BSDF) | 'box_x' | None"""
    spec = cbox_scene(width, height, spp, sppe, sppse, param=None)
    spec.bsdfs[1] = BsdfSpec((0.3, 0.4, 0.5), name="cat", type=1, specular=(0.6, 0.5, 0.4), roughness=0.35, two_sided=two_sided)
    spec.bsdfs[2] = BsdfSpec((0.7, 0.7, 0.7), name="white", type=1, specular=(0.04, 0.04, 0.04), roughness=0.6)
    if param == "roughness":
        spec.bsdfs[1].d_roughness = 1.0
    elif param == "specular":
        spec.bsdfs[1].d_specular = (1.0, 1.0, 1.0)
    elif param == "diffuse":
        spec.bsdfs[1].d_reflectance = (1.0, 1.0, 1.0)
    elif param == "box_x":
        dT = np.zeros((4, 4), dtype=np.float32)
        dT[0, 3] = 100.0
        spec.meshes[1].d_to_world_left = dT
    elif param is not None:
        raise ValueError(param)
    return spec


def conductor_cbox_scene(width=48, height=48, spp=8, sppe=0, sppse=0, param="alpha"):
    """The README Cornell box with a gold-like anisotropic RoughConductor small box and an isotropic copper-like tall box
    (reference src/bsdf/roughconductor.cpp).  param: 'alpha' | 'eta' | 'k' | 'box_x' | None (on the small box)"""
    spec = cbox_scene(width, height, spp, sppe, sppse, param=None)
    spec.bsdfs.append(BsdfSpec(name="gold", type=2, alpha_u=0.15, alpha_v=0.35, eta=(0.14, 0.37, 1.44), k=(3.98, 2.38, 1.60), specular=(1.0, 0.9, 0.8)))
    spec.bsdfs.append(BsdfSpec(name="copper", type=2, alpha_u=0.3, alpha_v=0.3, eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), specular=(1.0, 1.0, 1.0), two_sided=True))
    spec.meshes[1].bsdf = 5
    spec.meshes[2].bsdf = 6
    g = spec.bsdfs[5]
    if param == "alpha":
        g.d_alpha_u, g.d_alpha_v = 1.0, 0.5
    elif param == "eta":
        g.d_eta = (1.0, 1.0, 1.0)
    elif param == "k":
        g.d_k = (1.0, 0.5, 0.25); g.d_specular = (0.0, 1.0, 0.0)
    elif param == "box_x":
        dT = np.zeros((4, 4), dtype=np.float32)
        dT[0, 3] = 100.0
        spec.meshes[1].d_to_world_left = dT
    elif param is not None:
        raise ValueError(param)
    return spec


def dielectric_cbox_scene(width=48, height=48, spp=8, sppe=0, sppse=0, param="alpha"):
    """The README Cornell box with a rough-glass small box (one-sided: rays refract in and out through the closed mesh) and a
    two-sided frosted tall box (reference src/bsdf/roughdielectric.cpp).  param: 'alpha' | 'eta' | 'box_x' | None (small box)"""
    spec = cbox_scene(width, height, spp, sppe, sppse, param=None)
    e1, e2 = np.float32(1.5), np.float32(1.33)
    spec.bsdfs.append(BsdfSpec(name="glass", type=3, alpha_u=0.12, alpha_v=0.12, eta=(float(e1), float(np.float32(1.0) / e1), 0.0)))
    spec.bsdfs.append(BsdfSpec(name="frosted", type=3, alpha_u=0.4, alpha_v=0.4, eta=(float(e2), float(np.float32(1.0) / e2), 0.0), two_sided=True))
    spec.meshes[1].bsdf = 5
    spec.meshes[2].bsdf = 6
    g = spec.bsdfs[5]
    if param == "alpha":
        g.d_alpha_u, g.d_alpha_v = 1.0, 1.0
    elif param == "eta":
        g.d_eta = (1.0, float(-np.float32(1.0) / (e1 * e1)), 0.0)
    elif param == "box_x":
        dT = np.zeros((4, 4), dtype=np.float32)
        dT[0, 3] = 100.0
        spec.meshes[1].d_to_world_left = dT
    elif param is not None:
        raise ValueError(param)
    return spec


def textured_microfacet_scene(width=48, height=48, spp=4, sppe=0, sppse=0, param=None):
    """textured_scene (Cornell luminaire) whose floor is a MicrofacetBSDF with all three parameters as bitmaps of different
    resolutions (reference microfacet.cpp:38-45).  param: 'diffuse' | 'specular' | 'roughness' (d texel / dP = 1) | 'box_x' | None"""
    spec = textured_scene(width, height, spp, sppe, sppse, texture=checker_texture(8, 8), param="box_x" if param == "box_x" else None, env=False)
    rng = np.random.default_rng(5)
    b = spec.bsdfs[0]
    b.type = 1
    b.spec_texture = (0.2 + 0.7 * rng.random((6, 5, 3))).astype(np.float32)
    yy, xx = np.meshgrid(np.linspace(0, 1, 7, dtype=np.float32), np.linspace(0, 1, 9, dtype=np.float32), indexing="ij")
    b.rough_texture = (0.15 + 0.5 * (0.5 + 0.5 * np.sin(5 * xx) * np.cos(4 * yy))).astype(np.float32)
    if param == "diffuse":
        b.d_texture = np.ones_like(b.texture)
    elif param == "specular":
        b.d_spec_texture = np.ones_like(b.spec_texture)
    elif param == "roughness":
        b.d_rough_texture = np.ones_like(b.rough_texture)
    elif param not in (None, "box_x"):
        raise ValueError(param)
    return spec


def envmap_tutorial_scene(width=48, height=48, spp=4, sppe=0, sppse=0, param="bunny_x", env_stride=8, schlick_g=False):
    """The reference's tutorials/Forward_AD_envmap.ipynb scene: bunny_low.obj (4968 triangles) with
    MicrofacetBSDF([0.2, 0.9, 0.9], [0.01, 0.01, 0.01], 0.3), translated to z = -100, lit by ballroom_1k.exr (the tutorial's own
    PIZ-compressed environment map, read with psdr_jit_amd.exr and decimated by env_stride to keep the oracle quick), camera
    fov 80 looking down -z.  param: 'bunny_x' (Mesh[0].to_world_left = T(100 P, 0, 0), as in the notebook) | 'roughness' | None"""
    from psdr_jit_amd import exr
    root = os.path.dirname(os.path.dirname(DATA))
    env = exr.read_rgb(os.path.join(root, "data", "envmap", "ballroom_1k.exr"))
    env = np.ascontiguousarray(env[::env_stride, ::env_stride])
    path = os.path.join(root, "data", "mesh", "bunny_low.obj")
    v, f, uv, fuv = load_obj(path)
    bunny = MeshSpec(vertices=v, faces=f, uvs=uv, face_uvs=fuv, bsdf=0, emitter=-1, path=path)
    bunny.to_world_raw = translate(0.0, 0.0, -100.0)
    bsdfs = [BsdfSpec((0.01, 0.01, 0.01), name="bunny", type=1, specular=(0.2, 0.9, 0.9), roughness=0.3)]
    cam_m = np.diag([-1.0, 1.0, -1.0, 1.0]).astype(np.float32)
    cam = CameraSpec(80.0, 0.000001, 10000000.0, to_world_raw=cam_m)
    emitters = [EmitterSpec(type=1, env_data=env, env_scale=1.0)]
    if param == "bunny_x":
        dT = np.zeros((4, 4), dtype=np.float32)
        dT[0, 3] = 100.0
        bunny.d_to_world_left = dT
    elif param == "roughness":
        bsdfs[0].d_roughness = 1.0
    elif param is not None:
        raise ValueError(param)
    if schlick_g:
        # the SAME parameters through MicrofacetBSDFPerVertex (reference src/bsdf/microfacet_pv.cpp:49-63): equal values on every
        # vertex make it the Microfacet BSDF with the Schlick-k geometry term G1(c) = c / (c (1 - k) + k), k = (roughness + 1)^2 / 8
        # in place of microfacet.cpp:50's Smith term - the only difference between the two classes' eval
        b, n = bsdfs[0], len(v)
        b.type = 4
        b.pv_specular = np.tile(np.float32(b.specular), (n, 1))
        b.pv_diffuse = np.tile(np.float32(b.reflectance), (n, 1))
        b.pv_roughness = np.full(n, b.roughness, np.float32)
    return SceneSpec([bunny], bsdfs, emitters, [cam], width, height, spp, sppe, sppse)


def pervertex_scene(width=48, height=48, spp=4, sppe=0, sppse=0, param=None):
    """The Cornell box whose large sphere-less furniture is replaced by the tutorial's small sphere (cbox_smallball.obj, smooth
    normals) with a MicrofacetBSDFPerVertex (reference src/bsdf/microfacet_pv.cpp): seeded per-vertex specular / diffuse /
    roughness.  param: 'diffuse' | 'specular' | 'roughness' (d value / dP = 1 on every vertex) | 'ball_x' | None"""
    spec = cbox_scene(width, height, spp, sppe, sppse, param=None)
    ball = _mesh("cbox_smallball.obj", 5)
    spec.meshes[1] = ball                      # replaces the small box
    n = len(ball.vertices)
    rng = np.random.default_rng(9)
    b = BsdfSpec(name="pv", type=4)
    b.pv_specular = (0.2 + 0.6 * rng.random((n, 3))).astype(np.float32)
    b.pv_diffuse = (0.1 + 0.5 * rng.random((n, 3))).astype(np.float32)
    b.pv_roughness = (0.2 + 0.5 * rng.random(n)).astype(np.float32)
    spec.bsdfs.append(b)
    if param == "diffuse":
        b.d_pv_diffuse = np.ones_like(b.pv_diffuse)
    elif param == "specular":
        b.d_pv_specular = np.ones_like(b.pv_specular)
    elif param == "roughness":
        b.d_pv_roughness = np.ones_like(b.pv_roughness)
    elif param == "ball_x":
        dT = np.zeros((4, 4), dtype=np.float32)
        dT[0, 3] = 100.0
        ball.d_to_world_left = dT
    elif param is not None:
        raise ValueError(param)
    return spec


def bumpy_normal_map(width=16, height=16, amp=0.35):
    """A smooth normal map (rgb = 0.5 + 0.5 n, n the unit normal of a sine bump field), bitmap.cpp conventions"""
    yy, xx = np.meshgrid(np.linspace(0, 1, height, dtype=np.float32), np.linspace(0, 1, width, dtype=np.float32), indexing="ij")
    nx, ny = amp * np.cos(7 * xx) * np.sin(5 * yy), amp * np.sin(6 * xx + 1.0) * np.cos(4 * yy)
    n = np.stack([nx, ny, np.ones_like(nx)], axis=-1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    return (0.5 + 0.5 * n).astype(np.float32)


def normalmap_scene(width=48, height=48, spp=4, sppe=0, sppse=0, param=None, nested="microfacet", nmap="bumpy"):
    """textured_scene (Cornell luminaire) whose uv-mapped floor is a NormalMapBSDF (reference src/bsdf/normalmap.cpp) over a
    Microfacet / Diffuse / RoughConductor BSDF.  nmap: 'bumpy' (16x16 map) | 'flat' (the constant (.499999, .499999, 1) the reference's
    add_BSDF uses) | 'tilted' (a constant).  param: 'nmap' (d texel / dP = (1, 0.5, 0)) | 'nested' (d nested diffuse colour) | 'box_x' | None"""
    spec = textured_scene(width, height, spp, sppe, sppse, texture=checker_texture(8, 8), param="box_x" if param == "box_x" else None, env=False)
    inner = {"microfacet": BsdfSpec((0.3, 0.25, 0.2), name="inner", type=1, specular=(0.7, 0.6, 0.5), roughness=0.35),
             "diffuse": BsdfSpec((0.6, 0.5, 0.4), name="inner"),
             "roughconductor": BsdfSpec(name="inner", type=2, alpha_u=0.2, alpha_v=0.3, eta=(0.2, 0.9, 1.1), k=(3.9, 2.4, 2.1), specular=(1.0, 0.9, 0.8))}[nested]
    spec.bsdfs.append(inner)
    b = spec.bsdfs[0]
    b.type, b.nested, b.texture = 5, len(spec.bsdfs) - 1, None
    if nmap == "bumpy":
        b.texture = bumpy_normal_map()
    elif nmap == "flat":
        b.reflectance = (0.499999, 0.499999, 1.0)
    else:
        b.reflectance = (0.62, 0.43, 0.93)
    if param == "nmap":
        if b.texture is not None:
            d = np.zeros_like(b.texture); d[..., 0], d[..., 1] = 1.0, 0.5
            b.d_texture = d
        else:
            b.d_reflectance = (1.0, 0.5, 0.0)
    elif param == "nested":
        inner.d_reflectance = (1.0, 1.0, 1.0)
    elif param not in (None, "box_x"):
        raise ValueError(param)
    return spec


def textured_ggx_scene(width=48, height=48, spp=4, sppe=0, sppse=0, kind="roughconductor", param=None):
    """textured_scene (Cornell luminaire) whose uv-mapped floor is a RoughConductor with eta / k / alpha bitmaps or a two-sided
    RoughDielectric with an alpha bitmap (reference roughconductor.cpp:37-43, roughdielectric.cpp:75-78).
    param: 'eta' | 'k' | 'alpha' (d texel / dP = 1) | 'box_x' | None"""
    spec = textured_scene(width, height, spp, sppe, sppse, texture=checker_texture(8, 8), param="box_x" if param == "box_x" else None, env=False)
    rng = np.random.default_rng(21)
    b = spec.bsdfs[0]
    yy, xx = np.meshgrid(np.linspace(0, 1, 6, dtype=np.float32), np.linspace(0, 1, 7, dtype=np.float32), indexing="ij")
    alpha = (0.12 + 0.3 * (0.5 + 0.5 * np.sin(6 * xx) * np.cos(5 * yy))).astype(np.float32)
    if kind == "roughconductor":
        b.type, b.specular = 2, (1.0, 0.95, 0.9)
        b.texture = (0.15 + 1.2 * rng.random((5, 4, 3))).astype(np.float32)          # eta map
        b.spec_texture = (2.0 + 2.0 * rng.random((4, 6, 3))).astype(np.float32)      # k map
        b.rough_texture = alpha
    else:
        b.type, b.texture, b.two_sided = 3, None, True
        e = np.float32(1.5)
        b.eta = (float(e), float(np.float32(1.0) / e), 0.0)
        b.rough_texture = alpha
    if param == "eta":
        b.d_texture = np.ones_like(b.texture)
    elif param == "k":
        b.d_spec_texture = np.ones_like(b.spec_texture)
    elif param == "alpha":
        b.d_rough_texture = np.ones_like(b.rough_texture)
    elif param not in (None, "box_x"):
        raise ValueError(param)
    return spec


def hdr_step_square_scene(width=64, height=64, spp=0, sppe=128, a=20.0, d=100.0, fov=60.0, L_pos=100.0, L_neg=1.5):
    """A black, flat-shaded square (half-size a, depth d, facing the camera at the origin that looks down -z) moving along x by
    100 P in front of a lat-long map that is L_pos where the direction has x > 0 and L_neg where x < 0: the closed-form case of the
    primary-edge term (tests/test_oracle_envmap.py).  -> (spec, W H (2 a k)(100 k)) with k = 1 / (2 d tan(fov / 2))"""
    v = np.float32([[-a, -a, -d], [a, -a, -d], [a, a, -d], [-a, a, -d]])
    f = np.int32([[0, 1, 2], [0, 2, 3]])
    quad = MeshSpec(vertices=v, faces=f, bsdf=0, emitter=-1)
    quad.use_face_normals = True
    dT = np.zeros((4, 4), np.float32)
    dT[0, 3] = 100.0
    quad.d_to_world_left = dT
    cam = CameraSpec(fov, 1e-6, 1e7, to_world_raw=np.diag([-1.0, 1.0, -1.0, 1.0]).astype(np.float32))
    w, h = 64, 32
    env = np.empty((h, w, 3), np.float32)
    env[:, : w // 2] = L_pos                  # u = atan2(x, -z) / 2 pi in [0, 1/2): x > 0
    env[:, w // 2:] = L_neg
    spec = SceneSpec([quad], [BsdfSpec((0.0, 0.0, 0.0), name="black")], [EmitterSpec(type=1, env_data=env, env_scale=1.0)], [cam],
                     width, height, spp, sppe, 0)
    k = 1.0 / (2.0 * d * np.tan(np.radians(fov) / 2.0))
    return spec, width * height * (2 * a * k) * (100.0 * k)
