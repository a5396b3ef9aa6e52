"""Synthetic inputs that need no file: the ~100k-triangle mesh and the lat-long map of BASELINE config 5, and that
configuration's scene on the package's public surface.  tests/scenes.py describes the same scene for the oracle from the same
two generators, so bench.py's config-5 leg and the parity tests see identical inputs."""
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def icosphere(level=3, radius=1.0, noise=0.0, seed=0):
    """Icosphere with 20*4^level triangles; `noise` scales a seeded radial perturbation (BASELINE config 5's
    ~100k-triangle mesh: level 6 = 81920 triangles, numpy.random.default_rng(0))."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]]
    f = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
         [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]]
    v = [np.asarray(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = nf
    v = np.asarray(v)
    if noise > 0:
        v = v * (1.0 + noise * np.random.default_rng(seed).standard_normal(len(v)))[:, None]
    return (v * radius).astype(np.float32), np.asarray(f, dtype=np.int32)


def synthetic_envmap(width=64, height=32, sun=True):
    """Lat-long radiance image [H, W, 3] by a fixed formula (BASELINE config 5 / SURVEY §8d): a constant sky
    (0.6, 0.7, 0.9) plus, when `sun`, one Gaussian sun of peak (40, 36, 30) at (u, v) = (0.30, 0.25), sigma 0.04."""
    u = (np.arange(width, dtype=np.float64) + 0.5) / width
    v = (np.arange(height, dtype=np.float64) + 0.5) / height
    uu, vv = np.meshgrid(u, v)
    img = np.empty((height, width, 3), dtype=np.float64)
    img[...] = (0.6, 0.7, 0.9)
    if sun:
        du = np.minimum(np.abs(uu - 0.30), 1.0 - np.abs(uu - 0.30))
        g = np.exp(-(du * du + (vv - 0.25) ** 2) / (2 * 0.04 ** 2))
        img += g[..., None] * np.array([40.0, 36.0, 30.0])
    return img.astype(np.float32)


def config5_scene(psdr, res=1024, spp=64, level=6, env_res=(1024, 512)):
    """BASELINE config 5 through the public API: the noisy icosphere (81 920 triangles at level 6) over the Cornell floor under the
    synthetic map, camera fov 60 at (278, 400, -700) pitched 25 degrees; the parameter is the blob's DiffuseBSDF albedo.
    -> (scene, albedo leaf)"""
    from psdr_jit_amd import Matrix4fC, Matrix4fD
    sc = psdr.Scene()
    sc.opts.spp = sc.opts.sppe = sc.opts.sppse = spp
    sc.opts.width = sc.opts.height = res
    sc.opts.log_level = 0
    a = np.radians(25.0)
    c, s = float(np.cos(a)), float(np.sin(a))
    sensor = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    sensor.to_world = Matrix4fD([[1., 0., 0., 278.], [0., c, -s, 400.], [0., s, c, -700.], [0., 0., 0., 1.]])
    sc.add_Sensor(sensor)
    sc.add_BSDF(psdr.DiffuseBSDF([0.5, 0.5, 0.5]), "blob")
    sc.add_BSDF(psdr.DiffuseBSDF([0.8, 0.8, 0.8]), "white")
    v, f = icosphere(level, radius=150.0, noise=0.01, seed=0)
    blob = psdr.Mesh()
    blob.load_raw(v, f, np.zeros((0, 2), np.float32), np.zeros((0, 3), np.int32))
    blob.to_world = Matrix4fC([[1., 0., 0., 278.], [0., 1., 0., 160.], [0., 0., 1., 280.], [0., 0., 0., 1.]])
    sc.add_Mesh(blob, "blob", None)
    sc.add_Mesh(write_cornell_box()["floor"], Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]), "white", None)
    sc.add_EnvironmentMap(psdr.EnvironmentMap(synthetic_envmap(env_res[0], env_res[1])))
    sc.configure()
    sc.configure([0])
    return sc, sc.param_map["BSDF[0]"].reflectance


# ---------------------------------------------------------------------------------------------------------------------------
# The README's Cornell box from coordinates: the eight meshes of the reference README example as literal vertex / texture-
# coordinate / face lists - the classic Cornell-box measurements (552.8 x 548.8 x 559.2 room, the two blocks, the 130 x 105 luminaire)
# in the decimal spelling psdr-jit's tutorial data uses, so that float parsing gives the same bits.  bench.py and tests/scenes.py build
# the C3 / C4 workload from write_cornell_box(), not from files shipped with the reference.
CORNELL_BOX = {
    "luminaire": {
        "v": ["343 540.79999 227", "343 540.79999 332", "213 540.79999 332", "213 540.79999 227"],
        "f": ["1 2 3 4"],
    },
    "smallbox": {
        "v": ["130.000000 165.000000 65.000000", "82.000000 165.000000 225.000000", "240.000000 165.000000 272.000000", "290.000000 165.000000 114.000000", "290.000000 0.000000 114.000000", "240.000000 0.000000 272.000000", "130.000000 0.000000 65.000000", "82.000000 0.000000 225.000000"],
        "f": ["1 2 3", "1 3 4", "5 4 3", "5 3 6", "7 1 4", "7 4 5", "8 2 1", "8 1 7", "6 3 2", "6 2 8", "5 6 8", "5 8 7"],
    },
    "largebox": {
        "v": ["423.000000 329.999969 247.000061", "265.000000 329.999939 296.000061", "314.000000 329.999939 456.000061", "472.000000 329.999939 406.000061", "423.000000 -0.000040 247.000000", "472.000000 -0.000066 406.000000", "314.000000 -0.000074 456.000000", "265.000000 -0.000048 296.000000"],
        "vt": ["0.994838 0.753967", "0.663615 0.753540", "0.663030 0.500000", "0.994838 0.501892", "0.335054 0.500000", "0.335054 0.000000", "0.668172 0.000000", "0.668172 0.500000", "1.000000 0.000000", "1.000000 0.500000", "0.335054 0.000000", "0.335054 0.500000", "0.000000 0.500000", "0.000000 0.000000", "0.000000 1.000000", "0.331223 0.500000", "0.331223 1.000000", "0.663030 0.752075", "0.331223 0.753998", "0.331808 0.500458", "0.663030 0.500000"],
        "f": ["1/1 2/2 3/3", "1/1 3/3 4/4", "5/5 1/6 4/7", "5/5 4/7 6/8", "6/8 4/7 3/9", "6/8 3/9 7/10", "7/11 3/12 2/13", "7/11 2/13 8/14", "8/15 2/13 1/16", "8/15 1/16 5/17", "6/18 7/19 8/20", "6/18 8/20 5/21"],
    },
    "floor": {
        "v": ["552.79999 0 0", "0 0 0", "0 0 559.20001", "549.59998 0 559.20001", "130 0 65", "82 0 225", "240 0 272", "290 0 114", "423 0 247", "265 0 296", "314 0 456", "472 0 406"],
        "f": ["1 2 3 4"],
    },
    "ceiling": {
        "v": ["556 548.79999 0", "556 548.79999 559.20001", "0 548.79999 559.20001", "0 548.79999 0"],
        "f": ["1 2 3 4"],
    },
    "back": {
        "v": ["549.599976 -0.000091 559.200012", "0.000000 -0.000091 559.200012", "0.000000 548.799927 559.200073", "556.000000 548.799927 559.200073"],
        "vt": ["0.987061 0.011536", "0.987061 1.000000", "0.000000 1.000000", "0.000000 0.000000"],
        "f": ["1/1 2/2 3/3 4/4"],
    },
    "greenwall": {
        "v": ["0 0 559.20001", "0 0 0", "0 548.79999 0", "0 548.79999 559.20001"],
        "f": ["1 2 3 4"],
    },
    "redwall": {
        "v": ["552.79999 0 0", "549.59998 0 559.20001", "556 548.79999 559.20001", "556 548.79999 0"],
        "f": ["1 2 3 4"],
    },
}


def cornell_box_obj(name):
    """Wavefront OBJ text of one mesh of the README scene (luminaire, smallbox, largebox, floor, ceiling, back, greenwall, redwall)"""
    m = CORNELL_BOX[name]
    lines = ["o cbox_%s" % name] + ["v " + s for s in m["v"]] + ["vt " + s for s in m.get("vt", [])] + ["f " + s for s in m["f"]]
    return "\n".join(lines) + "\n"


_cbox_dir = None


def write_cornell_box(dirpath=None):
    """Writes the eight meshes as cbox_<name>.obj (into a fresh temporary directory by default, once per process) -> {name: path}"""
    global _cbox_dir
    import tempfile
    if dirpath is None:
        if _cbox_dir is None or not os.path.isdir(_cbox_dir):
            _cbox_dir = tempfile.mkdtemp(prefix="psdr_cbox_")
        dirpath = _cbox_dir
    os.makedirs(dirpath, exist_ok=True)
    paths = {}
    for name in CORNELL_BOX:
        paths[name] = os.path.join(dirpath, "cbox_%s.obj" % name)
        if not os.path.exists(paths[name]):
            with open(paths[name], "w") as fh:
                fh.write(cornell_box_obj(name))
    return paths
