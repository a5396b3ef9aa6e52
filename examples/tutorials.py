"""The five notebooks of the reference's tutorials/ directory (Forward_AD, Forward_AD_envmap, batch_render,
different_integrator, secondary_edge_guiding) as plain functions on psdr_jit_amd.

What changes against the notebooks: `import psdr_jit_amd as psdr`; drjit arrays become torch tensors (FloatD(0.) is a leaf
tensor, Matrix4fD a 4x4 tensor expression); `drjit.set_grad(P, 1); drjit.forward_to(img); drjit.grad(img)` becomes
`psdr.forward_grad(img, P)`, `drjit.backward(loss)` becomes `loss.backward()`.  Everything else is the notebooks' own calls.
Each function takes the resolution / sample counts so that tests can run it small; the defaults are the notebooks' values.

    python examples/tutorials.py forward_ad            # writes forward_ad.npy / forward_ad_grad.npy
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import psdr_jit_amd as psdr                                                    # noqa: E402
from psdr_jit_amd import FloatD, Matrix4fC, Matrix4fD                          # noqa: E402

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
I4 = [[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]


def _camera(sc, x=278., y=273., z=-500., fov=60):
    sensor = psdr.PerspectiveCamera(fov, 0.000001, 10000000.)
    sensor.to_world = Matrix4fD([[1., 0., 0., x], [0., 1., 0., y], [0., 0., 1., z], [0., 0., 0., 1.]])
    sc.add_Sensor(sensor)


def _sphere_box(sc, large_bsdf=None):
    """the Cornell box with the two spheres every notebook but the envmap one builds (660 triangles)"""
    sc.add_BSDF(large_bsdf if large_bsdf is not None else psdr.DiffuseBSDF([0.2, 0.9, 0.9]), "sphere_large")
    sc.add_BSDF(psdr.DiffuseBSDF(0.5), "back")
    sc.add_BSDF(psdr.DiffuseBSDF(0.5), "light")
    sc.add_BSDF(psdr.DiffuseBSDF([0.9, 0.6, 0.1]), "sphere_small")
    sc.add_BSDF(psdr.DiffuseBSDF([0.95, 0.95, 0.95]), "white")
    sc.add_BSDF(psdr.DiffuseBSDF([0.2, 0.9, 0.2]), "green")
    sc.add_BSDF(psdr.DiffuseBSDF([0.9, 0.2, 0.2]), "red")
    cb = os.path.join(DATA, "cbox")
    sc.add_Mesh(os.path.join(cb, "cbox_luminaire.obj"), Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., -0.5], [0., 0., 1., 0.], [0., 0., 0., 1.]]), "light",
                psdr.AreaLight([20.0, 20.0, 8.0]))
    for f, b in (("smallball", "sphere_small"), ("largeball", "sphere_large"), ("floor", "white"), ("ceiling", "white"), ("back", "back"),
                 ("greenwall", "green"), ("redwall", "red")):
        sc.add_Mesh(os.path.join(cb, "cbox_%s.obj" % f), Matrix4fC(I4), b, None)


def _scene(width, height, spp, sppe, sppse):
    sc = psdr.Scene()
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse = spp, sppe, sppse
    sc.opts.height, sc.opts.width = height, width
    sc.opts.log_level = 0
    return sc


def _move(sc, names):
    P = FloatD(0.).requires_grad_()
    for n in names:
        sc.param_map[n].set_transform(Matrix4fD([[1., 0., 0., P * 100], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
    sc.configure([0])
    return P


def forward_ad(width=512, height=512, spp=32, sppe=32, sppse=32, depth=3):
    """Forward_AD.ipynb: d image / d(x-translation of the luminaire and the small sphere), all three terms"""
    sc = _scene(width, height, spp, sppe, sppse)
    integrator = psdr.PathTracer(depth)
    _camera(sc)
    _sphere_box(sc)
    sc.configure()
    sc.configure([0])
    P = _move(sc, ["Mesh[0]", "Mesh[1]"])
    img = integrator.renderD(sc, 0)
    return img.detach(), psdr.forward_grad(img, P)


def forward_ad_envmap(width=128, height=128, spp=128, term="interior", per_vertex=False):
    """Forward_AD_envmap.ipynb: the bunny under ballroom_1k.exr, one term of the derivative per cell.  per_vertex=True puts the
    notebook's three parameters on every vertex of a MicrofacetBSDFPerVertex instead - the same BSDF with the Schlick-k geometry
    term (reference src/bsdf/microfacet_pv.cpp:49-63) the notebook's figures were rendered with (tests/test_oracle_notebooks.py)"""
    n = {"interior": (spp, 0, 0), "primary": (0, spp, 0), "secondary": (0, 0, spp)}[term]
    sc = _scene(width, height, *n)
    integrator = psdr.PathTracer(1)
    sensor = psdr.PerspectiveCamera(80, 0.000001, 10000000.)
    sensor.to_world = Matrix4fD([[-1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., -1., 0.], [0., 0., 0., 1.]])
    sc.add_Sensor(sensor)
    bunny = os.path.join(DATA, "mesh", "bunny_low.obj")
    if per_vertex:
        n = sum(1 for line in open(bunny) if line.startswith("v "))
        sc.add_BSDF(psdr.MicrofacetBSDFPerVertex(np.tile(np.float32([0.2, 0.9, 0.9]), (n, 1)), np.tile(np.float32([0.01, 0.01, 0.01]), (n, 1)),
                                                 np.full(n, 0.3, np.float32)), "bunny")
    else:
        sc.add_BSDF(psdr.MicrofacetBSDF([0.2, 0.9, 0.9], [0.01, 0.01, 0.01], 0.3), "bunny")
    sc.add_Mesh(bunny, Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., -100.], [0., 0., 0., 1.]]), "bunny", None)
    sc.add_EnvironmentMap(os.path.join(DATA, "envmap", "ballroom_1k.exr"), I4, 1.0)
    sc.configure()
    sc.configure([0])
    P = _move(sc, ["Mesh[0]"])
    img = integrator.renderD(sc, 0)
    return img.detach(), psdr.forward_grad(img, P)


def batch_render(width=400, height=300, spp=32, crop=((150, 100), (250, 200))):
    """batch_render.ipynb: a RoughConductor sphere, the full frame and a crop through batch_pix"""
    sc = _scene(width, height, spp, 0, 0)
    integrator = psdr.PathTracer(2)
    _camera(sc)
    alpha = psdr.Bitmap1fD(0.01)
    eta = psdr.Bitmap3fD([0.155475, 0.116753, 0.138334])
    k = psdr.Bitmap3fD([4.83181, 3.12296, 2.1486])
    _sphere_box(sc, psdr.RoughConductorBSDF(alpha, eta, k))
    sc.configure()
    full = integrator.renderC(sc, 0)
    (y0, x0), (y1, x1) = crop
    pix_id = np.arange(height * width).reshape(height, width)[y0:y1, x0:x1].reshape(-1)
    part = integrator.renderC(sc, 0, seed=0, batch_pix=pix_id)
    return full, part, pix_id


def different_integrator(name="silhouette 1", width=512, height=512, spp=32, sppe=32, sppse=32):
    """different_integrator.ipynb: PathTracer(1) | CollocatedIntegrator(1000000) | FieldExtractionIntegrator(...)"""
    sc = _scene(width, height, spp, sppe, sppse)
    if name == "path":
        integrator = psdr.PathTracer(1)
    elif name == "collocated":
        integrator = psdr.CollocatedIntegrator(1000000)
    else:
        integrator = psdr.FieldExtractionIntegrator(name)
    _camera(sc)
    _sphere_box(sc)
    sc.configure()
    sc.configure([0])
    P = _move(sc, ["Mesh[0]", "Mesh[1]"])
    img = integrator.renderD(sc, 0)
    return img.detach(), psdr.forward_grad(img, P)


def secondary_edge_guiding(width=512, height=512, sppse=4, guide=(2000, 5, 5, 32)):
    """secondary_edge_guiding.ipynb: the secondary-edge term without and with the guiding grid"""
    sc = _scene(width, height, 0, 0, sppse)
    integrator = psdr.PathTracer(1)
    _camera(sc)
    _sphere_box(sc)
    sc.configure()
    sc.configure([0])
    P = _move(sc, ["Mesh[0]", "Mesh[1]"])
    img = integrator.renderD(sc, 0)
    plain = psdr.forward_grad(img, P)
    P = _move(sc, ["Mesh[0]", "Mesh[1]"])
    integrator.preprocess_secondary_edges(sc, 0, list(guide), 1)
    img = integrator.renderD(sc, 0)
    return plain, psdr.forward_grad(img, P)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "forward_ad"
    out = globals()[which]()
    for i, t in enumerate(out):
        np.save("%s_%d.npy" % (which, i), t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t))
    print(which, [tuple(t.shape) for t in out])
