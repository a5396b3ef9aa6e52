"""The reference's five tutorial notebooks, ported call for call in examples/tutorials.py, run small on the GPU: the API
surface a psdr-jit user touches is there, and the qualitative facts the notebooks' figures show hold."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


@pytest.fixture(scope="module")
def tut():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists)")
    import __graft_entry__
    __graft_entry__.build()
    import tutorials
    return tutorials


def test_forward_ad(tut):
    img, d = tut.forward_ad(96, 96, 16, 16, 16)
    img = img.cpu().numpy().reshape(96, 96, 3); d = d.cpu().numpy().reshape(96, 96, 3).mean(axis=2)
    assert np.isfinite(img).all() and np.isfinite(d).all()
    iy, ix = np.unravel_index(img[..., 0].argmax(), (96, 96))
    assert iy < 32 and 32 < ix < 64 and np.allclose(img[iy, ix], [20.0, 20.0, 8.0], rtol=0.05)     # the luminaire, top centre
    left, right = img[30:70, 2:10].mean(axis=(0, 1)), img[30:70, 86:94].mean(axis=(0, 1))
    assert left[0] > 1.5 * left[1] and right[1] > 1.5 * right[0]  # red wall on the left, green on the right
    # the light and the small sphere move in +x: the silhouette of the sphere and the shadow edges carry the derivative
    assert np.abs(d).max() > 0.05 and d[:, :48].sum() > 0 > d[:, 48:].sum()      # brighter on the left, darker on the right


def test_forward_ad_envmap(tut):
    tot = 0.0
    for term in ("interior", "primary", "secondary"):
        img, d = tut.forward_ad_envmap(48, 48, 16, term)
        assert np.isfinite(d.cpu().numpy()).all()
        tot += float(d.abs().sum())
        if term == "interior":
            a = img.cpu().numpy().reshape(48, 48, 3)
            assert a.mean() > 0.05 and a[20:28, 20:28].std() > 0      # the bunny in front of the ballroom
    assert tot > 0


def test_batch_render(tut):
    full, part, pix = tut.batch_render(100, 76, 8, crop=((30, 20), (60, 70)))
    assert tuple(full.shape) == (76 * 100, 3) and tuple(part.shape) == (len(pix), 3)
    f = full.cpu().numpy()[pix]; p = part.cpu().numpy()
    assert np.isfinite(p).all() and p.mean() > 0.01
    # same pixels, different sampler streams: equal in the mean
    assert abs(f.mean() - p.mean()) < 0.15 * f.mean()


@pytest.mark.parametrize("name", ["path", "collocated", "silhouette 0", "silhouette 1", "depth"])
def test_different_integrator(tut, name):
    img, d = tut.different_integrator(name, 64, 64, 8, 8, 8)
    a = img.cpu().numpy()
    assert np.isfinite(a).all() and np.isfinite(d.cpu().numpy()).all() and a.max() > 0
    if name == "silhouette 1":        # the small sphere alone, moving: its mask is 0/1 and the derivative lives on its outline
        assert set(np.unique(np.round(a, 3))) <= {0.0, 1.0} or a.max() <= 1.0 + 1e-5
        assert float(d.abs().sum()) > 0


def test_secondary_edge_guiding(tut):
    plain, guided = tut.secondary_edge_guiding(64, 64, 4, guide=(200, 4, 4, 8))
    p, g = plain.cpu().numpy(), guided.cpu().numpy()
    assert np.isfinite(p).all() and np.isfinite(g).all() and np.abs(g).sum() > 0
    # guiding puts the samples where the integrand is: more pixels receive a contribution
    assert (np.abs(g).sum(axis=1) > 0).sum() >= (np.abs(p).sum(axis=1) > 0).sum()
