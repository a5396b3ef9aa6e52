"""The product's two file readers against the REFERENCE's own: psdr_jit_amd/exr.py against tinyexr + miniz as
src/core/bitmap_loader.cpp:12-52 drives them, the host model's OBJ reader (csrc/host/scene_host.cpp Mesh::load) against
tiny_obj_loader as src/shape/mesh.cpp:166-243 drives it - on all 12 data files of the reference's tutorials, bit for bit.

Two legs: (1) against tests/golden/ref_loaders.json, digests of the reference readers' outputs generated in the build container
by tests/golden/make_ref_loader_vectors.py (runs on any box); (2) where oracle/_ref/libref_loaders.so exists (built from
/root/reference by `make -C oracle ref`, __graft_entry__.build() does it when the reference is present) against the library
itself, array by array."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_ref_loader_vectors as gen                                         # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_loaders.json")))


@pytest.fixture(scope="module")
def psdr():
    import __graft_entry__
    __graft_entry__.build()
    import psdr_jit_amd
    return psdr_jit_amd


def product_obj(psdr, path):
    m = psdr.Mesh()
    m.load(path)
    v = np.asarray(m._get("vertex_positions", False), np.float32).reshape(-1, 3)          # m_vertex_positions_raw, as read
    f = np.asarray(m.face_indices, np.int32).reshape(-1, 3)
    vt = np.asarray(m.vertex_uv, np.float32).reshape(-1, 2)
    fuv = np.asarray(m.face_uv_indices, np.int32).reshape(-1, 3)
    return v, f, (vt if len(vt) else None), (fuv if len(fuv) else None)


def product_exr(psdr, path):
    from psdr_jit_amd import exr
    return np.asarray(exr.read_rgb(path), np.float32)


def test_the_file_list_is_the_tutorials_data():
    assert sorted(GOLD) == sorted(gen.data_files()) and len(GOLD) == 12


@pytest.mark.parametrize("rel", sorted(GOLD))
def test_reader_matches_the_reference_digest(psdr, rel):
    want = GOLD[rel]
    path = os.path.join(gen.DATA, rel)
    if want["kind"] == "exr":
        a = product_exr(psdr, path)
        assert list(a.shape) == want["shape"][:2] + [3]
        assert gen.digest(a) == want["rgb_sha256"] and float(a.max()) == want["max"]
        return
    v, f, vt, fuv = product_obj(psdr, path)
    assert (v.shape[0], f.shape[0]) == (want["n_vertices"], want["n_faces"])
    assert gen.digest(v) == want["vertices_sha256"] and gen.digest(f) == want["faces_sha256"]
    assert (0 if vt is None else vt.shape[0]) == want["n_texcoords"]
    if vt is not None:
        assert gen.digest(vt) == want["texcoords_sha256"] and gen.digest(fuv) == want["face_uvs_sha256"]


def test_readers_match_the_reference_library(psdr):
    L = gen.ref_lib()
    if L is None:
        pytest.skip("oracle/_ref/libref_loaders.so not built (needs /root/reference)")
    for rel in sorted(GOLD):
        path = os.path.join(gen.DATA, rel)
        assert gen.describe(L, rel) == GOLD[rel], rel           # the fixture is what the library returns today
        if rel.endswith(".exr"):
            assert np.array_equal(product_exr(psdr, path), gen.ref_exr(L, path)[:, :, :3])
        else:
            got, want = product_obj(psdr, path), gen.ref_obj(L, path)
            for g, w in zip(got, want):
                assert (g is None) == (w is None)
                if g is not None:
                    assert np.array_equal(g, w), rel
