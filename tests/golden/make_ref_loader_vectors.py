"""Generates tests/golden/ref_loaders.json: SHA-256 digests (and a few plain numbers) of what the REFERENCE's own vendored file
readers - tinyexr + miniz as src/core/bitmap_loader.cpp drives them, tiny_obj_loader as src/shape/mesh.cpp drives it, compiled
from /root/reference by `make -C oracle ref` into oracle/_ref/libref_loaders.so - return for every data file of the reference's
tutorials (tutorials/data/{cbox,mesh,envmap}; the product ships byte-identical copies under examples/data as inputs).

The digests are DATA (outputs of the reference run here); tests/test_ref_loaders.py checks the product's readers against them on
any box, and against the library itself where it exists.

    python tests/golden/make_ref_loader_vectors.py
"""
import ctypes as C
import hashlib
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DATA = os.path.join(ROOT, "examples", "data")
REF_DATA = "/root/reference/tutorials/data"


class RefObj(C.Structure):
    _fields_ = [("n_vertices", C.c_int), ("n_texcoords", C.c_int), ("n_faces", C.c_int), ("vertices", C.POINTER(C.c_float)),
                ("texcoords", C.POINTER(C.c_float)), ("faces", C.POINTER(C.c_int)), ("face_uvs", C.POINTER(C.c_int))]


def ref_lib():
    path = os.path.join(ROOT, "oracle", "_ref", "libref_loaders.so")
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.ref_load_exr_rgba.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ref_load_obj.argtypes = [C.c_char_p, C.POINTER(RefObj)]
    L.ref_free.argtypes = [C.c_void_p]
    L.ref_free_obj.argtypes = [C.POINTER(RefObj)]
    return L


def ref_exr(L, path):
    out, w, h = C.POINTER(C.c_float)(), C.c_int(), C.c_int()
    rc = L.ref_load_exr_rgba(path.encode(), C.byref(out), C.byref(w), C.byref(h))
    assert rc == 0, (path, rc)
    a = np.ctypeslib.as_array(out, shape=(h.value, w.value, 4)).copy()
    L.ref_free(out)
    return a


def ref_obj(L, path):
    o = RefObj()
    rc = L.ref_load_obj(path.encode(), C.byref(o))
    assert rc == 0, (path, rc)
    v = np.ctypeslib.as_array(o.vertices, shape=(o.n_vertices, 3)).copy()
    f = np.ctypeslib.as_array(o.faces, shape=(o.n_faces, 3)).copy()
    vt = np.ctypeslib.as_array(o.texcoords, shape=(o.n_texcoords, 2)).copy() if o.n_texcoords else None
    fuv = np.ctypeslib.as_array(o.face_uvs, shape=(o.n_faces, 3)).copy() if o.n_texcoords else None
    L.ref_free_obj(C.byref(o))
    return v, f, vt, fuv


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def data_files():
    out = []
    for sub in ("cbox", "mesh", "envmap"):
        d = os.path.join(DATA, sub)
        out += [os.path.join(sub, f) for f in sorted(os.listdir(d))]
    return out


def describe(L, rel, root=DATA):
    path = os.path.join(root, rel)
    if rel.endswith(".exr"):
        a = ref_exr(L, path)
        return dict(kind="exr", shape=list(a.shape), rgb_sha256=digest(a[:, :, :3].astype(np.float32)), max=float(a[:, :, :3].max()),
                    sum=float(a[:, :, :3].astype(np.float64).sum()))
    v, f, vt, fuv = ref_obj(L, path)
    d = dict(kind="obj", n_vertices=int(v.shape[0]), n_faces=int(f.shape[0]), vertices_sha256=digest(v.astype(np.float32)),
             faces_sha256=digest(f.astype(np.int32)), n_texcoords=0 if vt is None else int(vt.shape[0]))
    if vt is not None:
        d["texcoords_sha256"] = digest(vt.astype(np.float32))
        d["face_uvs_sha256"] = digest(fuv.astype(np.int32))
    return d


if __name__ == "__main__":
    L = ref_lib()
    assert L is not None, "needs /root/reference (make -C oracle ref)"
    out = {}
    for rel in data_files():
        out[rel] = describe(L, rel)
        # the shipped copy is the reference's file
        assert open(os.path.join(DATA, rel), "rb").read() == open(os.path.join(REF_DATA, rel), "rb").read(), rel
    json.dump(out, open(os.path.join(HERE, "ref_loaders.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(out), "entries")
