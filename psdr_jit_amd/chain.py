"""Host-side chain rule of reverse mode: from the adjoints of the configured snapshot (what
psdr_hip_render_d_bwd returns: TriangleInfo rows, edge tables, colours) to the user's leaf tensors
(mesh transforms, raw vertex positions, reflectances, radiances).

The differentiable part of Scene::configure (reference src/shape/mesh.cpp:23-62,317-369 process_mesh +
SecondaryEdgeInfo, src/sensor/perspective.cpp:130-143 primary-edge projection) is restated with torch
ops in float64 and differentiated by torch.autograd — O(#triangles) glue; the per-sample work is all
in the HIP kernels.
"""
import math

import numpy as np
import torch

F64 = torch.float64


def _t(x):
    return torch.as_tensor(np.asarray(x, dtype=np.float64), dtype=F64)


def _xform_pos(M, V):
    h = V @ M[:3, :3].T + M[:3, 3]
    w = V @ M[3, :3] + M[3, 3]
    return h / w[:, None]


def _process_mesh(Vw, F):
    """rows [p0 e1 e2 n0 n1 n2 face_normal area] (22 floats) of process_mesh, reference mesh.cpp:23-62."""
    i0, i1, i2 = F[:, 0], F[:, 1], F[:, 2]
    p0 = Vw[i0]
    e1 = Vw[i1] - p0
    e2 = Vw[i2] - p0
    N = torch.cross(e1, e2, dim=1)
    a = N.norm(dim=1)
    vn = torch.zeros_like(Vw)
    for idx in (i0, i1, i2):
        vn = vn.index_add(0, idx, N)
    used = torch.zeros(Vw.shape[0], dtype=torch.bool)
    used[F.reshape(-1)] = True
    nrm = vn.norm(dim=1, keepdim=True)
    n_v = torch.where(used[:, None], vn / torch.where(used[:, None], nrm, torch.ones_like(nrm)), torch.zeros_like(vn))
    return torch.cat([p0, e1, e2, n_v[i0], n_v[i1], n_v[i2], N / a[:, None], (0.5 * a)[:, None]], dim=1)


def _camera_to_sample(fov_x, near, far, aspect, orthographic=False):
    recip = 1.0 / (far - near)
    P = torch.zeros((4, 4), dtype=F64)
    if orthographic:
        # OrthographicCamera (reference orthographic.cpp, transform.h:75-78): scale(1, 1, 1 / (far - near)) . translate(0, 0, -near)
        P[0, 0] = 1.0
        P[1, 1] = 1.0
        P[2, 2] = recip
        P[2, 3] = -near * recip
        P[3, 3] = 1.0
    else:
        cot = 1.0 / math.tan(math.radians(fov_x * 0.5))
        P[0, 0] = cot
        P[1, 1] = cot
        P[2, 2] = far * recip
        P[2, 3] = -near * far * recip
        P[3, 2] = 1.0
    S = torch.diag(torch.tensor([-0.5, -0.5 * aspect, 1.0, 1.0], dtype=F64))
    T = torch.eye(4, dtype=F64)
    T[0, 3] = -1.0
    T[1, 3] = -1.0 / aspect
    return S @ T @ P


def snapshot_tensors(scene, sensor_id, leaf_of):
    """leaf_of(obj, name) -> float64 tensor (a fresh leaf for differentiable parameters, a constant otherwise).
    Returns (tri_rows, sec_rows, prim_rows, refl_rows, rad_rows, camera to_world) in the snapshot's row order."""
    pm = scene.param_map
    Vws, tri_rows, sec_rows = [], [], []
    for i in range(scene.num_meshes):
        if "Mesh[%d]" % i not in pm:
            # the environment map's bounding cube (scene.cpp:442-480): 12 constant faces, not a parameter
            Vws.append(None)
            tri_rows.append(torch.zeros((12, 22), dtype=F64))
            continue
        m = pm["Mesh[%d]" % i]
        V = leaf_of(m, "vertex_positions").reshape(-1, 3)
        M = leaf_of(m, "to_world_left").reshape(4, 4) @ leaf_of(m, "to_world").reshape(4, 4) @ leaf_of(m, "to_world_right").reshape(4, 4)
        Vw = _xform_pos(M, V)
        Vws.append(Vw)
        n_edges = m.num_edges() if (scene.opts.sppse > 0 and m.enable_edges) else 0
        if not Vw.requires_grad:
            # a mesh none of whose parameters is wanted: its rows carry no gradient, only their count matters (config 5: the floor beside the 81 920-triangle blob)
            tri_rows.append(torch.zeros((m.num_faces, 22), dtype=F64))
            if n_edges:
                sec_rows.append(torch.zeros((n_edges, 6), dtype=F64))
            continue
        F = torch.as_tensor(np.asarray(m.face_indices, dtype=np.int64))
        tri_rows.append(_process_mesh(Vw, F))
        if n_edges:
            E = torch.as_tensor(np.asarray(m.edge_indices(), dtype=np.int64))
            sec_rows.append(torch.cat([Vw[E[:, 0]], Vw[E[:, 1]] - Vw[E[:, 0]]], dim=1))
    tri = torch.cat(tri_rows, dim=0)
    sec = torch.cat(sec_rows, dim=0) if sec_rows else torch.zeros((0, 6), dtype=F64)
    cam = pm["Sensor[%d]" % sensor_id]
    ids = torch.as_tensor(np.asarray(cam._primary_edge_ids(), dtype=np.int64)).reshape(-1, 3)
    # Sensor::configure: to_world = left . raw . right (sensor.cpp); the interior term's adjoint of it is psdr_grads.g_camera
    tw = leaf_of(cam, "to_world_left").reshape(4, 4) @ leaf_of(cam, "to_world").reshape(4, 4) @ leaf_of(cam, "to_world_right").reshape(4, 4)
    if ids.shape[0] > 0:
        fov, near, far = cam._camera_params()
        aspect = float(scene.opts.width) / float(scene.opts.height)
        w2s = _camera_to_sample(fov, near, far, aspect, bool(cam.orthographic)) @ torch.linalg.inv(tw)
        # end points of the kept edges, gathered per mesh (config 5 keeps 26 592 edges: one gather per mesh, not one per edge)
        q0 = torch.zeros((ids.shape[0], 3), dtype=F64)
        q1 = torch.zeros((ids.shape[0], 3), dtype=F64)
        for mid in torch.unique(ids[:, 0]).tolist():
            rows = torch.nonzero(ids[:, 0] == mid).reshape(-1)
            q0 = q0.index_add(0, rows, Vws[mid][ids[rows, 1]])
            q1 = q1.index_add(0, rows, Vws[mid][ids[rows, 2]])
        q0 = _xform_pos(w2s, q0)
        q1 = _xform_pos(w2s, q1)
        prim = torch.cat([q0[:, :2], q1[:, :2]], dim=1)
    else:
        prim = torch.zeros((0, 4), dtype=F64)
    nb = sum(1 for k in pm if k.startswith("BSDF[") and not k.startswith("BSDF[id="))
    ne = sum(1 for k in pm if k.startswith("Emitter[") and not k.startswith("Emitter[id="))
    def _refl(b):     # the colour row g_bsdf refers to: Diffuse reflectance, Microfacet diffuse reflectance (textures: no adjoint)
        if type(b).__name__ in ("RoughConductorBSDF", "RoughDielectricBSDF", "MicrofacetBSDFPerVertex", "NormalMapBSDF"):
            return torch.zeros(3, dtype=F64)
        if type(b).__name__ == "MicrofacetBSDF":
            r = leaf_of(b, "diffuseReflectance")
            return torch.zeros(3, dtype=F64) if r.dim() == 3 else r.reshape(-1).expand(3)
        r = leaf_of(b, "reflectance")
        return torch.zeros(3, dtype=F64) if r.dim() == 3 else r.reshape(-1).expand(3)
    refl = torch.stack([_refl(pm["BSDF[%d]" % i]) for i in range(nb)]) if nb else torch.zeros((0, 3), dtype=F64)
    def _rad(e):      # the environment map's texels are not differentiated (its row of g_emitter stays zero)
        return torch.zeros(3, dtype=F64) if type(e).__name__ == "EnvironmentMap" else leaf_of(e, "radiance").reshape(-1).expand(3)
    rad = torch.stack([_rad(pm["Emitter[%d]" % i]) for i in range(ne)]) if ne else torch.zeros((0, 3), dtype=F64)
    return tri, sec, prim, refl, rad, tw


# ------------------------------------------------------------------------------------------------------------------------------------
# The same chain rule without autograd: the per-triangle / per-edge part runs in the host core (Scene::chain_geometry, csrc/host/scene_host.cpp: double arithmetic on
# host threads), what is left here are 4 x 4 products.  snapshot_tensors above stays as the restatement the CPU suite checks it against (tests/test_host_cpu.py).
def _factor_grads(obj, g_M):
    """adjoints of (to_world_left, to_world, to_world_right) from the adjoint of their product M = L . R . Rt"""
    L, R, Rt = (np.asarray(obj._get(n, False), dtype=np.float64).reshape(4, 4) for n in ("to_world_left", "to_world", "to_world_right"))
    return {"to_world_left": g_M @ (R @ Rt).T, "to_world": L.T @ g_M @ Rt.T, "to_world_right": (L @ R).T @ g_M}


def native_geometry_grads(scene, sensor_id, wanted, g_tri, g_sec, g_prim, g_camera=None):
    """wanted: [(object, parameter name)] of Mesh / Sensor parameters whose adjoint is asked for -> {(id(object), name): float64 array}.
    g_tri [n_triangles, 22], g_sec [n_sec_edges, 6], g_prim [n_primary_edges, 4]: snapshot-row adjoints (numpy float32); g_camera [4, 4]: the interior / secondary-edge
    terms' adjoint of the sensor's to_world (psdr_grads.g_camera) or None."""
    pm = scene.param_map
    mesh_ix = {}
    for i in range(scene.num_meshes):
        m = pm.get("Mesh[%d]" % i)
        if m is not None:
            mesh_ix[id(m)] = i
    cam = pm["Sensor[%d]" % sensor_id]
    want_mesh = sorted({mesh_ix[id(o)] for o, _ in wanted if id(o) in mesh_ix})
    want_cam = any(o is cam for o, _ in wanted)
    f32 = lambda a, cols: np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1, cols))
    L, R, Rt = (np.asarray(cam._get(n, False), dtype=np.float64).reshape(4, 4) for n in ("to_world_left", "to_world", "to_world_right"))
    tw = L @ R @ Rt
    fov, near, far = cam._camera_params()
    C = _camera_to_sample(fov, near, far, float(scene.opts.width) / float(scene.opts.height), bool(cam.orthographic)).numpy()
    inv = np.linalg.inv(tw)
    meshes, g_w2s = scene._chain_geometry(sensor_id, f32(g_tri, 22), f32(g_sec, 6), f32(g_prim, 4), want_mesh, want_cam, np.ascontiguousarray(C @ inv))
    out = {}
    by_mesh = {int(i): (np.asarray(gm), np.asarray(gv)) for i, gm, gv in meshes}
    for o, name in wanted:
        if id(o) in mesh_ix:
            gm, gv = by_mesh[mesh_ix[id(o)]]
            out[(id(o), name)] = gv if name == "vertex_positions" else _factor_grads(o, gm)[name]
    if want_cam:
        g_tw = np.zeros((4, 4)) if g_camera is None else np.asarray(g_camera, dtype=np.float64).reshape(4, 4).copy()
        g_tw += -inv.T @ (C.T @ np.asarray(g_w2s)) @ inv.T            # world_to_sample = C . tw^-1;  Y = X^-1: g_X = -Y^T g_Y Y^T
        fg = {"to_world_left": g_tw @ (R @ Rt).T, "to_world": L.T @ g_tw @ Rt.T, "to_world_right": (L @ R).T @ g_tw}
        for o, name in wanted:
            if o is cam:
                out[(id(o), name)] = fg[name]
    return out
