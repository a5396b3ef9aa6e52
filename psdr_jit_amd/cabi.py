"""ctypes view of the C ABI in include/psdr_hip.h (libpsdr_hip.so), used by bench.py and the
parity tests to call the entry points directly with device pointers."""
import ctypes as C
import os

from . import build as _build

SYMBOLS = [
    "psdr_hip_last_error", "psdr_hip_abi_version", "psdr_hip_device_count", "psdr_hip_set_device",
    "psdr_hip_scene_create", "psdr_hip_scene_update", "psdr_hip_scene_last_update", "psdr_hip_scene_check_tree", "psdr_hip_scene_check_rows", "psdr_hip_scene_destroy", "psdr_hip_scene_stats", "psdr_hip_scene_live_pixels", "psdr_hip_bvh_node_bytes", "psdr_hip_scene_tex_layout", "psdr_hip_trace", "psdr_hip_trace_pairs", "psdr_hip_ray_intersect", "psdr_hip_env_sample", "psdr_hip_env_pdf", "psdr_hip_env_cell_masses", "psdr_hip_env_cell_masses_xf",
    "psdr_hip_render_c", "psdr_hip_render_d_fwd", "psdr_hip_render_d_bwd", "psdr_hip_render_c_counted", "psdr_hip_render_d_fwd_counted",
    "psdr_hip_li_lanes", "psdr_hip_guiding_build", "psdr_hip_guiding_mass", "psdr_hip_guiding_num_cells",
    "psdr_hip_guiding_destroy", "psdr_hip_tea64", "psdr_hip_sampler_floats",
]


class Sampler(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("skip", C.c_uint64)]


class RenderArgs(C.Structure):
    _fields_ = [("sensor_id", C.c_int32), ("max_depth", C.c_int32), ("hide_emitters", C.c_int32),
                ("samplers", Sampler * 3), ("pix_ids", C.c_void_p), ("n_pix", C.c_int32), ("terms", C.c_int32),
                ("shard_rank", C.c_int32), ("shard_count", C.c_int32), ("guiding", C.c_void_p), ("zero_output", C.c_int32), ("direct_mode", C.c_int32),
                ("field_mode", C.c_int32), ("field_object", C.c_int32), ("intensity", C.c_float), ("d_intensity", C.c_float), ("skip_static_edges", C.c_int32), ("shard_mode", C.c_int32)]


class Grads(C.Structure):
    _fields_ = [("g_triangles", C.c_void_p), ("g_bsdf", C.c_void_p), ("g_emitter", C.c_void_p), ("g_sec_edges", C.c_void_p),
                ("g_prim_edges", C.c_void_p), ("mesh_filter", C.c_void_p), ("skip_bsdf", C.c_int32), ("skip_emitter", C.c_int32),
                ("g_tex", C.c_void_p), ("g_camera", C.c_void_p), ("g_env", C.c_void_p), ("g_env_scale", C.c_void_p), ("g_mat", C.c_void_p), ("g_env_from_world", C.c_void_p), ("g_uv_xf", C.c_void_p),
                ("prim_edge_filter", C.c_void_p)]


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("shaded_hits", C.c_uint64)]


class UpdateInfo(C.Structure):      # psdr_update_info
    _fields_ = [("tree", C.c_int32), ("reallocated", C.c_int32), ("bytes_uploaded", C.c_int64), ("sah_cost", C.c_double), ("sah_cost_built", C.c_double),
                ("ms_tree", C.c_double), ("ms_fill", C.c_double), ("ms_upload", C.c_double), ("ms_total", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_build.HIP_LIB):
            raise ImportError("libpsdr_hip.so is not built: run python -m psdr_jit_amd.build")
        L = C.CDLL(_build.HIP_LIB)
        L.psdr_hip_last_error.restype = C.c_char_p
        L.psdr_hip_tea64.restype = C.c_uint64
        L.psdr_hip_tea64.argtypes = [C.c_uint64, C.c_uint64]
        L.psdr_hip_sampler_floats.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p]
        L.psdr_hip_trace.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.psdr_hip_trace_pairs.argtypes = L.psdr_hip_trace.argtypes
        L.psdr_hip_env_sample.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 6
        L.psdr_hip_env_pdf.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 5
        L.psdr_hip_render_c.argtypes = [C.c_void_p, C.POINTER(RenderArgs), C.c_void_p, C.c_void_p]
        L.psdr_hip_render_d_fwd.argtypes = [C.c_void_p, C.POINTER(RenderArgs), C.c_void_p, C.c_void_p, C.c_void_p]
        L.psdr_hip_render_d_bwd.argtypes = [C.c_void_p, C.POINTER(RenderArgs), C.c_void_p, C.POINTER(Grads), C.c_void_p]
        L.psdr_hip_render_c_counted.argtypes = [C.c_void_p, C.POINTER(RenderArgs), C.c_void_p, C.POINTER(Counters), C.c_void_p]
        L.psdr_hip_render_d_fwd_counted.argtypes = [C.c_void_p, C.POINTER(RenderArgs), C.c_void_p, C.c_void_p, C.POINTER(Counters), C.c_void_p]
        L.psdr_hip_li_lanes.argtypes = [C.c_void_p, C.POINTER(RenderArgs), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        L.psdr_hip_scene_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.psdr_hip_scene_last_update.argtypes = [C.c_void_p, C.POINTER(UpdateInfo)]
        L.psdr_hip_scene_check_tree.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.psdr_hip_scene_live_pixels.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def check(rc):
    if rc:
        raise RuntimeError(lib().psdr_hip_last_error().decode())


def make_args(sensor_id=0, max_depth=1, hide_emitters=False, seeds=(0, 0, 0), skips=(0, 0, 0), pix_ids_ptr=0, n_pix=0,
              terms=7, shard_rank=0, shard_count=1, guiding=None, zero_output=True, direct_mis=-1, field=-1, field_object=-1, intensity=1.0, d_intensity=0.0,
              skip_static_edges=False, shard_mode=0):
    a = RenderArgs()
    a.sensor_id, a.max_depth, a.hide_emitters = sensor_id, max_depth, int(hide_emitters)
    for k in range(3):
        a.samplers[k].seed, a.samplers[k].skip = int(seeds[k]), int(skips[k])
    a.pix_ids, a.n_pix, a.terms = pix_ids_ptr or None, n_pix, terms
    a.shard_rank, a.shard_count, a.guiding, a.zero_output = shard_rank, shard_count, guiding, int(zero_output)
    a.direct_mode = int(direct_mis) + 1
    a.field_mode, a.field_object, a.intensity, a.d_intensity = int(field) + 1, int(field_object), float(intensity), float(d_intensity)
    a.skip_static_edges = int(skip_static_edges)
    a.shard_mode = int(shard_mode)          # 0: interleaved 256-lane chunks, 1: contiguous runs (pixel-row tiles), include/psdr_hip.h
    return a
