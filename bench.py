#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s (spp x pixels / s) of PathTracer(3).renderD on the README
Cornell box, 512x512, spp = sppe = sppse = 32, derivative w.r.t. the x-translation of Mesh[0]
(BASELINE.json configs[2]), on N MI355X GPUs of one node.

A "step" = one full renderD: image + d(image)/d(theta) = interior path tracer with its forward
tangent + primary-edge + secondary-edge boundary integrals (3 kernels) and, for N > 1, ONE RCCL
all-reduce of the stacked [image | derivative] buffer.  Inputs (scene snapshot, BVH) are resident in
HBM before the timed region.  Work is never skipped: every step renders all lanes with fresh seeds.

Multi-GPU (weak scaling): the frame stays 512x512 and every sampler's spp grows to 32*N; rank r
renders the 256-lane chunks k with k % N == r of each sampler (interleaved pixel tiles), so the
per-GPU work is that of the N=1 run; partial images are summed with all_reduce (torch.distributed
"nccl" = RCCL over xGMI).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     — for the dominant kernel: algorithmic BVH-traversal bytes (64 B/node visit + 48 B/triangle
                 test, counted by the instrumented build) / its HIP-event duration.  The scene is
                 LDS-resident, so HBM is NOT the binding roofline: the fp32-VALU and
                 LDS figures that do bound it are reported next to it.
  cpu_baseline — the CPU restatement (oracle/, test infrastructure) timed on this host on a bounded
                 1/k interleaved-chunk sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

W, H, SPP, DEPTH = 512, 512, 32, 3
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3       # fp32 vector peak
LDS_PEAK_GBS = 150000.0        # ds_read_b128 aggregate, every CU streaming


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=0, help="1/k of the lanes for the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # one rank per GPU over RCCL.  (PSDR_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs
    # than ranks - ranks then share devices; never used for reported numbers.)
    backend = os.environ.get("PSDR_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    import psdr_jit_amd as psdr
    from psdr_jit_amd import cabi
    import product
    import scenes

    n = world
    spp = SPP * n
    spec = scenes.cbox_scene(W, H, spp, spp, spp, param="light_x")
    sc = product.build_scene(spec)                      # host configure + BVH + upload (not timed)
    handle = sc._hip_handle()
    L = cabi.lib()
    npx = W * H
    buf = torch.empty((2, npx, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step(i, terms=7):
        a = cabi.make_args(max_depth=DEPTH, seeds=(i, i, i), terms=terms, shard_rank=rank, shard_count=world)
        cabi.check(L.psdr_hip_render_d_fwd(handle, C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), stream))
        if world > 1:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(1000 + i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    samples_per_step = float(npx) * spp                  # spp x pixels of the whole job
    value = samples_per_step / (dt / args.steps) / 1e6

    out = {
        "metric": "Msamples/s (spp x pixels/s) renderD, Cornell box depth=3",
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "README Cornell box (36 triangles) %dx%d PathTracer(%d) renderD d/d(Mesh[0] x-translation), "
                               "spp=sppe=sppse=%d (32 per GPU)" % (W, H, DEPTH, spp),
                   "rays_per_step": int(npx * spp * (1 + 2 * DEPTH) + npx * spp * 2 * (1 + 2 * DEPTH) + npx * spp * 3),
                   "parallelism": "interleaved 256-lane chunks over %d GPU(s) + all_reduce(sum)" % n},
    }

    # ---------------------------------------------------------------- roofline of the dominant kernel (rank 0, N = 1)
    if rank == 0 and n == 1 and not args.no_roofline:
        names = {1: "k_interior<AD>", 2: "k_primary_edges", 4: "k_secondary_edges"}
        per = {}
        for terms in (1, 2, 4):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            step(77, terms)
            torch.cuda.synchronize()
            reps = max(5, args.steps // 2)
            ev0.record()
            for i in range(reps):
                a = cabi.make_args(max_depth=DEPTH, seeds=(i, i, i), terms=terms, zero_output=False)
                cabi.check(L.psdr_hip_render_d_fwd(handle, C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), stream))
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / reps
            c = cabi.Counters()
            a = cabi.make_args(max_depth=DEPTH, seeds=(0, 0, 0), terms=terms)
            cabi.check(L.psdr_hip_render_d_fwd_counted(handle, C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), C.byref(c), stream))
            per[terms] = {"kernel": names[terms], "ms": ms, "rays": c.rays, "nodes": c.nodes_visited, "tris": c.tris_tested, "hits": c.shaded_hits}
        dom = max(per.values(), key=lambda r: r["ms"])
        bytes_alg = 64.0 * dom["nodes"] + 48.0 * dom["tris"]
        flops_alg = 60.0 * dom["nodes"] + 45.0 * dom["tris"] + 250.0 * dom["hits"]
        sec = dom["ms"] * 1e-3
        achieved = bytes_alg / sec / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")     # written from a rocprofv3 --pmc pass, see profiles/README.md
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom["kernel"])
            except Exception:
                traffic = None
        out["roofline"] = {
            "bound": "hbm", "kernel": dom["kernel"], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "note": "algorithmic BVH bytes (64 B/node + 48 B/triangle); the scene is LDS-resident so these bytes are served by LDS, "
                    "not HBM - the binding ceilings are valu/lds below",
            "avg_launch_ms": round(dom["ms"], 4), "rays": dom["rays"], "nodes_per_ray": round(dom["nodes"] / max(dom["rays"], 1), 2),
            "tris_per_ray": round(dom["tris"] / max(dom["rays"], 1), 2),
            "valu": {"achieved": round(flops_alg / sec / 1e12, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(flops_alg / sec / 1e12 / VALU_PEAK_TFLOPS, 4)},
            "lds": {"achieved": round(achieved, 1), "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / LDS_PEAK_GBS, 4)},
            "kernels": [{"kernel": r["kernel"], "avg_launch_ms": round(r["ms"], 4), "rays": r["rays"],
                         "Mrays_per_s": round(r["rays"] / (r["ms"] * 1e-3) / 1e6, 1)} for r in per.values()],
        }

    # ---------------------------------------------------------------- CPU baseline (rank 0, N = 1): the oracle, bounded sample
    if rank == 0 and n == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        ref = orc.OracleScene(spec, [0])
        cores = orc.get_num_threads()
        k = args.cpu_sample
        if k <= 0:
            # calibrate: aim at ~15 s of CPU work
            t = time.perf_counter()
            ref.render_d(max_depth=DEPTH, seeds=(1, 1, 1), shard_rank=0, shard_count=256)
            t256 = time.perf_counter() - t
            k = int(min(256, max(1, round(256 * t256 / 15.0))))
        t = time.perf_counter()
        ref.render_d(max_depth=DEPTH, seeds=(0, 0, 0), shard_rank=0, shard_count=k)
        tc = time.perf_counter() - t
        lanes = sum(1 for c in range((npx * spp + 255) // 256) if c % k == 0) * 256
        out["cpu_baseline"] = {
            "value": round(lanes / tc / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "oracle/ (CPU restatement, OpenMP) renderD on every %d-th 256-lane chunk of the same 512x512 spp=sppe=sppse=32 "
                      "depth-3 workload (%d interior lanes + the same share of edge lanes), %.1f s" % (k, lanes, tc),
        }

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
