#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s (spp x pixels / s) of PathTracer(3).renderD on the README
Cornell box, derivative w.r.t. the x-translation of Mesh[0], on N MI355X GPUs of one node.

Workloads (BASELINE.json `configs`):
  config 3 (default at N = 1): 512 x 512,  spp = sppe = sppse = 32  - the configuration the metric is quoted on
  config 4 (default at N > 1): 2048 x 2048, spp = sppe = sppse = 64 - STRONG scaling: the frame is fixed and its
           256-lane chunks are dealt round-robin to the ranks (interleaved pixel tiles), ONE RCCL all-reduce of the
           stacked [image | derivative] buffer (96 MiB) assembles it.  `--config 4 --gpus 1` runs it on one GPU.
  --weak   (N > 1): last round's mode - config 3 with spp = 32 * N, per-GPU work fixed.

A "step" = one full renderD: image + d(image)/d(theta) = interior path tracer with its forward tangent +
primary-edge + secondary-edge boundary integrals (3 kernels) and, for N > 1, the all-reduce.  The scene is built
through the package's public Python surface (the reference README's calls) and is resident in HBM before the timed
region; the boundary hands over device pointers only (no PCIe term).  Work is never skipped: every step renders all
lanes with fresh seeds.

Prints ONE JSON line (rank 0).  Extra objects (N = 1, rank 0):
  roofline     - the dominant kernel against the ceiling that binds it.  The scene is LDS / SGPR resident and the
                 compulsory HBM traffic is the two output images, so the bound is fp32 VALU issue: achieved =
                 SURVEY §8(d) flops (60 / node visit + 45 / triangle test + 250 / shaded hit, counted by the
                 instrumented build in this run) / the kernel's HIP-event duration measured in this run.  `traffic`
                 and `issue` are rocprofv3 PMC figures of the same command; they cannot be collected from inside the
                 process, so they are read from the committed summary named in `counters_source` (null if absent).
  parity       - relative L2 of this run's GPU image / derivative against the oracle on the same shard and seeds.
  cpu_baseline - the CPU restatement (oracle/, test infrastructure), built -O3 -march=native on this host, 1 warm-up +
                 median of 5 on a bounded shard of the same workload, all physical cores and one thread.
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "examples")):
    if p not in sys.path:
        sys.path.insert(0, p)

DEPTH = 3
CONFIGS = {3: dict(res=512, spp=32), 4: dict(res=2048, spp=64)}
VALU_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32 vector peak
COUNTERS = os.path.join(ROOT, "profiles", "counters.json")      # written by tools/prof_summary.py from the rocprofv3 --pmc passes


def readme_scene(psdr, res, spp):
    """The reference README's Cornell box (README.md:54-90) through the public API: 8 OBJ meshes, 5 Diffuse BSDFs, one area
    light, camera fov 60 at (208, 273, -800), Mesh[0] translated by 100 * P in x."""
    from psdr_jit_amd import FloatD, Matrix4fC, Matrix4fD
    sc = psdr.Scene()
    sc.opts.spp, sc.opts.sppe, sc.opts.sppse = spp, spp, spp
    sc.opts.width = sc.opts.height = res
    sc.opts.log_level = 0
    sensor = psdr.PerspectiveCamera(60, 0.000001, 10000000.)
    sensor.to_world = Matrix4fD([[1., 0., 0., 208.], [0., 1., 0., 273.], [0., 0., 1., -800.], [0., 0., 0., 1.]])
    sc.add_Sensor(sensor)
    for name, rgb in (("light", [0.0, 0.0, 0.0]), ("cat", [0.5, 0.5, 0.5]), ("white", [0.95, 0.95, 0.95]), ("green", [0.20, 0.90, 0.20]), ("red", [0.90, 0.20, 0.20])):
        sc.add_BSDF(psdr.DiffuseBSDF(rgb), name)
    data = os.path.join(ROOT, "examples", "data", "cbox")
    eye = [[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]
    sc.add_Mesh(os.path.join(data, "cbox_luminaire.obj"), Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., -0.5], [0., 0., 1., 0.], [0., 0., 0., 1.]]), "light",
                psdr.AreaLight([20.0, 20.0, 8.0]))
    for f, b in (("smallbox", "cat"), ("largebox", "cat"), ("floor", "white"), ("ceiling", "white"), ("back", "white"), ("greenwall", "green"), ("redwall", "red")):
        sc.add_Mesh(os.path.join(data, "cbox_%s.obj" % f), Matrix4fC(eye), b, None)
    P = FloatD(0.).requires_grad_()
    sc.param_map["Mesh[0]"].set_transform(Matrix4fD([[1., 0., 0., P * 100], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
    sc.configure()
    sc.configure([0])
    return sc, P


def host_cpu():
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif k == "" and phys is not None and core is not None:
                cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return model, (len(cores) or os.cpu_count() or 1), (os.cpu_count() or 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, choices=(3, 4), default=0, help="BASELINE config (default: 3 at N = 1, 4 at N > 1)")
    ap.add_argument("--weak", action="store_true", help="N > 1: config 3 with spp = 32 * N (weak scaling)")
    ap.add_argument("--cpu-shard", type=int, default=0, help="the CPU baseline renders every k-th 256-lane chunk (0 = calibrate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
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

    n = world
    weak = args.weak and n > 1
    cfg = args.config or (3 if (n == 1 or weak) else 4)
    res = CONFIGS[cfg]["res"]
    spp = CONFIGS[cfg]["spp"] * (n if weak else 1)
    sc, P = readme_scene(psdr, res, spp)                # host configure + filter primitives + upload (not timed)
    integ = psdr.PathTracer(DEPTH)
    # forward tangent of the parameter: d to_world_left / dP of Mesh[0]; one untimed call installs it in the device scene
    leaf = sc.param_map["Mesh[0]"].to_world_left
    d_leaf = np.zeros((4, 4), np.float32)
    d_leaf[0, 3] = 100.0
    psdr.render_d_fwd(integ, sc, 0, seed=12345, tangents={leaf: d_leaf})
    handle = sc._hip_handle()
    L = cabi.lib()
    npx = res * res
    buf = torch.empty((2, npx, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def launch(seed, terms=7, shard_rank=rank, shard_count=world, zero=True, out=buf):
        a = cabi.make_args(max_depth=DEPTH, seeds=(seed, seed, seed), terms=terms, shard_rank=shard_rank, shard_count=shard_count, zero_output=zero)
        cabi.check(L.psdr_hip_render_d_fwd(handle, C.byref(a), out[0].data_ptr(), out[1].data_ptr(), stream))

    def step(i):
        launch(i)
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
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config %d%s: README Cornell box (36 triangles) %dx%d PathTracer(%d) renderD d/d(Mesh[0] x-translation), "
                               "spp=sppe=sppse=%d" % (cfg, " weak-scaled" if weak else "", res, res, DEPTH, spp),
                   "rays_per_step": int(npx * spp * (1 + 2 * DEPTH) + npx * spp * 2 * (1 + 2 * DEPTH) + npx * spp * 3),
                   "parallelism": "256-lane chunks dealt round-robin to %d GPU(s)%s" % (n, " + one all_reduce(sum) of [image | derivative]" if n > 1 else "")},
    }

    # ---------------------------------------------------------------- roofline of the dominant kernel (rank 0, N = 1)
    if rank == 0 and n == 1 and not args.no_roofline:
        names = {1: "k_interior<AD>", 2: "k_primary_edges", 4: "k_secondary_edges"}
        per = {}
        for terms in (1, 2, 4):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # torch's current stream = the launch stream
            launch(77, terms)
            torch.cuda.synchronize()
            reps = max(5, args.steps // 2)
            ev0.record()
            for i in range(reps):
                launch(i, terms, zero=False)
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / reps
            c = cabi.Counters()
            a = cabi.make_args(max_depth=DEPTH, seeds=(0, 0, 0), terms=terms)
            cabi.check(L.psdr_hip_render_d_fwd_counted(handle, C.byref(a), buf[0].data_ptr(), buf[1].data_ptr(), C.byref(c), stream))
            per[terms] = {"kernel": names[terms], "ms": ms, "rays": c.rays, "nodes": c.nodes_visited, "tris": c.tris_tested, "hits": c.shaded_hits}
        dom = max(per.values(), key=lambda r: r["ms"])
        flops_alg = 60.0 * dom["nodes"] + 45.0 * dom["tris"] + 250.0 * dom["hits"]
        bytes_alg = 64.0 * dom["nodes"] + 48.0 * dom["tris"]
        sec = dom["ms"] * 1e-3
        achieved = flops_alg / sec / 1e12
        counters = None
        if os.path.exists(COUNTERS):
            try:
                counters = json.load(open(COUNTERS))
            except Exception:
                counters = None
        krec = (counters or {}).get("kernels", {}).get(dom["kernel"], {})
        out["roofline"] = {
            "bound": "valu", "kernel": dom["kernel"], "achieved": round(achieved, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / VALU_PEAK_TFLOPS, 4),
            "traffic": krec.get("hbm_bytes_per_launch"), "counters_source": (counters or {}).get("source"),
            "issue": krec.get("issue"),
            "note": "SURVEY 8(d) flops (60/node + 45/triangle + 250/shaded hit, counted in this run) / HIP-event time of this run; "
                    "compulsory HBM bytes per launch = the two output images (%.1f MB), so HBM is not the ceiling" % (2 * npx * 12 / 1e6),
            "avg_launch_ms": round(dom["ms"], 4), "rays": dom["rays"], "nodes_per_ray": round(dom["nodes"] / max(dom["rays"], 1), 2),
            "tris_per_ray": round(dom["tris"] / max(dom["rays"], 1), 2), "flops_per_launch": flops_alg,
            "scene_bytes": {"algorithmic_per_launch": bytes_alg, "rate_GBps": round(bytes_alg / sec / 1e9, 1),
                            "served_by": "wave-uniform scalar loads / LDS (one fetch serves 64 lanes): not HBM traffic"},
            "kernels": [{"kernel": r["kernel"], "avg_launch_ms": round(r["ms"], 4), "rays": r["rays"],
                         "Mrays_per_s": round(r["rays"] / (r["ms"] * 1e-3) / 1e6, 1)} for r in per.values()],
        }

    # ---------------------------------------------------------------- the oracle legs (rank 0, N = 1): parity + CPU baseline
    if rank == 0 and n == 1 and not (args.no_cpu_baseline and args.no_parity):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import oracle as orc
        import scenes                                    # the oracle's neutral description of the same README scene
        orc.build(native=True)                           # -O3 -march=native on this host (bit-identical to the default build)
        spec = scenes.cbox_scene(res, res, spp, spp, spp, param="light_x")
        ref = orc.OracleScene(spec, [0])
        model, phys, logical = host_cpu()
        orc.set_num_threads(phys)
        n_chunks = (npx * spp + 255) // 256

        def lanes_of(k):
            return ((n_chunks + k - 1) // k) * 256

        k = args.cpu_shard
        if k <= 0:
            # shard for the parity check: ~1 s of oracle time
            t = time.perf_counter()
            ref.render_d(max_depth=DEPTH, seeds=(1, 1, 1), shard_rank=0, shard_count=512)
            t512 = time.perf_counter() - t
            k = int(min(512, max(1, round(512 * t512 / 1.0))))
        if not args.no_parity:
            want_img, want_d = ref.render_d(max_depth=DEPTH, seeds=(0, 0, 0), shard_rank=0, shard_count=k)
            launch(0, 7, shard_rank=0, shard_count=k)
            torch.cuda.synchronize()
            got = buf.cpu().numpy()

            def rel(a, b):
                return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))
            out["parity"] = {"rel_l2_image": rel(got[0], want_img), "rel_l2_derivative": rel(got[1], want_d), "tolerance": 1e-3,
                             "sample": "every %d-th 256-lane chunk of each sampler of this workload, seeds (0,0,0), HIP vs oracle" % k}
        if not args.no_cpu_baseline:
            # The sample: the same scene, resolution, depth and three terms at a REDUCED sample count per pixel (every pixel and
            # every sampler, so the OpenMP loops stay balanced - a sparse shard leaves most threads idle); samples per second
            # is the metric, so the figure is directly comparable.  spp is calibrated to ~3 s per run.
            def timed(sref, runs):
                ts = []
                for r in range(runs + 1):                # first run = warm-up
                    t = time.perf_counter()
                    sref.render_d(max_depth=DEPTH, seeds=(r, r, r))
                    ts.append(time.perf_counter() - t)
                return statistics.median(ts[1:])
            one = orc.OracleScene(scenes.cbox_scene(res, res, 1, 1, 1, param="light_x"), [0])
            t = time.perf_counter()
            one.render_d(max_depth=DEPTH, seeds=(9, 9, 9))
            t_one = time.perf_counter() - t
            cspp = int(min(spp, max(1, round(3.0 / max(t_one, 1e-3)))))
            sref = orc.OracleScene(scenes.cbox_scene(res, res, cspp, cspp, cspp, param="light_x"), [0])
            tc = timed(sref, 5)
            orc.set_num_threads(1)
            sres = max(16, res // 8)                      # one thread: the same scene at 1/64 of the pixels, 1 sample each
            tiny = orc.OracleScene(scenes.cbox_scene(sres, sres, 1, 1, 1, param="light_x"), [0])
            t1 = timed(tiny, 3)
            orc.set_num_threads(phys)
            out["cpu_baseline"] = {
                "value": round(npx * cspp / tc / 1e6, 4), "unit": "Msamples/s", "cores": phys, "kind": "port",
                "one_thread": round(sres * sres / t1 / 1e6, 5), "cpu": model, "logical_cpus": logical,
                "sample": "oracle/ (CPU restatement of the reference algorithm, g++ -O3 -march=native, OpenMP over %d physical cores) renderD of this "
                          "workload at spp=sppe=sppse=%d instead of %d (all %d pixels, all three terms): 1 warm-up + median of 5, %.2f s per run; "
                          "one thread: %dx%d pixels at 1 sample, median of 3, %.2f s" % (phys, cspp, spp, npx, tc, sres, sres, t1),
            }

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
