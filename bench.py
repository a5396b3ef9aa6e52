#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s (spp x pixels / s) of PathTracer(3).renderD on the README
Cornell box, derivative w.r.t. the x-translation of Mesh[0], on N MI355X GPUs of one node.

Workloads (BASELINE.json `configs`):
  config 3 (default at N = 1): 512 x 512,  spp = sppe = sppse = 32  - the configuration the metric is quoted on
  config 4 (default at N > 1): 2048 x 2048, spp = sppe = sppse = 64 - STRONG scaling: the frame is fixed and its
           256-lane chunks are dealt round-robin to the ranks (interleaved pixel tiles), ONE RCCL all-reduce of the
           stacked [image | derivative] buffer (96 MiB) assembles it.  `--config 4 --gpus 1` runs it on one GPU.
  --weak   (N > 1): last round's mode - config 3 with spp = 32 * N, per-GPU work fixed.
  config 5 (`--config 5`, N = 1; N > 1 shards it like config 4): the BVH path - envmap-lit 81 920-triangle mesh (examples/synth.py), 1024 x 1024,
           spp = sppe = sppse = 64, renderD w.r.t. the DiffuseBSDF albedo, secondary-edge guiding grid [2000, 5, 5, 32].

A "step" = one full renderD: image + d(image)/d(theta) = interior path tracer with its forward tangent +
primary-edge + secondary-edge boundary integrals (3 kernels) and, for N > 1, the all-reduce.  The scene is built
through the package's public Python surface (the reference README's calls) and is resident in HBM before the timed
region; the boundary hands over device pointers only (no PCIe term).  Every step renders the whole frame with fresh
seeds; nothing is cached between steps.  (Since round 4 the interior term passes over the samples of pixels no ray can
leave towards a triangle - a conservative coverage mask of the scene, built at scene creation; the value of such a sample is
zero in the reference as well, the frame is checked against the CPU restatement in the run (`parity`), and
PSDR_NO_LIVE_MASK=1 switches the mask off: 7.21 instead of 7.08 ms.  The samples still count in `value`, as they do for the
reference, which launches them.)

Prints ONE JSON line (rank 0).  Extra objects (N = 1, rank 0; each can be switched off, none is inside the timed region):
  roofline     - the dominant kernel against the ceiling that binds it.  The scene is LDS / SGPR resident and the
                 compulsory HBM traffic is the two output images, so the bound is fp32 VALU issue: achieved =
                 SURVEY §8(d) flops (60 / node visit + 45 / triangle test + 250 / shaded hit, counted by the
                 instrumented build in this run) / the kernel's HIP-event duration measured in this run.  `traffic`
                 and `issue` are rocprofv3 PMC figures of the same command; they cannot be collected from inside the
                 process, so they are read from the committed summary named in `counters_source` (null if absent).
  backward     - reverse mode on the same workload: ms per psdr_hip_render_d_bwd (adjoints of every triangle row, BSDF colour, emitter,
                 edge row: nothing filtered), all three terms and each alone, HIP events on the launch stream.
  config5      - (default run only) the BVH path at BASELINE config 5's full size: ms per renderD, per-kernel HIP-event times, counted
                 nodes / triangles per ray, the dominant kernel against the L2 ceiling (34.5 TB/s) with the HBM rate and L2 hit rate
                 of the committed rocprofv3 passes; `--config 5` makes it the headline workload instead.
  api          - (N = 1) through the public Python surface (tools/configure_timing.py): api_call_ms = renderD + forward_grad as a user calls them, configure_ms per kind
                 of change (nothing / a colour / moved vertices / the rebuild of rounds 1-4), step_ms = configure + renderD + backward; the same inside `config5`.
  rccl         - PSDR_BENCH_FORCE_DIST=1 runs the N > 1 code path (process group with device_id, device-pointer all-reduce, teardown)
                 at whatever world size the launcher gives, also 1 (the preflight of tests/test_gpu_distributed.py).
  parity       - relative L2 of this run's GPU image / derivative against the oracle on the same shard and seeds.
  cpu_baseline - the CPU restatement (oracle/, test infrastructure), built -O3 -march=native on this host, 1 warm-up +
                 median of 5 on the same frame at a reduced sample count, on the cores this process may use (affinity mask and
                 cgroup CPU quota, `cores`), and the same frame on one thread.
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
CONFIGS = {3: dict(res=512, spp=32), 4: dict(res=2048, spp=64), 5: dict(res=1024, spp=64)}
GUIDING = [2000, 5, 5, 32]     # config 5: preprocess_secondary_edges grid (tutorials/secondary_edge_guiding.ipynb)
VALU_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32 vector peak
L2_PEAK_TBPS = 34.5            # MI355X_MICROARCH.md: aggregate L2 bandwidth (8 XCDs x 4 MiB)
COUNTERS = os.path.join(ROOT, "profiles", "counters.json")      # written by tools/prof_summary.py from the rocprofv3 --pmc passes


def readme_scene(psdr, res, spp):
    """The reference README's Cornell box (README.md:54-90) through the public API: 8 meshes (generated from coordinates), 5 Diffuse BSDFs, one area
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
    import synth
    obj = synth.write_cornell_box()                  # the eight meshes from coordinates (examples/synth.py), written as OBJ text to a temporary directory
    eye = [[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]
    sc.add_Mesh(obj["luminaire"], Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., -0.5], [0., 0., 1., 0.], [0., 0., 0., 1.]]), "light",
                psdr.AreaLight([20.0, 20.0, 8.0]))
    for f, b in (("smallbox", "cat"), ("largebox", "cat"), ("floor", "white"), ("ceiling", "white"), ("back", "white"), ("greenwall", "green"), ("redwall", "red")):
        sc.add_Mesh(obj[f], Matrix4fC(eye), b, None)
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


def usable_cpus(phys):
    """How many cores this PROCESS may really keep busy: the physical cores of the host, cut down to its affinity mask and to the
    CPU quota of its cgroup (a container sees every core of the host in /proc/cpuinfo but is throttled to cpu.max - 128 OpenMP threads
    on a 16-CPU quota ran at 7 % parallel efficiency in round 2's baseline).  -> (threads to use, quota in CPUs or None)"""
    n = phys
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                      # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                # cgroup v1
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n), quota


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, choices=(3, 4, 5), default=0, help="BASELINE config (default: 3 at N = 1, 4 at N > 1)")
    ap.add_argument("--res", type=int, default=0, help="override the configuration's resolution (measurement aid; the line names it)")
    ap.add_argument("--spp", type=int, default=0, help="override the configuration's sample counts (measurement aid; the line names it)")
    ap.add_argument("--no-backward", action="store_true")
    ap.add_argument("--no-config5", action="store_true")
    ap.add_argument("--no-static-skip", action="store_true", help="leave out the extra run with psdr_render_args.skip_static_edges (profiling passes: one kind of launch per kernel)")
    ap.add_argument("--no-api", action="store_true", help="skip the legs through the public Python surface (api_call_ms, configure_ms, step_ms)")
    ap.add_argument("--weak", action="store_true", help="N > 1: config 3 with spp = 32 * N (weak scaling)")
    ap.add_argument("--cpu-shard", type=int, default=0, help="the CPU baseline renders every k-th 256-lane chunk (0 = calibrate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--dry", action="store_true", help="check that --gpus N can run on this node (visible devices, launcher environment), print the verdict, render nothing")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launch_line = "python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus %d" % (args.gpus, args.gpus)
    if args.dry:
        # the 1 -> 8 curve is the driver's to run; this says beforehand whether the node can: one rank per GPU needs N visible devices
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        ok = have >= args.gpus
        print(json.dumps({"dry": True, "gpus_requested": args.gpus, "devices_visible": have, "ok": ok,
                          "launch": launch_line if args.gpus > 1 else "python bench.py",
                          "message": "ok" if ok else "bench.py --gpus %d needs %d visible GPUs (one rank per GPU over RCCL), this node shows %d" % (args.gpus, args.gpus, have)}))
        raise SystemExit(0 if ok else 1)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with: %s   (WORLD_SIZE is %d)" % (launch_line, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if world > 1 and os.environ.get("PSDR_BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py --gpus %d: one rank per GPU over RCCL needs %d visible GPUs, this node shows %d (PSDR_BENCH_BACKEND=gloo shares devices, for tests only)" %
                         (world, world, torch.cuda.device_count()))
    # one rank per GPU over RCCL.  (PSDR_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs
    # than ranks - ranks then share devices; never used for reported numbers.)
    backend = os.environ.get("PSDR_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    force_dist = os.environ.get("PSDR_BENCH_FORCE_DIST", "0") == "1"
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if use_dist:
        dist.barrier()
    import psdr_jit_amd as psdr
    from psdr_jit_amd import cabi

    n = world
    weak = args.weak and n > 1
    cfg = args.config or (3 if (n == 1 or weak) else 4)
    res = args.res or CONFIGS[cfg]["res"]
    spp = (args.spp or CONFIGS[cfg]["spp"]) * (n if weak else 1)
    guiding = None
    if cfg == 5:
        import synth
        sc, leaf = synth.config5_scene(psdr, res, spp)   # BVH build + upload + environment cell masses (not timed)
        d_leaf = np.ones(3, np.float32)
        integ = psdr.PathTracer(DEPTH)
        integ.preprocess_secondary_edges(sc, 0, GUIDING, 1, 0)
        guiding = integ._guiding_handle(0) or None       # psdr_render_args.guiding: the grid the integrator holds for sensor 0
    else:
        sc, P = readme_scene(psdr, res, spp)            # host configure + filter primitives + upload (not timed)
        integ = psdr.PathTracer(DEPTH)
        # forward tangent of the parameter: d to_world_left / dP of Mesh[0]; one untimed call installs it in the device scene
        leaf = sc.param_map["Mesh[0]"].to_world_left
        d_leaf = np.zeros((4, 4), np.float32)
        d_leaf[0, 3] = 100.0
    integ.trace_static_edges = True                   # (the untimed call that installs the tangent traces what the timed C-ABI launches trace)
    psdr.render_d_fwd(integ, sc, 0, seed=12345, tangents={leaf: d_leaf})
    handle = sc._hip_handle()
    L = cabi.lib()
    npx = res * res
    buf = torch.empty((2, npx, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def launch(seed, terms=7, shard_rank=rank, shard_count=world, zero=True, out=buf, skip_static=False, shard_mode=0):
        a = cabi.make_args(max_depth=DEPTH, seeds=(seed, seed, seed), terms=terms, shard_rank=shard_rank, shard_count=shard_count, zero_output=zero, guiding=guiding,
                           skip_static_edges=skip_static, shard_mode=shard_mode)
        cabi.check(L.psdr_hip_render_d_fwd(handle, C.byref(a), out[0].data_ptr(), out[1].data_ptr(), stream))

    single_collective = os.environ.get("PSDR_SINGLE_COLLECTIVE", "0") == "1"
    edge = torch.empty((2, npx, 3), dtype=torch.float32, device="cuda") if use_dist else None

    rows_timed = os.environ.get("PSDR_SHARD", "interleaved").lower() == "rows"       # the partition of the timed region (psdr_jit_amd._shard_mode)
    tile = psdr._row_tile(npx, res, world, False) if use_dist else 0

    def step_rows(i):
        # PSDR_SHARD=rows, BASELINE north_star's partition: contiguous pixel-row tiles - the interior term is assembled by all_gather_into_tensor (1 / N of the bytes per
        # rank, started under the edge kernels), the edge terms' derivative (contiguous lane runs) is summed
        launch(i, terms=1, shard_mode=1)
        done = psdr._gather_tiles(buf, npx, tile, world, async_op=True)
        launch(i, terms=6, out=edge, shard_mode=1)
        dist.all_reduce(edge[1], op=dist.ReduceOp.SUM)
        full = done()
        full[1] += edge[1]
        return full

    def step(i, single=None):
        if use_dist and rows_timed and single is None:
            return step_rows(i)
        # N > 1: the form psdr_jit_amd._render_terms uses - the interior term (disjoint pixels per rank, first to finish) goes into its own all_reduce, which RCCL's
        # stream runs under the edge kernels; only the edge terms' derivative is reduced after them.  PSDR_SINGLE_COLLECTIVE=1: rounds 1-4's one all_reduce of everything.
        if not use_dist:
            launch(i)
        elif single_collective if single is None else single:
            launch(i)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        else:
            launch(i, terms=1)
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
            launch(i, terms=6, out=edge)
            dist.all_reduce(edge[1], op=dist.ReduceOp.SUM)
            work.wait()
            buf[1] += edge[1]

    def sync():
        if use_dist:
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
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    # N > 1 (outside the timed region): what a step consists of on every rank - HIP events on the launch stream around the render and around the
    # all-reduce of three extra steps - so that a scaling curve can be attributed (a slow rank, a slow collective, or launch skew)
    breakdown = None
    if use_dist:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        acc = [0.0, 0.0]
        reps = 3
        for i in range(reps):
            sync()
            ev[0].record()
            launch(5000 + i)
            ev[1].record()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            ev[2].record()
            torch.cuda.synchronize()
            acc[0] += ev[0].elapsed_time(ev[1]); acc[1] += ev[1].elapsed_time(ev[2])
        # the two collective forms, whole steps, wall clock between synchronisations (what the overlap hides = their difference)
        form_ms = {}
        for name, flag in (("split", False), ("single", True)):
            step(6000, single=flag)
            sync()
            t_f = time.perf_counter()
            for i in range(reps):
                step(6001 + i, single=flag)
            sync()
            form_ms[name] = (time.perf_counter() - t_f) / reps * 1e3
        # ... and the contiguous row-tile partition (render times per rank show its balance, the step its all-gather)
        rows_acc, rows_error = 0.0, None
        try:
            for i in range(reps):
                sync()
                ev[0].record()
                launch(7000 + i, shard_mode=1)
                ev[1].record()
                torch.cuda.synchronize()
                rows_acc += ev[0].elapsed_time(ev[1])
            step_rows(7100)
            sync()
            t_f = time.perf_counter()
            for i in range(reps):
                step_rows(7101 + i)
            sync()
            form_ms["rows"] = (time.perf_counter() - t_f) / reps * 1e3
        except Exception as e:                      # (an extra outside the timed region must not cost the line)
            rows_error = repr(e)[:300]
            form_ms["rows"] = 0.0
        mine = torch.tensor([acc[0] / reps, acc[1] / reps, rows_acc / reps], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        breakdown = {"render_ms_per_rank": [round(float(t[0]), 3) for t in allr], "allreduce_ms_per_rank": [round(float(t[1]), 3) for t in allr],
                     "all_reduce_bytes": int(buf.numel() * 4),
                     "step_ms_split_collectives": round(form_ms["split"], 3), "step_ms_single_collective": round(form_ms["single"], 3),
                     "overlap_ms": round(form_ms["single"] - form_ms["split"], 3), "timed_form": "rows" if rows_timed else ("single" if single_collective else "split"),
                     "row_tiles": {"render_ms_per_rank": [round(float(t[2]), 3) for t in allr], "step_ms": round(form_ms["rows"], 3), "all_gather_bytes_per_rank": int(2 * tile * 3 * 4),
                                   "all_reduce_bytes": int(npx * 3 * 4), "error": rows_error,
                                   "note": "PSDR_SHARD=rows: contiguous pixel-row tiles (psdr_render_args.shard_mode 1), interior term by all_gather_into_tensor, edge derivative by all_reduce; "
                                           "render_ms_per_rank shows the balance of the tiles against the interleaved chunks above"},
                     "note": "HIP events on the launch stream, mean of %d steps outside the timed region; a rank's all-reduce time includes its wait for the slowest rank's render; "
                             "the interior tiles of the ranks are disjoint, the edge terms scatter over the frame - the whole [image | derivative] buffer is summed" % reps}
    samples_per_step = float(npx) * spp                  # spp x pixels of the whole job
    value = samples_per_step / (dt / args.steps) / 1e6

    out = {
        "metric": "Msamples/s (spp x pixels/s) renderD, %s depth=3" % ("envmap-lit 82k-triangle mesh" if cfg == 5 else "Cornell box"),
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("BASELINE config 5: envmap-lit 81 920-triangle mesh + floor, %dx%d PathTracer(%d) renderD d/d(albedo), spp=sppe=sppse=%d, "
                                "guiding grid %s" % (res, res, DEPTH, spp, GUIDING)) if cfg == 5 else
                               ("BASELINE config %d%s: README Cornell box (36 triangles) %dx%d PathTracer(%d) renderD d/d(Mesh[0] x-translation), "
                                "spp=sppe=sppse=%d" % (cfg, " weak-scaled" if weak else "", res, res, DEPTH, spp)),
                   "rays_per_step": int(npx * spp * (1 + 2 * DEPTH) + npx * spp * 2 * (1 + 2 * DEPTH) + npx * spp * 3),
                   "parallelism": "256-lane chunks dealt round-robin to %d GPU(s)%s" % (n, (" + one all_reduce(sum) of [image | derivative]" if single_collective else
                                    " + all_reduce(sum) of the interior term's [image | derivative] under the edge kernels, then of the edge terms' derivative") if n > 1 else "")},
    }

    if breakdown is not None:
        out["scale_breakdown"] = breakdown
    if n == 1 and not args.no_static_skip:
        # psdr_render_args.skip_static_edges (ABI 14), what the Python surface asks for: a primary-edge sample on an edge that does not move under the installed tangent
        # adds exactly zero to the derivative image and is not traced - the same numbers reach the image and the derivative (tests/test_gpu_configs.py).  The headline above traces
        # every sample, as the reference does and as rounds 1-4 did.
        launch(2000, skip_static=True)
        sync()
        t1 = time.perf_counter()
        for i in range(args.steps):
            launch(i, skip_static=True)
        sync()
        dt1 = time.perf_counter() - t1
        out["static_edges_skipped"] = {"ms_per_step": round(dt1 / args.steps * 1e3, 4), "value": round(samples_per_step / (dt1 / args.steps) / 1e6, 3),
                                       "note": "the same workload with psdr_render_args.skip_static_edges = 1: identical outputs, the primary-edge samples with a zero edge velocity not traced"}
    try:        # the live-pixel mask of the timed scene (psdr_hip_scene_live_pixels): which part of the frame the interior term passes over
        n_live = C.c_int64(0)
        cabi.check(cabi.lib().psdr_hip_scene_live_pixels(C.c_void_p(handle), 0, None, C.byref(n_live)))
        # `value` counts every sample the reference would launch; `value_traced` only the interior samples this build traces (live pixels x spp) - the
        # like-for-like figure against builds without the mask (PSDR_NO_LIVE_MASK=1) and against rounds 1-3
        out["traced_samples_per_step"] = int(n_live.value * spp)
        out["value_traced"] = round(n_live.value * spp / (dt / args.steps) / 1e6, 3)
        out["live_pixels"] = {"fraction": round(n_live.value / float(npx), 4),
                              "note": "interior-term samples of the other pixels are provably zero and are passed over before seeding (they count in `value`, as for the reference, "
                                      "which launches them); PSDR_NO_LIVE_MASK=1 renders them"}
    except Exception as e:      # (never in the way of the line)
        out["live_pixels"] = {"error": str(e)[:200]}

    counters = None
    if os.path.exists(COUNTERS):
        try:
            counters = json.load(open(COUNTERS))
        except Exception:
            counters = None

    def per_kernel(do_launch, h, bufs, gd, reps):
        """HIP-event time of each term's kernel launched alone (torch's current stream = the launch stream) + the counted build's
        rays / node visits / triangle tests / shaded hits for the same launch"""
        names = {1: "k_interior<AD>", 2: "k_primary_edges", 4: "k_secondary_edges"}
        per = {}
        for terms in (1, 2, 4):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            do_launch(77, terms)
            torch.cuda.synchronize()
            ev0.record()
            for i in range(reps):
                do_launch(i, terms, zero=False)
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / reps
            c = cabi.Counters()
            a = cabi.make_args(max_depth=DEPTH, seeds=(0, 0, 0), terms=terms, guiding=gd)
            cabi.check(L.psdr_hip_render_d_fwd_counted(h, C.byref(a), bufs[0].data_ptr(), bufs[1].data_ptr(), C.byref(c), stream))
            per[terms] = {"kernel": names[terms], "ms": ms, "rays": c.rays, "nodes": c.nodes_visited, "tris": c.tris_tested, "hits": c.shaded_hits}
        return per

    def kernel_rows(per):
        return [{"kernel": r["kernel"], "avg_launch_ms": round(r["ms"], 4), "rays": r["rays"],
                 "Mrays_per_s": round(r["rays"] / (r["ms"] * 1e-3) / 1e6, 1)} for r in per.values()]

    def bvh_roofline(per, h, section):
        """The dominant kernel of a BVH scene against the L2 ceiling: algorithmic bytes = node bytes x node visits + 48 x triangle
        tests (with the node size of this build's tree), counted in this run, / the kernel's HIP-event time of this
        run.  HBM bytes per launch and the L2 hit rate come from the committed rocprofv3 --pmc passes of the same command."""
        nn, nl, md, nb = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        cabi.check(L.psdr_hip_scene_stats(C.c_void_p(h), C.byref(nn), C.byref(nl), C.byref(md), C.byref(nb)))
        node_bytes = int(L.psdr_hip_bvh_node_bytes())
        dom = max(per.values(), key=lambda r: r["ms"])
        sec = dom["ms"] * 1e-3
        bytes_alg = float(node_bytes) * dom["nodes"] + 48.0 * dom["tris"]
        flops_alg = 60.0 * dom["nodes"] + 45.0 * dom["tris"] + 250.0 * dom["hits"]
        krec = ((counters or {}).get(section) or {}).get("kernels", {}).get(dom["kernel"], {})
        hbm = krec.get("hbm_bytes_per_launch")
        return {
            "bound": "l2", "kernel": dom["kernel"], "achieved": round(bytes_alg / sec / 1e12, 3), "peak": L2_PEAK_TBPS, "unit": "TB/s",
            "frac": round(bytes_alg / sec / 1e12 / L2_PEAK_TBPS, 4),
            "traffic": hbm, "hbm_GBps": (round(hbm / (krec.get("avg_launch_us", 0) * 1e-6) / 1e9, 1) if hbm and krec.get("avg_launch_us") else None),
            "l2_hit_rate": krec.get("l2_hit_rate"), "counters_source": ((counters or {}).get(section) or {}).get("source"),
            "counters_workload": ((counters or {}).get(section) or {}).get("workload"), "issue": krec.get("issue"),
            "note": "algorithmic bytes (%d B per node visit + 48 B per triangle test, counted in this run) / HIP-event time of this run against the "
                    "aggregate L2 bandwidth; valu_frac = SURVEY 8(d) flops / time / %.1f TFLOP/s" % (node_bytes, VALU_PEAK_TFLOPS),
            "avg_launch_ms": round(dom["ms"], 4), "rays": dom["rays"], "Grays_per_s": round(dom["rays"] / sec / 1e9, 3),
            "nodes_per_ray": round(dom["nodes"] / max(dom["rays"], 1), 2), "tris_per_ray": round(dom["tris"] / max(dom["rays"], 1), 2),
            "bytes_per_launch": bytes_alg, "valu_frac": round(flops_alg / sec / 1e12 / VALU_PEAK_TFLOPS, 4),
            "bvh": {"nodes": nn.value, "leaves": nl.value, "depth": md.value, "node_bytes": node_bytes},
            "kernels": kernel_rows(per),
        }

    # ---------------------------------------------------------------- roofline of the dominant kernel (rank 0, N = 1)
    if rank == 0 and n == 1 and not args.no_roofline:
        per = per_kernel(launch, handle, buf, guiding, max(3 if cfg == 5 else 5, args.steps // 2))
        if cfg == 5:
            out["roofline"] = bvh_roofline(per, handle, "config5")
        else:
            dom = max(per.values(), key=lambda r: r["ms"])
            flops_alg = 60.0 * dom["nodes"] + 45.0 * dom["tris"] + 250.0 * dom["hits"]
            bytes_alg = 64.0 * dom["nodes"] + 48.0 * dom["tris"]
            sec = dom["ms"] * 1e-3
            achieved = flops_alg / sec / 1e12
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
                "kernels": kernel_rows(per),
            }

    # ---------------------------------------------------------------- reverse mode on the same workload (rank 0, N = 1)
    def backward_leg(scene, h, gd, n_pixels, reps):
        """ms per psdr_hip_render_d_bwd with adjoints of EVERYTHING in the snapshot wanted (no mesh filter, colours and emitters
        included) - what loss.backward() costs before the host chain rule; all three terms and each term alone"""
        snap = scene._snapshot()
        cam = scene.param_map["Sensor[0]"]
        n_tri, n_sec = int(snap["triangles"].shape[0]), int(snap["sec_edges"].shape[0])
        n_prim = int(np.asarray(cam._primary_edge_ids()).reshape(-1, 3).shape[0])
        pm = scene.param_map
        nb_ = sum(1 for k in pm if k.startswith("BSDF[") and not k.startswith("BSDF[id="))
        ne_ = sum(1 for k in pm if k.startswith("Emitter[") and not k.startswith("Emitter[id="))
        g_tri = torch.zeros((n_tri, 22), device="cuda")
        g_b, g_e = torch.zeros((max(1, nb_), 3), device="cuda"), torch.zeros((max(1, ne_), 3), device="cuda")
        g_s, g_p = torch.zeros((max(1, n_sec), 6), device="cuda"), torch.zeros((max(1, n_prim), 4), device="cuda")
        w = torch.ones((n_pixels, 3), device="cuda")            # d loss / d image
        g = cabi.Grads(g_tri.data_ptr(), g_b.data_ptr(), g_e.data_ptr(), g_s.data_ptr(), g_p.data_ptr())
        res_ = {}
        for terms, name in ((7, "all"), (1, "interior"), (2, "primary_edges"), (4, "secondary_edges")):
            def bwd(seed):
                a = cabi.make_args(max_depth=DEPTH, seeds=(seed, seed, seed), terms=terms, guiding=gd)
                cabi.check(L.psdr_hip_render_d_bwd(h, C.byref(a), w.data_ptr(), C.byref(g), stream))
            bwd(99)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for i in range(reps):
                bwd(i)
            ev1.record()
            torch.cuda.synchronize()
            res_[name] = round(ev0.elapsed_time(ev1) / reps, 4)
        return {"ms": res_["all"], "ms_per_term": {k: v for k, v in res_.items() if k != "all"},
                "adjoints": "triangle rows [%d, 22], BSDF colours [%d, 3], emitters [%d, 3], secondary-edge rows [%d, 6], primary-edge rows [%d, 4]; "
                            "nothing filtered" % (n_tri, nb_, ne_, n_sec, n_prim),
                "timing": "HIP events on the launch stream around %d psdr_hip_render_d_bwd calls (fresh seeds), d loss / d image = 1" % reps}

    if rank == 0 and n == 1 and not args.no_backward:
        b = backward_leg(sc, handle, guiding, npx, max(3, args.steps // 4))
        b["forward_ms"] = round(ms_per_step, 4)
        out["backward"] = b

    # ---------------------------------------------------------------- the BVH path at BASELINE config 5's size (default run only)
    if rank == 0 and n == 1 and cfg == 3 and not args.no_config5 and not (args.res or args.spp):
        import synth
        c5 = CONFIGS[5]
        sc5, leaf5 = synth.config5_scene(psdr, c5["res"], c5["spp"])
        integ5 = psdr.PathTracer(DEPTH)
        t_g = time.perf_counter()
        integ5.preprocess_secondary_edges(sc5, 0, GUIDING, 1, 0)
        torch.cuda.synchronize()
        t_g = time.perf_counter() - t_g
        gd5 = integ5._guiding_handle(0) or None
        integ5.trace_static_edges = True
        psdr.render_d_fwd(integ5, sc5, 0, seed=12345, tangents={leaf5: np.ones(3, np.float32)})
        h5 = sc5._hip_handle()
        buf5 = torch.empty((2, c5["res"] * c5["res"], 3), dtype=torch.float32, device="cuda")

        def launch5(seed, terms=7, zero=True, skip_static=False):
            a = cabi.make_args(max_depth=DEPTH, seeds=(seed, seed, seed), terms=terms, zero_output=zero, guiding=gd5, skip_static_edges=skip_static)
            cabi.check(L.psdr_hip_render_d_fwd(h5, C.byref(a), buf5[0].data_ptr(), buf5[1].data_ptr(), stream))
        launch5(1000)
        torch.cuda.synchronize()
        steps5 = 3
        t5 = time.perf_counter()
        for i in range(steps5):
            launch5(i)
        torch.cuda.synchronize()
        t5 = (time.perf_counter() - t5) / steps5
        r5 = bvh_roofline(per_kernel(launch5, h5, buf5, gd5, 2), h5, "config5")
        out["config5"] = {
            "workload": "BASELINE config 5 on one GPU: envmap-lit 81 920-triangle mesh + floor (examples/synth.py), 1024 x 512 map, %dx%d PathTracer(%d) renderD "
                        "d/d(albedo), spp=sppe=sppse=%d, guiding grid %s" % (c5["res"], c5["res"], DEPTH, c5["spp"], GUIDING),
            "ms_per_step": round(t5 * 1e3, 3), "value": round(c5["res"] * c5["res"] * c5["spp"] / t5 / 1e6, 2), "unit": "Msamples/s", "steps": steps5,
            "guiding_build_ms": round(t_g * 1e3, 2), "finite": bool(torch.isfinite(buf5).all()), "roofline": r5,
        }
        # (the parameter of BASELINE config 5 is a colour: no edge moves, so with psdr_render_args.skip_static_edges the primary-edge term traces nothing)
        launch5(2000, skip_static=True)
        torch.cuda.synchronize()
        t5s = time.perf_counter()
        for i in range(steps5):
            launch5(i, skip_static=True)
        torch.cuda.synchronize()
        t5s = (time.perf_counter() - t5s) / steps5
        out["config5"]["static_edges_skipped"] = {"ms_per_step": round(t5s * 1e3, 3), "value": round(c5["res"] * c5["res"] * c5["spp"] / t5s / 1e6, 2)}
        if not args.no_backward:
            out["config5"]["backward"] = backward_leg(sc5, h5, gd5, c5["res"] * c5["res"], 2)
        if not args.no_api:
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import configure_timing
                out["config5"]["api"] = configure_timing.measure(psdr, sc5, "BSDF[0]", reps=5, depth=DEPTH, heavy_reps=2)
            except Exception as e:
                out["config5"]["api"] = {"error": repr(e)[:300]}
        del buf5, sc5

    if use_dist:
        out["rccl"] = {"backend": backend, "ranks": world, "all_reduce_bytes": int(buf.numel() * 4),
                       "note": "process group bound with device_id, one all_reduce(sum) of the device buffer per step, destroy_process_group at exit"}

    # ---------------------------------------------------------------- the oracle legs (rank 0, N = 1): parity + CPU baseline
    if rank == 0 and n == 1 and not (args.no_cpu_baseline and args.no_parity):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import oracle as orc
        import scenes                                    # the oracle's neutral description of the same README scene
        orc.build(native=True)                           # -O3 -march=native on this host (bit-identical to the default build)
        def spec_of(r, k):
            return scenes.config5_scene(r, r, k, k, k) if cfg == 5 else scenes.cbox_scene(r, r, k, k, k, param="light_x")
        spec = spec_of(res, spp)
        ref = orc.OracleScene(spec, [0])
        g_ref = ref.guiding_build(0, GUIDING, nrounds=1, seed=0, max_depth=DEPTH) if cfg == 5 else None
        model, phys, logical = host_cpu()
        n_thr, quota = usable_cpus(phys)
        orc.set_num_threads(n_thr)
        n_chunks = (npx * spp + 255) // 256

        def lanes_of(k):
            return ((n_chunks + k - 1) // k) * 256

        k = args.cpu_shard
        if k <= 0:
            # shard for the parity check: ~1 s of oracle time
            t = time.perf_counter()
            ref.render_d(max_depth=DEPTH, seeds=(1, 1, 1), shard_rank=0, shard_count=512, guiding=g_ref)
            t512 = time.perf_counter() - t
            k = int(min(512, max(1, round(512 * t512 / 1.0))))
        if not args.no_parity:
            want_img, want_d = ref.render_d(max_depth=DEPTH, seeds=(0, 0, 0), shard_rank=0, shard_count=k, guiding=g_ref)
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
            # is the metric, so the figure is directly comparable.  Threads = the cores this process may use (usable_cpus);
            # the one-thread figure is the SAME frame at one sample per pixel.
            def timed(sref, runs, warm=1):
                ts = []
                for r in range(runs + warm):                # first run = warm-up
                    t = time.perf_counter()
                    sref.render_d(max_depth=DEPTH, seeds=(r, r, r))
                    ts.append(time.perf_counter() - t)
                return statistics.median(ts[warm:])
            one = orc.OracleScene(spec_of(res, 1), [0])
            t_one = timed(one, 1)                         # all usable threads, 1 sample per pixel: calibrates the sample count
            cspp = int(min(spp, max(1, round(3.0 / max(t_one, 1e-3)))))
            sref = orc.OracleScene(spec_of(res, cspp), [0])
            tc = timed(sref, 5)
            orc.set_num_threads(1)
            t1 = timed(one, 1) if t_one * n_thr < 20.0 else None       # (skipped where one thread would need more than ~20 s per frame)
            orc.set_num_threads(n_thr)
            out["cpu_baseline"] = {
                "value": round(npx * cspp / tc / 1e6, 4), "unit": "Msamples/s", "cores": n_thr, "kind": "port",
                "one_thread": round(npx / t1 / 1e6, 5) if t1 else None, "parallel_efficiency": round((npx * cspp / tc) / (npx / t1) / n_thr, 3) if t1 else None,
                "cpu": model, "physical_cores": phys, "logical_cpus": logical, "cgroup_cpu_quota": quota,
                "sample": "oracle/ (CPU restatement of the reference algorithm, g++ -O3 -march=native, OpenMP) renderD of this workload at spp=sppe=sppse=%d "
                          "instead of %d (all %d pixels, all three terms) on %d threads = the cores this process may use (host: %d physical cores, cgroup "
                          "CPU quota %s): 1 warm-up + median of 5, %.2f s per run; one thread: the same frame at 1 sample per pixel, %s"
                          % (cspp, spp, npx, n_thr, phys, ("%.1f" % quota) if quota else "none", tc, ("%.2f s" % t1) if t1 else "not timed"),
            }

    # ---------------------------------------------------------------- the calls a user makes (rank 0, N = 1; last: these legs reconfigure the scene)
    if rank == 0 and n == 1 and not args.no_api and cfg != 5:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import configure_timing
            sc.param_map["Mesh[0]"].set_transform(psdr.Matrix4fC([[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]]))
            api = configure_timing.measure(psdr, sc, "BSDF[1]", reps=7, depth=DEPTH)
            api["note"] = ("through the public Python surface, device synchronised around every call, median after one warm-up: api_call_ms = PathTracer(3).renderD(sc, 0) + "
                           "forward_grad(img, P) (SURVEY 8(d): wall time of one renderD call with its derivative); configure_ms = Scene.configure() after no change / a colour / "
                           "a translation of Mesh[0] (`rebuild`: the device scene destroyed and created as rounds 1-4 did); step_ms = configure + renderD + loss + backward "
                           "(reverse_*) or + forward_grad (forward_vertices)")
            api["api_call_over_ms_per_step"] = round(api["api_call_ms"] / ms_per_step, 3)
            out["api"] = api
        except Exception as e:      # (never in the way of the line)
            out["api"] = {"error": repr(e)[:300]}

    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
