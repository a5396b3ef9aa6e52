"""The multi-GPU path with the HIP kernels on BOTH ranks: two processes (torch.distributed, gloo rendezvous on 127.0.0.1) share the
one GPU of the test box, each renders its lane shard with the product, the partial [image | derivative] buffers and the parameter
adjoints are all-reduced, and the result must equal the single-process render.  (tests/test_distributed_cpu.py covers the
collective glue on CPU; an 8-GPU RCCL run is the driver's to make.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(script_args, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)


def test_two_ranks_forward_and_backward_equal_one_process():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists)")
    r = _torchrun([os.path.join("tools", "check_2rank_backward.py")])
    assert r.returncode == 0 and "backward OK" in r.stdout, r.stdout[-3000:]


def test_bench_two_ranks_config4_strong_scaling_line():
    """bench.py --gpus 2 as the driver launches it (here: gloo, both ranks on the one GPU): config 4, strong scaling, one JSON line"""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists)")
    r = _torchrun(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"], {"PSDR_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "config 4" in d["config"]["workload"] and "2048x2048" in d["config"]["workload"]
    assert d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["unit"] == "Msamples/s"
    assert "roofline" not in d and "cpu_baseline" not in d        # N = 1 only
    # what the line offers for reading a scaling curve: per-rank render and all-reduce times (HIP events, outside the timed region)
    sb = d["scale_breakdown"]
    assert len(sb["render_ms_per_rank"]) == 2 and len(sb["allreduce_ms_per_rank"]) == 2 and min(sb["render_ms_per_rank"]) > 0
    assert sb["all_reduce_bytes"] == 2 * 2048 * 2048 * 3 * 4
    # both collective forms were timed as whole steps (the interior term's all_reduce under the edge kernels / one all_reduce at the end)
    assert sb["step_ms_split_collectives"] > 0 and sb["step_ms_single_collective"] > 0 and sb["timed_form"] == "split"
    assert abs(sb["overlap_ms"] - (sb["step_ms_single_collective"] - sb["step_ms_split_collectives"])) < 1e-2
    # ... and the contiguous row-tile partition beside it (PSDR_SHARD=rows): render time per rank (the tiles' balance), the step, the bytes of its collectives
    rt = sb["row_tiles"]
    assert len(rt["render_ms_per_rank"]) == 2 and min(rt["render_ms_per_rank"]) > 0 and rt["step_ms"] > 0
    assert rt["all_gather_bytes_per_rank"] == 2 * 1024 * 2048 * 3 * 4 and rt["all_reduce_bytes"] == 2048 * 2048 * 3 * 4


def test_bench_dry_run_says_whether_the_node_can_run_n_ranks():
    """`bench.py --gpus N --dry`: visible devices against the ranks asked for, with the launch line - no rendering"""
    for n, want in ((1, 0), (64, 1)):
        r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--dry"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        assert r.returncode == want and d["dry"] and d["ok"] == (want == 0) and d["gpus_requested"] == n, r.stdout[-1000:]
        if want:
            assert "needs 64 visible GPUs" in d["message"]


def test_rccl_preflight_world_size_one():
    """RCCL preflight for the driver's 8-GPU run: bench.py's N > 1 code path - `init_process_group("nccl", device_id=...)`, the
    device-pointer all_reduce(sum) of the [image | derivative] buffer after every renderD, the max-over-ranks timing reduce, the
    barriers and destroy_process_group - executed through RCCL itself on the test box's one GPU, as torch.distributed.run launches
    a single rank (PSDR_BENCH_FORCE_DIST=1 keeps the collectives in at world size 1).  Config 4's scene at a reduced frame."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update({"PSDR_BENCH_FORCE_DIST": "1", "PSDR_BENCH_BACKEND": "nccl"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--config", "4", "--res", "256", "--spp", "8", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-parity", "--no-roofline", "--no-backward", "--no-config5"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    d = json.loads(lines[0])
    assert d["rccl"]["backend"] == "nccl" and d["rccl"]["ranks"] == 1 and d["rccl"]["all_reduce_bytes"] == 2 * 256 * 256 * 3 * 4
    assert len(d["scale_breakdown"]["render_ms_per_rank"]) == 1 and d["scale_breakdown"]["allreduce_ms_per_rank"][0] >= 0.0
    assert d["value"] > 0 and d["n_gpus"] == 1 and "config 4" in d["config"]["workload"]
    # the same frame without the process group: the all-reduce of one rank must not change the image (checked through the rate only
    # being finite here; equality of sharded and unsharded frames is test_two_ranks_forward_and_backward_equal_one_process)
