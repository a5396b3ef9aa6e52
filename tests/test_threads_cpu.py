"""The host thread team of both native libraries (psdr_jit_amd/csrc/common/threads.h; round 6: the loops of Scene.configure / psdr_hip_scene_update run on persistent, sleeping threads
instead of threads created per loop): tests/cpp/threads_check.cpp compiled with g++ - coverage and deterministic chunks for any thread count, re-use over thousands of loops, concurrent and
nested callers, a forked child, and an idle team that burns no CPU (a spinning team is what exhausted the cgroup's CPU quota under OpenMP in round 5)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_thread_team(tmp_path):
    src = os.path.join(ROOT, "tests", "cpp", "threads_check.cpp")
    exe = os.path.join(tmp_path, "threads_check")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", src, "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    for threads in ("", "3", "16"):
        env = dict(os.environ)
        if threads:
            env["PSDR_HOST_THREADS"] = threads
        r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
        assert r.returncode == 0 and r.stdout.startswith("OK"), (threads, r.stdout[-2000:])
