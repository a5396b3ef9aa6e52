"""How the CPU restatement (oracle/, the bench's cpu_baseline) scales with threads on this host, and how many cores the process may
really use (affinity mask, cgroup quota):  python tools/cpu_scaling.py [res] [spp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc
import scenes

res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, open(f).read().strip())
    except OSError:
        pass
print("OMP env", {k: v for k, v in os.environ.items() if k.startswith(("OMP_", "GOMP_", "KMP_"))})
orc.build(native=True)
ref = orc.OracleScene(scenes.cbox_scene(res, res, spp, spp, spp, param="light_x"), [0])
n = 1
base = None
while n <= 2 * (os.cpu_count() or 1):
    orc.set_num_threads(n)
    ref.render_d(max_depth=3, seeds=(0, 0, 0))
    ts = []
    for r in range(3):
        t = time.perf_counter(); ref.render_d(max_depth=3, seeds=(r + 1,) * 3); ts.append(time.perf_counter() - t)
    dt = sorted(ts)[1]
    base = base or dt
    print("%4d threads  %.3f s  %.3f Msamples/s  speed-up %.1f" % (n, dt, res * res * spp / dt / 1e6, base / dt))
    n *= 2
