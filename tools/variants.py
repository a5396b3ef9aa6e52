"""Measurement variants of libpsdr_hip.so (development builds with -D knobs), built here and timed on the GPU box in one gpurun call.

    python tools/variants.py build NAME "-DPSDR_CLS_MASK=4 -DPSDR_SHADE_MIN=36"     # -> _dev/variants/NAME/{libpsdr_hip.so, .sig, flags}
    python tools/variants.py run NAME python bench.py --config 5 ...                # on the GPU box: puts the variant in place and runs the command
    python tools/variants.py restore                                                # puts the default library back (saved by the first `run`)
_dev/ is git-ignored and travels with the gpurun snapshot."""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "_dev", "variants")
LIB = os.path.join(ROOT, "psdr_jit_amd", "lib", "libpsdr_hip.so")
SIG = LIB + ".sig"


def load_build():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_psdr_build", os.path.join(ROOT, "psdr_jit_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def put(src_dir):
    shutil.copy2(os.path.join(src_dir, "libpsdr_hip.so"), LIB)
    shutil.copy2(os.path.join(src_dir, "libpsdr_hip.so.sig"), SIG)


cmd = sys.argv[1]
if cmd == "build":
    name, flags = sys.argv[2], tuple(sys.argv[3].split())
    d = os.path.join(VDIR, name)
    os.makedirs(d, exist_ok=True)
    keep = os.path.join(VDIR, "_default")
    if not os.path.exists(keep) and os.path.exists(SIG):
        os.makedirs(keep)
        shutil.copy2(LIB, keep); shutil.copy2(SIG, keep)
    load_build().build_hip(extra_flags=flags)
    shutil.copy2(LIB, d); shutil.copy2(SIG, d)
    with open(os.path.join(d, "flags"), "w") as fh:
        fh.write(" ".join(flags) + "\n")
    if os.path.exists(keep):
        put(keep)
elif cmd == "run":
    name = sys.argv[2]
    d = os.path.join(VDIR, name)
    put(d)
    env = dict(os.environ)
    env["PSDR_HIP_FLAGS"] = open(os.path.join(d, "flags")).read().strip()
    sys.exit(subprocess.call(sys.argv[3:], env=env, cwd=ROOT))
elif cmd == "restore":
    put(os.path.join(VDIR, "_default"))
