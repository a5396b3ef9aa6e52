"""Measurement variants of libpsdr_hip.so (development builds with -D knobs), built here and timed on the GPU box in one gpurun call.

    python tools/variants.py build NAME "-DPSDR_CLS_MASK=4 -DPSDR_SHADE_MIN=36"     # -> _dev/variants/NAME/{libpsdr_hip.so, .sig, flags}
    python tools/variants.py run NAME python bench.py --config 5 ...                # on the GPU box: puts the variant in place and runs the command
    python tools/variants.py restore                                                # puts the default library back (saved by the first `run`)
    python tools/variants.py stage NAME                                             # -> _dev/variants/NAME/pkg/psdr_jit_amd: a private copy of the package around the
                                                                                    #    variant library (a subprocess with that directory first on sys.path loads it;
                                                                                    #    the in-tree library - possibly mapped by the calling process - is not touched)
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


def stage(name):
    """private package copy around the variant library; returns the directory to put first on sys.path"""
    d = os.path.join(VDIR, name)
    pkg_src = os.path.join(ROOT, "psdr_jit_amd")
    top = os.path.join(d, "pkg")
    pkg = os.path.join(top, "psdr_jit_amd")
    if os.path.exists(top):
        shutil.rmtree(top)
    os.makedirs(os.path.join(pkg, "lib"))
    for f in os.listdir(pkg_src):
        src = os.path.join(pkg_src, f)
        if os.path.isfile(src) and (f.endswith(".py") or f.endswith(".so")):
            shutil.copy2(src, pkg)
    for f in os.listdir(os.path.join(pkg_src, "lib")):
        if f.startswith("_psdr_core") and f.endswith(".sig"):
            shutil.copy2(os.path.join(pkg_src, "lib", f), os.path.join(pkg, "lib"))
    shutil.copy2(os.path.join(d, "libpsdr_hip.so"), os.path.join(pkg, "lib"))
    shutil.copy2(os.path.join(d, "libpsdr_hip.so.sig"), os.path.join(pkg, "lib"))
    return top


def build_variant(name, flags):
    d = os.path.join(VDIR, name)
    os.makedirs(d, exist_ok=True)
    load_build().build_hip(extra_flags=tuple(flags), target=os.path.join(d, "libpsdr_hip.so"))      # (its own object directory: variants build side by side)
    with open(os.path.join(d, "flags"), "w") as fh:
        fh.write(" ".join(flags) + "\n")
    return d


def variant_is_current(name, flags):
    """the variant library exists and was built from the sources and flags of this tree"""
    b = load_build()
    d = os.path.join(VDIR, name)
    lib, sig = os.path.join(d, "libpsdr_hip.so"), os.path.join(d, "libpsdr_hip.so.sig")
    if not (os.path.exists(lib) and os.path.exists(sig)):
        return False
    want = b._signature(b.HIP_SRCS + b.HIP_DEPS, [f for f in b.HIP_FLAGS if f != "-shared"] + list(flags))
    with open(sig) as fh:
        return fh.read().strip() == want


if __name__ != "__main__":
    cmd = None
else:
    cmd = sys.argv[1]
if cmd == "build":
    build_variant(sys.argv[2], tuple(sys.argv[3].split()))
elif cmd == "run":
    name = sys.argv[2]
    d = os.path.join(VDIR, name)
    keep = os.path.join(VDIR, "_default")
    if not os.path.exists(keep) and os.path.exists(SIG):
        os.makedirs(keep)
        shutil.copy2(LIB, keep); shutil.copy2(SIG, keep)
    put(d)
    env = dict(os.environ)
    env["PSDR_HIP_FLAGS"] = open(os.path.join(d, "flags")).read().strip()
    sys.exit(subprocess.call(sys.argv[3:], env=env, cwd=ROOT))
elif cmd == "stage":
    print(stage(sys.argv[2]))
elif cmd == "restore":
    put(os.path.join(VDIR, "_default"))
