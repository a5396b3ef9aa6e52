"""In-tree build of the two native pieces (no pip, no JIT cache):

  psdr_jit_amd/lib/libpsdr_hip.so   hipcc --offload-arch=gfx950: the kernels + C ABI (include/psdr_hip.h)
  psdr_jit_amd/_psdr_core*.so        g++ + pybind11: host scene model, links the C ABI

hipcc cross-compiles gfx950 without a GPU.  A library is rebuilt when the SHA-256 over its sources, headers and compiler flags
differs from the signature stored beside it (lib/*.sig) - not by modification times, which do not survive the copy to the GPU
box - and build_all() checks that the loaded library reports the ABI version of include/psdr_hip.h.
"""
import hashlib
import os
import re
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
HIP_LIB = os.path.join(LIBDIR, "libpsdr_hip.so")
CORE_LIB = os.path.join(HERE, "_psdr_core" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))

API_SRC = os.path.join(CSRC, "hip", "api.hip")                  # the render entry points and every kernel (compiled as seven units, see _compile_hip)
SCENE_SRC = os.path.join(CSRC, "hip", "scene_build.hip")        # psdr_hip_scene_create / _update: tree build and refit, blob layout, uploads
HIP_SRCS = [API_SRC, SCENE_SRC]
BUILD_DEPS = [os.path.join(HERE, "isa_lint.py")]          # part of the recipe: a change of the lint re-builds (and re-lints) the library
_H = lambda *names: [os.path.join(CSRC, "hip", f) for f in names]
COMMON_DEPS = _H("scene_obj.h", "scene_dev.h", "dmath.h", "trav4.h") + [os.path.join(ROOT, "include", "psdr_hip.h"), os.path.join(CSRC, "common", "threads.h")]
API_DEPS = COMMON_DEPS + _H("sampler.h", "shade.h", "edges.h", "paths.h", "adjoint.h", "adjoint_mat.h", "microfacet.h") + [os.path.join(CSRC, "common", "envmath.h")] + BUILD_DEPS
SCENE_DEPS = COMMON_DEPS + _H("bvh.h", "filter.h") + [os.path.join(CSRC, "host", "hnum.h")] + BUILD_DEPS
HIP_DEPS = sorted(set(API_DEPS + SCENE_DEPS))
HOST_SRCS = [os.path.join(CSRC, "host", f) for f in ("scene_host.cpp", "bindings.cpp", "exr_piz.cpp")]
HOST_DEPS = [os.path.join(CSRC, "host", f) for f in ("scene_host.h", "hnum.h", "exr_piz.h")] + [os.path.join(ROOT, "include", "psdr_hip.h"), os.path.join(CSRC, "common", "envmath.h"), os.path.join(CSRC, "common", "threads.h")]

# -ffp-contract=off: every fused multiply-add in the kernels is an explicit fma so that the
# arithmetic matches the scalar CPU restatement the parity tests compare against.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-slp-vectorize"]


def _signature(files, flags):
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(hashlib.sha256(fh.read()).digest())
    h.update(("\0".join(flags)).encode())
    return h.hexdigest()


def _sig_path(target):
    return os.path.join(LIBDIR, os.path.basename(target) + ".sig")


def _stale(target, deps, flags=()):
    """True when `target` is missing or was built from other sources / with other flags than `deps` / `flags`"""
    if not os.path.exists(target) or not os.path.exists(_sig_path(target)):
        return True
    with open(_sig_path(target)) as fh:
        return fh.read().strip() != _signature(deps, flags)


def _stamp(target, sig):
    """`sig`: the signature taken BEFORE the compilers read the sources (an edit made while they run must leave the target stale)"""
    with open(_sig_path(target), "w") as fh:
        fh.write(sig + "\n")


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return r.stdout


N_KERNEL_UNITS = 8      # api.hip's PSDR_TU1..8: the heavy kernel templates of each scene class
UNIT_CLASS_BIT = {1: 1, 2: 1, 6: 1, 8: 1, 3: 2, 4: 4, 7: 4, 5: 8}      # the PSDR_CLS_MASK bit of the scene class a unit instantiates


def build_hip(force=False, extra_flags=(), target=None):
    """api.hip is compiled as seven translation units in parallel - the host code with the small kernels (-DPSDR_SPLIT) and six units
    that only instantiate the heavy kernel templates of one scene class (-DPSDR_TU=k) -, scene_build.hip as an eighth, and all are linked
    into one library: ~4 minutes of wall time instead of ~10 for the single unit (PSDR_BUILD_JOBS=1 compiles them one after the other)."""
    os.makedirs(LIBDIR, exist_ok=True)
    flags = [f for f in HIP_FLAGS if f != "-shared"] + list(extra_flags)
    if target is not None:
        # a development variant (tools/variants.py): its own library and object directory, so that several can be built side by side
        return _compile_hip(flags, target, os.path.join(os.path.dirname(target), "obj"))
    if force or _stale(HIP_LIB, HIP_SRCS + HIP_DEPS, flags):
        _compile_hip(flags, HIP_LIB, os.path.join(LIBDIR, "obj"))
    return HIP_LIB


# The allocator of clang 22 / ROCm 7.2 can place vector instructions it inserts at the head of a join block (re-materialised constants,
# split copies, reloads) in FRONT of the `s_or_b64 exec` that switches the other branch's lanes back on - those lanes then go on with a stale
# register.  It happens when its scalar phase has left copies in front of the exec restore; it cost round 3 its open item (the class-2
# reverse sweep, 0.5-2.5 % off in builds whose only difference was an unrelated knob; LABNOTES.md section 4).  isa_lint.py finds the pattern
# in the ISA; a unit that shows it is compiled again with the scalar allocator that does not split live ranges (-sgpr-regalloc=basic: the
# copies become SGPR spills, which the compiler does recognise as block prologue; measured +4 % kernel time, so it is not the default),
# and a unit that still shows it fails the build.
LINT_FALLBACK_FLAGS = ["-mllvm", "-sgpr-regalloc=basic"]


def _load_lint():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_psdr_isa_lint", os.path.join(HERE, "isa_lint.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _lint_units(hipcc, flags, units, objdir):
    lint = _load_lint()
    report = os.path.join(objdir, "lint.txt")
    if os.environ.get("PSDR_BUILD_NO_LINT"):          # development only: keeps a flagged unit as the compiler made it (to show what the lint is for, tools/variants.py)
        with open(report, "w") as fh:
            fh.write("lint skipped (PSDR_BUILD_NO_LINT)\n")
        return
    if not lint.available():
        # the lint is the only guard against the allocator defect described above: a build that cannot run it does not ship silently
        raise RuntimeError("build failed: llvm-objdump not found (looked in $HIPCC's directory, $ROCM_PATH, /opt/rocm and PATH), the ISA lint of the kernels "
                           "cannot run; PSDR_BUILD_NO_LINT=1 builds without it (development only)")
    lines = []
    for name, src, defs, _deps in units:
        obj = os.path.join(objdir, "api_%s.o" % name)
        found = lint.lint(obj)
        if found:
            lines.append("%s: %d join block(s) with vector instructions ahead of the exec restore (%s ...): compiled again with %s" %
                         (name, len(found), found[0][0][:60], " ".join(LINT_FALLBACK_FLAGS)))
            _run([hipcc] + flags + LINT_FALLBACK_FLAGS + defs + ["-c", src, "-o", obj])
            again = lint.lint(obj)
            if again:
                raise RuntimeError("build failed: unit %s still has vector instructions ahead of an exec restore with %s:\n%s" %
                                   (name, " ".join(LINT_FALLBACK_FLAGS), "\n".join("%s +0x%x: %s" % (n, o, "; ".join(x[:4])) for n, o, x in again)))
        else:
            lines.append("%s: clean" % name)
    with open(report, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    for l in lines:
        if not l.endswith("clean"):
            sys.stderr.write("psdr_jit_amd.build: " + l + "\n")


def _compile_hip(flags, target, objdir):
    """compiles and links api.hip with `flags` into `target`; objects go to `objdir`"""
    sig = _signature(HIP_SRCS + HIP_DEPS, flags)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(objdir, exist_ok=True)
    # development builds (-DPSDR_CLS_MASK=m: the host code launches the kernels of those scene classes only) skip the other classes' units
    mask = 15
    for f in flags:
        if f.startswith("-DPSDR_CLS_MASK="):
            mask = int(f.split("=")[1])
    units = [("main", API_SRC, ["-DPSDR_SPLIT"], API_DEPS)] + [("tu%d" % k, API_SRC, ["-DPSDR_TU=%d" % k], API_DEPS) for k in (1, 6, 8, 2, 4, 7, 5, 3) if UNIT_CLASS_BIT[k] & mask] + \
            [("scene", SCENE_SRC, [], SCENE_DEPS)]
    jobs = max(1, int(os.environ.get("PSDR_BUILD_JOBS", str(min(len(units), os.cpu_count() or 1)))))
    objs, pending, running = [], [], []
    # an object is compiled again only when ITS sources, headers or flags changed (signature beside it): editing the scene-build unit or
    # bvh.h leaves the six kernel units - minutes of compile time - alone
    usigs = {}
    for name, src, defs, deps in units:
        obj = os.path.join(objdir, "api_%s.o" % name)
        objs.append(obj)
        usigs[name] = _signature([src] + deps, flags + defs)
        fresh = False
        if os.path.exists(obj) and os.path.exists(obj + ".sig"):
            with open(obj + ".sig") as fh:
                fresh = fh.read().strip() == usigs[name]
        if not fresh:
            for f in (obj, obj + ".sig"):
                if os.path.exists(f):
                    os.remove(f)            # objects of an earlier (possibly failed, possibly differently flagged) build never get linked
            pending.append((name, src, defs, deps))
    built = [u[0] for u in pending]
    while pending or running:
        while pending and len(running) < jobs:
            name, src, defs, _deps = pending.pop(0)
            obj = os.path.join(objdir, "api_%s.o" % name)
            cmd = [hipcc] + flags + defs + ["-c", src, "-o", obj]
            running.append((subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), cmd))
        proc, cmd = running.pop(0)
        out, _ = proc.communicate()
        if proc.returncode != 0:
            for q, _c in running:
                q.kill()
                q.wait()
            for name in built:
                o = os.path.join(objdir, "api_%s.o" % name)
                if os.path.exists(o):
                    os.remove(o)
            sys.stderr.write(out)
            raise RuntimeError("build failed: " + " ".join(cmd))
    _lint_units(hipcc, flags, units, objdir)
    for name in built:                           # stamped after the lint: a unit the lint compiled again keeps its signature (same sources, the fallback flags are the recipe's)
        with open(os.path.join(objdir, "api_%s.o.sig" % name), "w") as fh:
            fh.write(usigs[name] + "\n")
    arch = [f for f in flags if f.startswith("--offload-arch")]
    _run([hipcc] + arch + ["-shared", "-fPIC"] + objs + ["-o", target])
    if target == HIP_LIB:
        _stamp(HIP_LIB, sig)
    else:
        with open(target + ".sig", "w") as fh:
            fh.write(sig + "\n")
    return target


CORE_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-pthread"]


def build_core(force=False, hip_flags=()):
    build_hip(extra_flags=hip_flags)
    deps = HOST_SRCS + HOST_DEPS
    if force or _stale(CORE_LIB, deps, CORE_FLAGS):
        sig = _signature(deps, CORE_FLAGS)
        import pybind11
        inc = ["-I" + sysconfig.get_paths()["include"], "-I" + pybind11.get_include()]
        _run(["g++"] + CORE_FLAGS + inc + HOST_SRCS + ["-o", CORE_LIB, "-L" + LIBDIR, "-lpsdr_hip", "-Wl,-rpath,$ORIGIN/lib"])
        _stamp(CORE_LIB, sig)
    return CORE_LIB


def header_abi_version():
    with open(os.path.join(ROOT, "include", "psdr_hip.h")) as fh:
        m = re.search(r"#define\s+PSDR_HIP_ABI_VERSION\s+(\d+)", fh.read())
    return int(m.group(1))


def check_abi():
    """the library that will be loaded is the one this tree's header describes"""
    import ctypes
    L = ctypes.CDLL(HIP_LIB)
    got, want = int(L.psdr_hip_abi_version()), header_abi_version()
    if got != want:
        raise RuntimeError("libpsdr_hip.so reports ABI version %d, include/psdr_hip.h says %d: stale binary" % (got, want))


def build_all(force=False, hip_flags=None):
    """hip_flags: extra hipcc flags (development builds, e.g. -DPSDR_CLS_MASK=4); default: $PSDR_HIP_FLAGS split on blanks"""
    if hip_flags is None:
        hip_flags = tuple(os.environ.get("PSDR_HIP_FLAGS", "").split())
    build_hip(force, extra_flags=hip_flags)
    build_core(force, hip_flags=hip_flags)
    check_abi()
    return HIP_LIB, CORE_LIB


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv))
