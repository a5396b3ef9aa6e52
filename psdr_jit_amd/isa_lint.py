"""ISA lint for libpsdr_hip.so and its objects: vector instructions ahead of the lane re-enable of a control-flow join.
Part of the build recipe (build.py lints every translation unit and re-compiles a flagged one with the scalar allocator that does not
produce the pattern) and of the CPU test suite (tests/test_isa_lint.py).

The compiler bug this looks for (ROCm 7.2 / clang 22, AMDGPU split register allocation; LABNOTES.md section 4, "the class-2 sweep's open
item, closed"): the exec restore of a join block (`s_or_b64 exec, exec, s[a:b]`, SI_END_CF) is meant to be the block's first instruction,
and everything the VGPR allocator inserts at a block head (re-materialised constants, split copies, reloads from AGPRs or scratch) is
meant to go AFTER it.  When the SGPR allocation phase has put scalar copies in front of the restore, `SIInstrInfo::isBasicBlockPrologue`
stops at those copies and the vector instructions land BEFORE the restore: they run for the lanes of the fall-through branch only, and the
lanes that rejoin at this block keep whatever the register held before - a wrong value in a lane-dependent subset of the wave.

    python psdr_jit_amd/isa_lint.py [libpsdr_hip.so | api_tu4.o ...]      # exit code 1 when a kernel shows the pattern

A join block is recognised as the target of an `s_cbranch_execz` (the branch that skips a region when no lane wants it: when it is taken
exec is zero, so nothing of the program's own can sit between that label and the restore - a vector instruction there would do nothing on
the skipping path) or as the fall-through of an `s_cbranch_execnz` (the exit of a divergent loop, entered with exec == 0).  From the label the scan walks forward until the first instruction that writes exec, branches, or starts the next
block; if that instruction is `s_or_b64 exec, exec, <sgpr pair>` (or the `s_or_saveexec_b64` / `s_andn2_saveexec_b64` that flips a flow block to the lanes of the
other branch) and a vector instruction (v_*, ds_*, flat_/global_/scratch_/buffer_*)
came before it, the block is reported.  v_readlane / v_writelane (SGPR spill code, which the compiler does recognise as block prologue)
do not depend on exec and are ignored."""
import os
import re
import subprocess
import sys
import tempfile

def _find_objdump():
    """llvm-objdump of the toolchain that compiles the kernels: beside $HIPCC, under $ROCM_PATH, under /opt/rocm, then on PATH"""
    import shutil
    cands = []
    hipcc = os.environ.get("HIPCC")
    if hipcc:
        root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
        cands += [os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(root, "llvm", "bin", "llvm-objdump")]
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if root:
            cands += [os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(root, "llvm", "bin", "llvm-objdump")]
    for c in cands:
        if os.path.exists(c):
            return c
    return shutil.which("llvm-objdump")


OBJDUMP = _find_objdump()


def available():
    return OBJDUMP is not None and os.path.exists(OBJDUMP)

INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*[0-9A-Fa-f ]+(?:<([^>]+)>)?\s*$")
FUNC = re.compile(r"^([0-9a-f]+) <(.+)>:$")


def code_objects(lib, tmp):
    """the gfx950 code objects bundled in a host shared library (one per translation unit)"""
    import shutil
    local = os.path.join(tmp, "lib.so")          # (the bundles are extracted beside the file that is read)
    shutil.copy2(lib, local)
    out = subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
    if not objs:
        raise RuntimeError("no device code objects found in %s:\n%s" % (lib, out))
    return [os.path.join(tmp, f) for f in objs]


def is_vector(op):
    return op.startswith(("v_", "ds_", "global_", "flat_", "scratch_", "buffer_", "tbuffer_", "image_"))


def is_lane_op(op):
    return op.startswith(("v_readlane", "v_writelane", "v_readfirstlane"))


def is_allocator_kind(op):
    return op.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr_read", "v_accvgpr_write", "v_accvgpr_mov", "scratch_load", "scratch_store"))


def writes_exec(op, args):
    if "saveexec" in op or op.startswith("v_cmpx"):
        return True
    dst = args.split(",")[0].strip()
    return op.startswith("s_") and dst in ("exec", "exec_lo", "exec_hi")


def lint_object(path):
    dis = subprocess.run([OBJDUMP, "-d", path], stdout=subprocess.PIPE, text=True).stdout.split("\n")
    findings = []
    funcs = []          # (name, [(addr, op, args)])
    cur = None
    for line in dis:
        m = FUNC.match(line)
        if m:
            cur = (m.group(2), int(m.group(1), 16), [])
            funcs.append(cur)
            continue
        m = INS.match(line)
        if m and cur is not None:
            cur[2].append((int(m.group(3), 16), m.group(1), m.group(2), m.group(4)))
    for name, base, ins in funcs:
        # kernels end in s_endpgm; a device function the compiler did NOT inline (a large lambda called from several places) returns with
        # s_setpc_b64 - it is allocated by the same allocator and shows the same defect (round 5: the material sweep's bsdf_back)
        if not any(op in ("s_endpgm", "s_setpc_b64") for _a, op, _g, _t in ins):
            continue
        targets, skip_targets = set(), set()
        for addr, op, args, tgt in ins:
            if op.startswith(("s_cbranch", "s_branch")) and tgt and "+0x" in tgt:
                targets.add(base + int(tgt.rsplit("+0x", 1)[1], 16))
                if op == "s_cbranch_execz":
                    skip_targets.add(base + int(tgt.rsplit("+0x", 1)[1], 16))
        # A kernel too long for the 16-bit branch offset has its far branches relaxed into
        #     s_getpc_b64 s[a:b] ; s_add_u32 sa, sa, imm ; s_addc_u32 sb, sb, hi ; s_setpc_b64 s[a:b]
        # (target = the address after the s_getpc + imm).  A far `s_cbranch_execz L` becomes `s_cbranch_execnz <over the jump>` + that jump, so the
        # block behind the s_setpc is a join block exactly like an execz target - and was invisible to the scan without this.
        far = {}            # index of the s_getpc -> target address
        for k in range(len(ins) - 3):
            if ins[k][1] != "s_getpc_b64" or ins[k + 3][1] != "s_setpc_b64":
                continue
            pair = ins[k][2].replace(" ", "")
            lo, hi = ins[k + 1], ins[k + 2]
            if not (lo[1] in ("s_add_u32", "s_sub_u32") and hi[1] in ("s_addc_u32", "s_subb_u32") and ins[k + 3][2].replace(" ", "") == pair):
                continue
            try:
                imm_lo = int(lo[2].split(",")[2].strip(), 0)
                imm_hi = int(hi[2].split(",")[2].strip(), 0)
            except (ValueError, IndexError):
                continue
            off = (imm_hi << 32) | (imm_lo & 0xffffffff)
            if off >= 1 << 63:
                off -= 1 << 64
            if lo[1] == "s_sub_u32":
                off = -off
            far[k] = ins[k + 1][0] + off
            targets.add(far[k])
        # ... and the exit of a divergent loop: the block that follows the back edge `s_cbranch_execnz` is entered with exec == 0
        for k, (addr, op, args, tgt) in enumerate(ins[:-1]):
            if op == "s_cbranch_execnz":
                if (k + 1) in far:
                    skip_targets.add(far[k + 1])          # the relaxed form of `s_cbranch_execz <far>`
                else:
                    skip_targets.add(ins[k + 1][0])
            elif op == "s_cbranch_execz" and tgt and "+0x" in tgt:
                # the relaxed back edge of a divergent loop: `s_cbranch_execz <exit>` + far jump to the loop head; <exit> is already a skip target
                pass
        index = {a: i for i, (a, _o, _g, _t) in enumerate(ins)}
        for t in sorted(skip_targets):
            i = index.get(t)
            if i is None:
                continue
            seen = []
            while i < len(ins):
                addr, op, args, _tgt = ins[i]
                if i != index[t] and addr in targets:
                    break
                # the lane re-enable of a join (SI_END_CF) or the flip to the other branch's lanes at the head of a flow block (SI_ELSE)
                if (op == "s_or_b64" and args.replace(" ", "").startswith("exec,exec,s[")) or op in ("s_or_saveexec_b64", "s_andn2_saveexec_b64"):
                    if seen:
                        findings.append((name, t - base, seen))
                    break
                if writes_exec(op, args) or op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
                    break
                if is_vector(op) and not is_lane_op(op):
                    seen.append("%s %s" % (op, args))
                i += 1
    return findings


def lint(lib):
    with tempfile.TemporaryDirectory() as tmp:
        res = []
        for obj in code_objects(lib, tmp):
            res += lint_object(obj)
    return res


if __name__ == "__main__":
    libs = sys.argv[1:] or [os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpsdr_hip.so")]
    bad = 0
    for lib in libs:
        f = lint(lib)
        print("%s: %d block(s) with vector instructions ahead of the exec restore" % (lib, len(f)))
        by_kernel = {}
        for name, off, seen in f:
            by_kernel.setdefault(name, []).append((off, seen))
        for name, lst in by_kernel.items():
            print("  %s" % name)
            for off, seen in lst[:12]:
                print("    +0x%x: %s" % (off, "; ".join(seen[:6]) + (" ..." if len(seen) > 6 else "")))
        bad += len(f)
    sys.exit(1 if bad else 0)
