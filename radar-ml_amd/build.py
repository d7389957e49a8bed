"""Build libradarml_hip.so (hand-written HIP kernels for gfx950) in-tree with hipcc.

    python radar-ml_amd/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to the
GPU box with the gpurun snapshot.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libradarml_hip.so")
ARCH = "gfx950"
FORCE_ALL = False


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, variant=None, defines=(), only=()):
    """Compile every csrc/*.hip for gfx950 into one shared library.  Returns its path.
    ``variant``/``defines`` build an experimental copy libradarml_hip_<variant>.so with extra -D flags; with ``only`` (source
    file names) just those sources are compiled with the flags and the other objects come from the main build."""
    global LIB
    if variant:
        lib_out = os.path.join(HERE, "libradarml_hip_%s.so" % variant)
        return _build(lib_out, os.path.join(HERE, "build", variant), verbose, ["-D" + d for d in defines], only=tuple(only))
    global FORCE_ALL
    if not force and not is_stale():
        return LIB
    FORCE_ALL = bool(force)
    return _build(LIB, os.path.join(HERE, "build"), verbose, [])


def _build(LIB, objdir, verbose, extra, only=()):
    os.makedirs(objdir, exist_ok=True)
    maindir = os.path.join(HERE, "build")
    hipcc = _hipcc()
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + list(extra)
    objs = []
    procs = []
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    hdr_t = max([os.path.getmtime(h) for h in hdrs] + [0.0])
    stamp = os.path.join(objdir, "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if only and os.path.basename(src) not in only:
            objs.append(os.path.join(maindir, os.path.basename(src) + ".o"))     # unchanged source: the main build's object
            continue
        objs.append(obj)
        # per-object staleness: project.hip alone takes minutes, the others seconds
        if same_flags and not FORCE_ALL and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            continue
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
    with open(stamp, "w") as f:
        f.write(" ".join(flags))
    tmp = LIB + ".tmp"
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stdout.decode(errors="replace"))
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        rest = sys.argv[i + 2:]
        only = [a for a in rest if a.endswith(".hip")]
        print(build(variant=sys.argv[i + 1], defines=[a for a in rest if not a.endswith(".hip")], only=only, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
