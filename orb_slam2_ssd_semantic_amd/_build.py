"""In-tree build of liborbfe.so (HIP kernels + C-ABI) for gfx950 with hipcc.

`python -m orb_slam2_ssd_semantic_amd._build` or `_build.build()`.  hipcc cross-compiles without a
GPU; the resulting .so travels to the GPU box with the repo snapshot (it is git-ignored only).
"""
import os
import shutil
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
LIB = os.path.join(_PKG, "liborbfe.so")
SOURCES = ["orbfe_kernels.hip", "orbfe_api.hip", "orbfe_match.hip", "orbfe_io.hip", "orbfe_group.hip", "orbfe_hostgeom.hip",
           "orbfe_pipeline.hip"]
HEADERS = ["orbfe_common.h", "orbfe_kernels.h", "orbfe_pattern.inc", "orbfe_fast_body.inc", "orbfe_fast_body_u.inc", "orbfe_fast_body_c.inc", "orbfe_pyr_body.inc", os.path.join(_ROOT, "include", "orbfe.h")]
ARCH = "gfx950"
# -ffp-contract=off: no FMA contraction anywhere (outputs must be bit-exact)
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"--offload-arch={ARCH}", "-Wall",
         "-Wno-unused-function", "-Wno-unused-variable"]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: liborbfe.so cannot be built (there is no CPU fallback)")


FLAGFILE = LIB + ".flags"   # the flag set the .so was built with (a variant left by an aborted A/B run must not pass for the default)


def _flagset():
    return " ".join(FLAGS + os.environ.get("ORBFE_EXTRA_FLAGS", "").split())


def _stale():
    if not os.path.exists(LIB):
        return True
    try:
        if open(FLAGFILE).read().strip() != _flagset():
            return True
    except OSError:
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h)
                                                       for h in HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(tag, extra_flags):
    """A variant library next to the default one: ab/liborbfe_<tag>.so built with extra -D flags (ab/ is git-ignored and travels
    to the GPU box); select it at run time with ORBFE_LIB=<path>.  The default library is not touched."""
    out = os.path.join(_ROOT, "ab", f"liborbfe_{tag}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    return build(force=True, out_path=out, extra=list(extra_flags), bdir=os.path.join(_PKG, "build", "ab_" + tag))


def build(force=False, verbose=False, out_path=None, extra=None, bdir=None):
    variant = out_path is not None
    if not variant and not force and not _stale():
        return LIB
    cc = hipcc()
    if not variant and os.path.exists(FLAGFILE):
        os.remove(FLAGFILE)   # no sidecar while a build is in flight: an aborted build is stale
    if extra is None:
        extra = os.environ.get("ORBFE_EXTRA_FLAGS", "").split()   # developer A/B builds on the GPU box (e.g. -DQT_MIN_WAVES=7)
    objs = []
    bdir = bdir or os.path.join(_PKG, "build")
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for s in SOURCES:
        o = os.path.join(bdir, s.replace(".hip", ".o"))
        cmd = [cc, *FLAGS, *extra, "-I", os.path.join(_ROOT, "include"), "-I", CSRC, "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out_path or LIB, *objs, "-ldl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    if variant:
        return out_path
    with open(FLAGFILE, "w") as f:
        f.write(_flagset() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
