"""Builds libpcr_hip.so (every HIP source under csrc/) for gfx950 with hipcc, in-tree."""
import glob
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(PKG))
CSRC = os.path.join(os.path.dirname(PKG), "csrc")
LIBDIR = os.path.join(PKG, "lib")
# PCR_LIB_TAG (diagnostics): a second library beside the product one, e.g. PCR_LIB_TAG=tune with
# PCR_EXTRA_HIPCC_FLAGS=-DPCR_TUNING=1 builds lib/libpcr_hip_tune.so, which _lib.load() picks up under the same variable
_TAG = os.environ.get("PCR_LIB_TAG", "")
SO = os.path.join(LIBDIR, "libpcr_hip%s.so" % ("_" + _TAG if _TAG else ""))

# point_ops.hip must not contract a*b+c into fma (bit-exact index outputs); the MFMA model
# kernels keep the default.
FLAGS = {
    "point_ops.hip": ["-ffp-contract=off"] + os.environ.get("PCR_POINT_FLAGS", "").split(),
    "edge_kernels.hip": ["-ffp-contract=off"],
}


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm at /opt/rocm)")
    return exe


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def includes_of(src, _seen=None):
    """src and every project header it includes, transitively (`#include "x.h"` resolved against csrc/ and include/)"""
    import re
    seen = _seen if _seen is not None else {}
    if src in seen:
        return list(seen)
    seen[src] = True
    with open(src) as f:
        for name in re.findall(r'^\s*#\s*include\s*"([^"]+)"', f.read(), flags=re.M):
            for d in (os.path.dirname(src), CSRC, os.path.join(ROOT, "include")):
                cand = os.path.join(d, name)
                if os.path.exists(cand):
                    includes_of(cand, seen)
                    break
    return list(seen)


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "pcr.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj" + ("_" + _TAG if _TAG else ""))
    os.makedirs(objdir, exist_ok=True)
    common = ["--offload-arch=gfx950", "-O3", "-fPIC", "-fvisibility=hidden", "-std=c++17",
              "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    common += os.environ.get("PCR_EXTRA_HIPCC_FLAGS", "").split()      # diagnostic builds only
    objs = []
    procs = []
    # the objects of a directory were all built with ONE flag set, recorded beside them: another set (a tagged library
    # first built without -DPCR_TUNING=1 and later with it, say) rebuilds everything instead of reusing stale objects
    flagset = " ".join(common + ["|"] + ["%s:%s" % (k, " ".join(v)) for k, v in sorted(FLAGS.items())])
    stamp = os.path.join(objdir, "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == flagset
    if not same_flags and os.path.exists(stamp):
        os.remove(stamp)
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if not force and same_flags and os.path.exists(obj) and \
                all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in includes_of(src)):
            continue        # incremental: this object is newer than its source and every header
        cmd = [hipcc()] + common + FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
    with open(stamp, "w") as f:
        f.write(flagset)
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs)
    return SO


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
