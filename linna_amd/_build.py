"""Build liblinna_hip.so (hipcc, gfx950) in-tree.  Used by __graft_entry__.build()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblinna_hip.so")
SOURCES = ["gemm.hip", "pointwise.hip", "net_stream.hip", "autocorr.hip", "api.hip", "comm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "linna_hip.h")]
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + os.environ.get("LINNA_HIPCC_EXTRA", "").split() + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def build_stamps(extra=("-DNS_STAMPS",), name="liblinna_hip_stamps.so", verbose=True, base=True, source="net_stream.hip"):
    """Diagnostic variant next to the product library: ONE source (net_stream.hip by default) compiled with extra
    definitions -- phase stamps (every launch of the whole-network kernel then needs LINNA_FUSED_STAMPS), experiment
    switches -- the other objects shared.  Load it with LINNA_LIB_PATH."""
    if base:
        build(verbose=verbose)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    o = os.path.join(CSRC, "%s_%s.o" % (os.path.splitext(source)[0], os.path.splitext(name)[0].replace("liblinna_hip_", "")))
    src = os.path.join(CSRC, source)
    if _stale(o, [src, os.path.join(CSRC, "common.h")]) or True:
        cmd = [hipcc] + FLAGS + list(extra) + ["-c", src, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in SOURCES if s != source] + [o]
    lib = os.path.join(HERE, name)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"])
    return lib


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        src = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--source=")]
        build_stamps(tuple(a for a in sys.argv[1:] if a.startswith("-D")) or ("-DNS_STAMPS",), source=src[0] if src else "net_stream.hip")
    else:
        build(force="--force" in sys.argv)
