"""Kernel metadata of the code objects inside liblinna_hip.so (or one .o): registers, scratch, LDS per kernel.
Test / diagnostic helper -- the clang offload bundles of the `.hip_fatbin` section are unpacked by hand (roc-obj-ls needs
a perl module this image lacks) and their AMDGPU metadata notes read with llvm-readelf.
usage: python tests/codeobj.py [file]   (a test helper: tests/test_abi.py reads every kernel's scratch size with it)"""
import os, re, struct, subprocess, sys, tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path):
    """(triple, bytes) of every device code object bundled in `path`."""
    data = open(path, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return out
        n, = struct.unpack_from("<Q", data, i + len(MAGIC))
        p = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "amdgcn" in triple and size:
                out.append((triple, data[i + off:i + off + size]))
        pos = i + len(MAGIC)


def kernels(path):
    """[{name, vgpr, sgpr, scratch, lds}] over every gfx code object in `path`."""
    res = []
    for triple, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob); f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        cur = {}
        for line in txt.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip().strip("'\"")
            if k == "agpr_count" and cur.get("name"):           # (first key of a kernel's map in llvm's YAML order)
                res.append(cur); cur = {}
            if k in ("name", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "symbol"):
                cur[k] = v
            if k == "wavefront_size" and "symbol" in cur:
                res.append(cur); cur = {}
        if cur.get("symbol"):
            res.append(cur)
    out = []
    for k in res:
        if "symbol" in k and k["symbol"].endswith(".kd"):
            out.append(dict(name=k.get("name", k["symbol"]), vgpr=int(k.get("vgpr_count", 0)), sgpr=int(k.get("sgpr_count", 0)),
                            scratch=int(k.get("private_segment_fixed_size", 0)), lds=int(k.get("group_segment_fixed_size", 0))))
    return out


if __name__ == "__main__":
    p = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblinna_hip.so")
    ks = kernels(p)
    for k in sorted(ks, key=lambda k: (-k["scratch"], -k["vgpr"])):
        print("%5d B scratch  %3d vgpr  %3d sgpr  %6d B lds  %s" % (k["scratch"], k["vgpr"], k["sgpr"], k["lds"], k["name"]))
    print("%d kernels" % len(ks))
