"""Build libgoofer_hip.so in-tree for gfx950 (``python -m goofer_amd.build``).

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the
repo snapshot.  No CPU fallback exists: if this library is missing the package raises on first use.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libgoofer_hip.so")
SOURCES = ["api.hip", "fft.hip", "pulse.hip", "binops.hip", "samples.hip", "assemble.hip", "stems.hip", "analysis.hip", "jitter.hip", "post.hip", "resample.hip", "planner.hip"]
HOST_ONLY = ("planner.hip",)      # host code in the library: no kernel, cannot change what a counter pass measures
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-align-mismatch"]


def source_hash() -> str:
    """sha256 over the kernel sources the library is built from (goofer_amd/csrc/*.hip, *.h without the host-only files, and include/goofer_hip.h, names
    and contents in sorted order).  The counter files under profiles/ carry the hash of the tree they were measured on;
    bench.py prints their numbers only beside the same hash."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) and f not in HOST_ONLY)
    for f in files + [os.path.join("..", "..", "include", "goofer_hip.h")]:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "goofer_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return OUT
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = ["hipcc", *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or (verbose and out.strip()):
            sys.stderr.write("== %s ==\n%s\n" % (src, out))
        failed |= p.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs], check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
