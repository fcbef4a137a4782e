"""Build libkalle_hip.so (gfx950 code objects + C-ABI host stubs) in-tree with hipcc.

`python -m kalle_audio_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libkalle_hip.so")
SOURCES = ["gemm.hip", "gemm2.hip", "norm.hip", "elementwise.hip", "attention.hip", "conv1d.hip", "conv1d_bwd.hip",
           "llasa.hip", "conformer.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffast-math", "-fno-finite-math-only",
         "-Wno-unused-value"]


def source_stamp():
    """sha256 (first 16 hex digits) over the kernel sources and the C-ABI header: names the code a profile was taken from on
    a box that has no git history (profiles/traffic_rNN.json carries it, bench.py compares it with the running tree)"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(HERE, "..", "include", "kalle_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _needs(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose=True, force=False, extra_flags=(), out=None, objdir_name="build", only=None):
    """extra_flags / out / objdir_name / only: a second library next to the product one for same-box A/B runs of build-time
    switches (python -m kalle_audio_amd.build --variant NAME -DKALLE_...=...; run with KALLE_LIB_PATH=.../libkalle_hip_NAME.so)"""
    global OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, objdir_name)
    out = out or OUT
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(HERE, "..", "include", "kalle_hip.h")]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _needs(obj, [src] + hdrs):
            jobs.append([hipcc] + FLAGS + list(extra_flags) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in srcs]
    if jobs or not os.path.exists(out):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:
        name = sys.argv[sys.argv.index("--variant") + 1]
        flags = [a for a in sys.argv[1:] if a.startswith("-D")]
        print(build(force=True, extra_flags=flags, out=os.path.join(HERE, f"libkalle_hip_{name}.so"), objdir_name="build_" + name))
    else:
        print(build(force="--force" in sys.argv))
