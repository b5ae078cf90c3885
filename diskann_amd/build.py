"""Build libdann_hip.so in-tree with hipcc for gfx950 (one object per translation unit,
compiled in parallel).  No torch dependency: the library is plain HIP + a C ABI."""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libdann_hip.so")
SOURCES = ["api.hip", "search_kernels.hip", "search_f32.hip", "search_f16.hip", "search_u8.hip", "search_i8.hip",
           "search_sq8.hip", "search_pq.hip", "search_pqlut.hip", "search_pqlut2.hip", "search_pqlut3.hip", "search_pqlut4.hip", "search_pair.hip", "server.hip", "sharded.hip", "paged_kernels.hip", "distance_kernels.hip", "build_kernels.hip", "pq_kernels.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fgpu-flush-denormals-to-zero=0" if False else "-fno-gpu-flush-denormals-to-zero", "-Wall",
         "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(not os.path.exists(d) or os.path.getmtime(d) > t for d in deps)


def _depfile_deps(dep):
    """prerequisites recorded by `hipcc -MD -MF` for one object (None if there is no usable depfile yet)"""
    try:
        text = open(dep).read()
    except OSError:
        return None
    text = text.replace("\\\n", " ")
    if ":" not in text:
        return None
    return [d for d in text.split(":", 1)[1].split() if not d.startswith("/opt/") and not d.startswith("/usr/")]


def build(force=False, verbose=False):
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "dann.h"))
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        deps = _depfile_deps(obj[:-2] + ".d")  # the headers this unit really includes; without a depfile: all of them
        if force or _stale(obj, [src] + (deps if deps is not None else headers)):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [_hipcc(), *FLAGS, "-MD", "-MF", obj[:-2] + ".d", "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)
        return obj

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(max(os.cpu_count() or 4, 4), len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or force or _stale(OUT, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
