"""Builds libshapegan_hip.so (the C-ABI HIP library) in-tree for gfx950.

    python -m shapegan_amd.build            # incremental
    python -m shapegan_amd.build --force

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libshapegan_hip.so")
SOURCES = ["conv3d.hip", "conv3d_halo.hip", "conv3d_edge.hip", "gemm.hip", "sdfnet.hip", "batchnorm.hip", "elementwise.hip", "pointnet.hip", "losses.hip", "sdf_batch.hip", "head.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "mfma_tile.h"), os.path.join(CSRC, "conv_common.h"),
           os.path.join(HERE, "..", "include", "shapegan_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + HEADERS):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    build_comm(force, verbose)
    build_cpu(force, verbose)
    return LIB


CPU_LIB = os.path.join(HERE, "libshapegan_cpu.so")


def build_cpu(force=False, verbose=True):
    """libshapegan_cpu.so: the plain-C++ twin of the C ABI (csrc_cpu/shapegan_cpu.cpp), g++ + OpenMP, no GPU code."""
    src = os.path.join(HERE, "csrc_cpu", "shapegan_cpu.cpp")
    if not (force or _stale(CPU_LIB, [src])):
        return CPU_LIB
    cmd = [os.environ.get("CXX", "g++"), "-O3", "-fopenmp", "-fPIC", "-shared", "-std=c++17", "-Wall", src, "-o", CPU_LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libshapegan_cpu.so failed:\n%s\n%s" % (r.stdout, r.stderr))
    return CPU_LIB


COMM_LIB = os.path.join(HERE, "libshapegan_comm.so")


def build_comm(force=False, verbose=True):
    """libshapegan_comm.so: the RCCL gradient exchange of the C ABI (csrc/comm.cpp); RCCL itself is bound at run time."""
    src = os.path.join(CSRC, "comm.cpp")
    if not (force or _stale(COMM_LIB, [src, HEADERS[-1]])):
        return COMM_LIB
    # host code only: g++, no -lrccl (RCCL is bound at run time, csrc/comm.cpp) and no RUNPATH into the toolchain's ROCm — the HIP
    # runtime it needs is the one the host process (PyTorch) has already mapped
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-D__HIP_PLATFORM_AMD__",
           "-I/opt/rocm/include", src, "-o", COMM_LIB, "-L/opt/rocm/lib", "-lamdhip64", "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libshapegan_comm.so failed:\n%s\n%s" % (r.stdout, r.stderr))
    return COMM_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
