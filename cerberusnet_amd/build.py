"""In-tree hipcc build of libcerberus_hip.so for gfx950 (MI355X).

    python -m cerberusnet_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but
travels to the GPU box with the repo snapshot.  No torch headers are involved:
the library is a plain C-ABI shared object (include/cerberus_hip.h).
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(CSRC, "_obj")
LIB = os.path.join(LIBDIR, "libcerberus_hip.so")
ARCH = "gfx950"
SOURCES = ["api.hip", "corr_generic.hip", "corr_d4.hip", "corr_d4_bwd.hip", "corr_strip.hip", "corr_coarse.hip", "corr_fwd_pipe.hip", "corr_mfma.hip", "corr_grad_prep.hip", "warp.hip", "upsample.hip"]
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall",
            "-Wno-unused-function", "-fno-fast-math",
            "-fhip-fp32-correctly-rounded-divide-sqrt"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libcerberus_hip.so)")


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "cerberus_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force, verbose, extra):
    obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    newest = max(os.path.getmtime(path), _deps())
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj, False
    cmd = [_hipcc()] + CXXFLAGS + extra + ["-c", path, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return obj, True


def build(force: bool = False, verbose: bool = False, extra_flags=()):
    """Compile every HIP source for gfx950 and link the shared library."""
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    extra = list(extra_flags) + os.environ.get("CERB_EXTRA_HIPCC_FLAGS", "").split()
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as pool:
        results = list(pool.map(lambda s: _compile(s, force, verbose, extra), SOURCES))
    objs = [o for o, _ in results]
    if force or any(changed for _, changed in results) or not os.path.exists(LIB):
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    try:    # the inference host is an optional extra: its failure must not fail the build of the product library
        build_runtime(force or any(changed for _, changed in results), verbose)
    except (subprocess.CalledProcessError, OSError) as exc:
        print("warning: cerberus_run (runtime/) was not built: %s" % exc, file=sys.stderr)
    return LIB


RUNTIME_DIR = os.path.join(HERE, "runtime")
RUNTIME_BIN = os.path.join(LIBDIR, "cerberus_run")


def build_runtime(force: bool = False, verbose: bool = False):
    """The torch-free C++ inference host (runtime/cerberus_run.cpp over runtime/cerberus_runtime.hpp): plain
    host code against the HIP runtime and the C ABI of libcerberus_hip.so, found at run time through
    $ORIGIN (the binary sits beside the library)."""
    srcs = [os.path.join(RUNTIME_DIR, f) for f in ("cerberus_run.cpp", "cerberus_runtime.hpp")]
    newest = max([os.path.getmtime(p) for p in srcs] + [_deps()])
    if not force and os.path.exists(RUNTIME_BIN) and os.path.getmtime(RUNTIME_BIN) >= newest:
        return RUNTIME_BIN
    cmd = [_hipcc(), "-O2", "-std=c++17", "-Wall", srcs[0], "-o", RUNTIME_BIN, "-L", LIBDIR, "-lcerberus_hip",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return RUNTIME_BIN


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
