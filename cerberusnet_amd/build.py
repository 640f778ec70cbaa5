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
# the product: dispatched code only
SOURCES = ["api.hip", "corr_generic.hip", "corr_d4.hip", "corr_d4_bwd.hip", "corr_strip.hip", "corr_coarse.hip", "corr_mfma.hip",
           "corr_grad_prep.hip", "warp.hip", "warp16.hip", "warp_corr.hip", "upsample.hip"]
# lib/libcerberus_hip_experiments.so (-DCERB_EXPERIMENTS): the product's sources + the measured-and-rejected kernels that no
# dispatch rule selects (forward variants 1, 2, 8, 17; backward 2, 6, 7, 9, 10; the two-item strip backward) -- kept for the
# record and tested through CERBERUS_HIP_LIB (tests/test_experiments_gpu.py), never loaded by default
EXPERIMENT_SOURCES = SOURCES + ["corr_fwd_pipe.hip"]
EXPERIMENTS_LIB = os.path.join(LIBDIR, "libcerberus_hip_experiments.so")
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


def _compile(src, force, verbose, extra, objdir=OBJDIR):
    obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    newest = max(os.path.getmtime(path), _deps())
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj, False
    cmd = [_hipcc()] + CXXFLAGS + extra + ["-c", path, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return obj, True


def _build_lib(lib, objdir, sources, extra, force, verbose):
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as pool:
        results = list(pool.map(lambda s: _compile(s, force, verbose, extra, objdir), sources))
    objs = [o for o, _ in results]
    changed = force or any(c for _, c in results) or not os.path.exists(lib)
    if changed:
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return changed


def build_variant(tag: str, flags, force: bool = False, verbose: bool = False, sources=None):
    """A diagnostic build of the same library beside the product: ``lib/libcerberus_hip_<tag>.so`` from objects of
    its own (``csrc/_obj_<tag>``), e.g. ``build_variant("stamp", ["-DCERB_STAMP"])``.  Loaded through
    ``CERBERUS_HIP_LIB``; never the default."""
    lib = os.path.join(LIBDIR, "libcerberus_hip_%s.so" % tag)
    _build_lib(lib, os.path.join(CSRC, "_obj_" + tag), sources or SOURCES, list(flags), force, verbose)
    return lib


def build_experiments(force: bool = False, verbose: bool = False):
    return build_variant("experiments", ["-DCERB_EXPERIMENTS"], force, verbose, sources=EXPERIMENT_SOURCES)


def build(force: bool = False, verbose: bool = False, extra_flags=()):
    """Compile every HIP source for gfx950 and link the shared library."""
    extra = list(extra_flags) + os.environ.get("CERB_EXTRA_HIPCC_FLAGS", "").split()
    changed = _build_lib(LIB, OBJDIR, SOURCES, extra, force, verbose)
    results = [(None, changed)]
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
    if "--experiments" in sys.argv:
        print(build_experiments(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
    elif "--variant" in sys.argv:     # python -m cerberusnet_amd.build --variant stamp -DCERB_STAMP
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith("-D")],
                            force="--force" in sys.argv, verbose="--verbose" in sys.argv))
    else:
        print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
