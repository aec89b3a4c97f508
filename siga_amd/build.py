"""Build the in-tree native library: siga_amd/lib/libsigax.so (HIP kernels + C-ABI, gfx950)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libsigax.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["sigax_kernels.hip", "sigax_api.cpp", "sigax_index_build.hip", "sigax_comm.cpp", "sigax_keys.hip"]
HEADERS = ["sigax_kernels.h", "fm_layout.h", os.path.join(ROOT, "include", "sigax.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_libsigax(force=False, verbose=False, out=None, defines=()):
    """out/defines: build a variant (tests use -DSIGAX_SUPER_SHIFT=12 to exercise the multi-superblock path)."""
    lib = out or LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    if not force and not _stale(lib, deps):
        return lib
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "-Wno-unused-value", "-Wno-int-to-void-pointer-cast",
           "-I" + os.path.join(ROOT, "include"), "-o", lib] + ["-D" + d for d in defines] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


HOST = os.path.join(HERE, "host")
HOSTLIB = os.path.join(HERE, "lib", "libsiga_host.so")
CLI = os.path.join(HERE, "lib", "siga")
CXX = os.environ.get("CXX", "g++")


def build_host(force=False, verbose=False):
    """libsiga_host.so (host C++ mirror of the reference classes over the C-ABI) and the `siga` CLI."""
    build_libsigax(force=force, verbose=verbose)
    deps = [os.path.join(HOST, f) for f in ("siga_host.cpp", "siga_host.hpp", "sais.hpp", "line_deflate.hpp", "siga_main.cpp")] + [LIB]
    libdir = os.path.dirname(LIB)
    common = [CXX, "-O2", "-std=c++17", "-fPIC", "-Wall", "-Wno-sign-compare", "-pthread"]
    link = ["-L" + libdir, "-lsigax", "-lz", "-ldl", "-Wl,-rpath,$ORIGIN"]
    if force or _stale(HOSTLIB, deps):
        cmd = common + ["-shared", "-o", HOSTLIB, os.path.join(HOST, "siga_host.cpp")] + link
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    if force or _stale(CLI, deps + [HOSTLIB]):
        cmd = common + ["-o", CLI, os.path.join(HOST, "siga_main.cpp"), "-lsiga_host"] + link
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOSTLIB, CLI


def build_all(force=False, verbose=False):
    build_libsigax(force, verbose)
    return build_host(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
