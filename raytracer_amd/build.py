"""Builds the native libraries in-tree.

    python -m raytracer_amd.build            # everything
    python -m raytracer_amd.build product    # raytracer_amd/librt_hip.so only

product   raytracer_amd/librt_hip.so     HIP kernels for gfx950 + host scene builder + the C ABI (include/rt.h)
oracle    oracle/liboracle.so            CPU restatement of the reference (test infrastructure)
hostsim   tests/hostsim/libhostsim.so    per-lane device code compiled for the CPU (test tool)

    python -m raytracer_amd.build --sanitize   # build/san/: the three libraries with the host code under ASan + UBSan

The sanitized set (librt_hip_san.so: host side of the product, i.e. scene builder, compiler, named scenes and the C ABI;
libhostsim_san.so: the per-lane device code on the CPU, with every reference checked against its section's length,
-DRT_BOUNDS_CHECK; liboracle_san.so) is what tools/sanitize.sh runs the CPU parity tests on. GPU code is not instrumented.

All three are compiled with -ffp-contract=off: the reference (Rust) never fuses a*b+c, and bit-level
agreement between the CPU checker and the GPU depends on neither side doing so.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracer_amd", "csrc")
PRODUCT_SO = os.path.join(ROOT, "raytracer_amd", "librt_hip.so")
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
HOSTSIM_SO = os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _csrc_files():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(ROOT, "include", "rt.h")]


def build_product(force=False):
    srcs = [os.path.join(CSRC, f) for f in ("rt_kernels.hip", "rt_tree_build.hip", "rt_builder.cpp", "rt_scenes.cpp")]
    if not force and _newer(PRODUCT_SO, _csrc_files()):
        return PRODUCT_SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    _run([hipcc, "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-fPIC", "-shared",
          "-Wno-unused-value", "-Wno-unused-result", *srcs, "-o", PRODUCT_SO])
    return PRODUCT_SO


def build_oracle(force=False):
    src = os.path.join(ROOT, "oracle", "oracle.cpp")
    if not force and _newer(ORACLE_SO, [src, os.path.join(ROOT, "include", "rt.h")]):
        return ORACLE_SO
    # (no -march=native: the library is built in one machine and run on another; bench.py's CPU baseline compiles its own
    # -march=native copy on the host it times)
    _run(["g++", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", src, "-o", ORACLE_SO])
    return ORACLE_SO


def build_hostsim(force=False):
    src = os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp")
    if not force and _newer(HOSTSIM_SO, [src, PRODUCT_SO] + _csrc_files()):
        return HOSTSIM_SO
    _run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", src, "-o", HOSTSIM_SO,
          "-L" + os.path.dirname(PRODUCT_SO), "-lrt_hip", "-Wl,-rpath,$ORIGIN/../../raytracer_amd"])
    return HOSTSIM_SO


SAN_DIR = os.path.join(ROOT, "build", "san")
SAN_FLAGS = ["-fsanitize=address", "-fsanitize=undefined", "-fno-omit-frame-pointer"]


def clang_asan_runtime():
    """The shared ASan runtime of the clang that hipcc drives: a sanitized shared library loaded into an uninstrumented
    python needs it preloaded."""
    out = subprocess.check_output(["/opt/rocm/lib/llvm/bin/clang++", "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    return out


def build_sanitized():
    """Host code of all three libraries under AddressSanitizer + UndefinedBehaviorSanitizer (one compiler for all of them,
    the clang behind hipcc, so that one runtime serves the process)."""
    os.makedirs(SAN_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    srcs = [os.path.join(CSRC, f) for f in ("rt_kernels.hip", "rt_tree_build.hip", "rt_builder.cpp", "rt_scenes.cpp")]
    host_only = []
    for f in SAN_FLAGS:  # (each flag directly behind its -Xarch_host: the device side is compiled without sanitizers)
        host_only += ["-Xarch_host", f]
    _run([hipcc, "-O1", "-g", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wno-unused-value",
          "-Wno-unused-result", *host_only, "-shared-libsan", *srcs, "-o", os.path.join(SAN_DIR, "librt_hip_san.so")])
    common = [clang, "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", *SAN_FLAGS, "-shared-libsan"]
    _run([*common, "-DRT_BOUNDS_CHECK", os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp"), "-o", os.path.join(SAN_DIR, "libhostsim_san.so"),
          "-L" + SAN_DIR, "-lrt_hip_san", "-Wl,-rpath," + SAN_DIR])
    _run([*common, "-pthread", os.path.join(ROOT, "oracle", "oracle.cpp"), "-o", os.path.join(SAN_DIR, "liboracle_san.so")])
    return SAN_DIR


def build_all(force=False):
    build_product(force)
    build_oracle(force)
    build_hostsim(force)


if __name__ == "__main__":
    what = sys.argv[1:] or ["all"]
    if "--sanitize" in what:
        build_sanitized()
        sys.exit(0)
    force = "--force" in what
    what = [w for w in what if not w.startswith("--")] or ["all"]
    for w in what:
        {"all": build_all, "product": build_product, "oracle": build_oracle, "hostsim": build_hostsim}[w](force)
