"""In-tree build of the native pieces (no cmake needed):

  fcc_qp_amd/libfccqp_hip.so        HIP kernels for gfx950 + the C ABI of include/fccqp.h  (hipcc)
  fcc_qp_amd/fcc_qp_solver.*.so     pybind11 module with the reference's Python surface     (g++)

The reference builds the same two layers with CMake (CMakeLists.txt:32-40: static lib fcc_qplib +
pybind11 module fcc_qp_solver); here the library is the HIP path instead of Eigen.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
import shutil
import subprocess
import sys
import sysconfig
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfccqp_hip.so")
ARCH = "gfx950"


def _ext_suffix() -> str:
    return sysconfig.get_config_var("EXT_SUFFIX") or ".so"


PYMOD = os.path.join(HERE, "fcc_qp_solver" + _ext_suffix())


def _newer(target: str, deps: List[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd: List[str]) -> None:
    print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")


def build_lib(force: bool = False) -> str:
    srcs = [os.path.join(CSRC, f) for f in ("fccqp_capi.cpp", "fccqp_generic.hip", "fccqp_fused.hip",
                                            "fccqp_fused_f32.hip", "fccqp_fused_split.hip", "fccqp_coop.hip",
                                            "fccqp_check.hip")]
    def c(*names):
        return [os.path.join(CSRC, x) for x in names]
    # what each translation unit includes (the unrolled tile kernels take minutes: only what changed is recompiled)
    fused_deps = c("fccqp_device.h", "fccqp_fused_impl.h", "fused_shapes.def")
    own_deps = {
        "fccqp_capi.cpp": c("fccqp_device.h") + [os.path.join(ROOT, "include", "fccqp.h")],
        "fccqp_generic.hip": c("fccqp_device.h", "fccqp_cod_impl.h"),
        "fccqp_fused.hip": fused_deps,
        "fccqp_fused_f32.hip": fused_deps + c("fccqp_fused.hip"),
        "fccqp_fused_split.hip": fused_deps,
        "fccqp_coop.hip": c("fccqp_device.h", "fccqp_fused_impl.h", "fccqp_coop_impl.h", "coop_shapes.def"),
        "fccqp_check.hip": c("fccqp_device.h"),
    }
    this_file = os.path.abspath(__file__)       # (the compiler flags live here)
    deps = srcs + [d for v in own_deps.values() for d in v] + [this_file]
    if force or _newer(LIB, deps):
        objs = [os.path.splitext(src)[0] + ".o" for src in srcs]

        def compile_one(job):
            src, obj = job
            if force or _newer(obj, [src, this_file] + own_deps[os.path.basename(src)]):
                # the fully unrolled MFMA tile code (up to 13 row blocks) is larger than LLVM's default
                # budget for `#pragma unroll`; left partly rolled it indexes register arrays dynamically
                # fused kernel: loop headers on 256-byte boundaries -- its ADMM loop ran 5 % slower or faster
                # with nothing but its placement in the code (the generic kernel, full of short loops, loses 2 %)
                align = ["-falign-loops=256"] if ("fused" in os.path.basename(src) or "coop" in os.path.basename(src)) else []
                # generic kernel: NO real function calls (every device function inlined).  With real calls the 8-slot
                # instantiation (256 < N <= 512: 256 VGPRs + 256 AGPRs, callee-saved registers parked in AGPRs, scratch)
                # returned wrong, run-to-run different results on MI355X -- z off by 1e-10 .. 1e+3, or NaN, depending
                # on the box -- while the same source with this switch is exact and bit-reproducible (same box,
                # alternating runs: tools/n512_probe.py).  Costs ~1.5 min of compile time.
                if os.path.basename(src) == "fccqp_generic.hip":
                    align = align + ["-mllvm", "-amdgpu-function-calls=false"]
                _run([hipcc(), "-x", "hip", f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
                      "-fno-gpu-rdc", "-ffp-contract=off"] + align + ["-mllvm", "-pragma-unroll-threshold=400000",
                      "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-c", src, "-o", obj])

        # the translation units are independent
        with ThreadPoolExecutor(max_workers=min(7, len(srcs))) as pool:
            list(pool.map(compile_one, zip(srcs, objs)))
        _run([hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs)
    return LIB


def build_pymod(force: bool = False) -> str:
    import pybind11

    srcs = [os.path.join(CSRC, f) for f in ("pybind_module.cpp", "fcc_qp.cpp")]
    deps = srcs + [os.path.join(CSRC, "fcc_qp.hpp"), os.path.join(ROOT, "include", "fccqp.h"), LIB]
    if force or _newer(PYMOD, deps):
        inc = [f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}",
               f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"]
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall"] + inc + srcs +
             ["-o", PYMOD, f"-L{HERE}", "-lfccqp_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"])
    return PYMOD


def build_all(force: bool = False) -> None:
    build_lib(force)
    build_pymod(force)


def add_shape(kind: str, dims) -> bool:
    """Appends a shape to csrc/fused_shapes.def (kind "fused": num_vars <= 63, one wavefront per QP) or
    csrc/coop_shapes.def (kind "coop": num_vars <= ~118, `waves` wavefronts per QP) unless it is listed already;
    the next build compiles the specialised kernel for it (static_asserts in the kernels' Shape structs say so if the
    shape does not qualify).  Every other shape runs on the generic kernel, several times slower."""
    dims = tuple(int(d) for d in dims)
    if kind == "fused":
        assert len(dims) == 4, "fused shape: num_vars num_equality_constraints nc lambda_c_start"
        path, line = os.path.join(CSRC, "fused_shapes.def"), "FCCQP_FUSED_SHAPE(%d, %d, %d, %d)" % dims
    else:
        assert len(dims) == 5, "coop shape: num_vars num_equality_constraints nc lambda_c_start waves"
        path, line = os.path.join(CSRC, "coop_shapes.def"), "FCCQP_COOP_SHAPE(%d, %d, %d, %d, %d)" % dims
    text = open(path).read()
    listed = {l.split("//")[0].replace(" ", "") for l in text.splitlines()}
    if line.replace(" ", "") in listed:
        return False
    with open(path, "a") as f:
        f.write(("" if text.endswith("\n") else "\n") + line + "   // added by fcc_qp_amd/build.py --add-%s-shape\n" % kind)
    return True


if __name__ == "__main__":
    # python fcc_qp_amd/build.py [--force] [--add-fused-shape n m nc s] [--add-coop-shape n m nc s waves]
    argv = sys.argv[1:]
    for flag, kind, k in (("--add-fused-shape", "fused", 4), ("--add-coop-shape", "coop", 5)):
        while flag in argv:
            i = argv.index(flag)
            added = add_shape(kind, argv[i + 1:i + 1 + k])
            print("[build] %s shape %s: %s" % (kind, " ".join(argv[i + 1:i + 1 + k]), "added" if added else "already listed"))
            del argv[i:i + 1 + k]
    build_all(force="--force" in argv)
