"""Build recipe for the oracle (test infrastructure only).

* ``build_oracle()``  -> oracle/libr3_oracle.so from oracle/r3_oracle.cpp (g++, no torch).
* ``build_ref()``     -> oracle/_ref/libref_*.so: the REFERENCE's own CPU sources, compiled
  from where they lie under /root/reference together with the thin C-ABI drivers in
  oracle/ref_harness/.  Only runs where /root/reference exists (this container); the GPU
  box uses the prebuilt files that travel with the snapshot.  No reference source is
  copied into the repository and no stand-in headers/libraries are written: the sources
  need only the torch headers and libtorch that ship in this image.

Flags for the reference TUs mirror what its setup.py gets from distutils (-O2, no
-march, so no FMA contraction on x86-64).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("R3DET_REFERENCE", "/root/reference")
REF_OPS = os.path.join(REF_ROOT, "r3det", "ops")
OUT_REF = os.path.join(HERE, "_ref")
ORACLE_SO = os.path.join(HERE, "libr3_oracle.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout)
        raise RuntimeError("oracle build failed: " + cmd[-1])


def build_oracle(force=False):
    src = os.path.join(HERE, "r3_oracle.cpp")
    if not force and _newer(ORACLE_SO, [src]):
        return ORACLE_SO
    _run(["g++", "-O2", "-ffp-contract=off", "-fopenmp", "-std=c++17", "-shared", "-fPIC",
          "-o", ORACLE_SO, src])
    return ORACLE_SO


def ref_available():
    return os.path.isdir(REF_OPS)


def _torch_flags():
    import torch
    from torch.utils import cpp_extension as ce
    import sysconfig
    inc = [f"-I{p}" for p in ce.include_paths()]
    inc.append("-I" + sysconfig.get_paths()["include"])  # torch/extension.h pulls Python.h
    libdir = ce.library_paths()[0]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cflags = ["-O2", "-std=c++17", "-fPIC", "-w", f"-D_GLIBCXX_USE_CXX11_ABI={abi}"] + inc
    ldflags = [f"-L{libdir}", f"-Wl,-rpath,{libdir}", "-ltorch", "-ltorch_cpu", "-lc10"]
    return cflags, ldflags


def build_ref(force=False):
    """Compile the reference CPU sources into oracle/_ref/. Returns dict name->path."""
    if not ref_available():
        return {}
    os.makedirs(OUT_REF, exist_ok=True)
    cflags, ldflags = _torch_flags()
    H = os.path.join(HERE, "ref_harness")
    jobs = {
        "libref_v1.so": dict(
            srcs=[os.path.join(H, "harness_v1.cpp")],
            defs=['-DREF_RNMS_CPU="%s"' % os.path.join(REF_OPS, "rnms/src/rcpu/rnms_cpu.cpp")],
            deps=[os.path.join(REF_OPS, "rnms/src/rcpu/rnms_cpu.cpp")]),
        "libref_iou_v3.so": dict(
            srcs=[os.path.join(H, "harness_v3.cpp"),
                  os.path.join(REF_OPS, "box_iou_rotated/src/box_iou_rotated_cpu.cpp")],
            defs=["-DHARNESS_IOU"], deps=[]),
        "libref_nms_v3.so": dict(
            srcs=[os.path.join(H, "harness_v3.cpp"),
                  os.path.join(REF_OPS, "nms_rotated/src/nms_rotated_cpu.cpp")],
            defs=["-DHARNESS_NMS"], deps=[]),
        "libref_v2.so": dict(
            srcs=[os.path.join(H, "harness_v2.cpp")],
            defs=['-DREF_ML_UTILS_H="%s"' % os.path.join(
                REF_OPS, "ml_nms_rotated/src/box_iou_rotated_utils.h")],
            deps=[os.path.join(REF_OPS, "ml_nms_rotated/src/box_iou_rotated_utils.h")]),
        "libref_convex.so": dict(
            srcs=[os.path.join(H, "harness_rank4.cpp"), os.path.join(REF_OPS, "convex/src/convex_cpu.cpp")],
            defs=["-DHARNESS_CONVEX"], deps=[]),
        "libref_polygon.so": dict(
            srcs=[os.path.join(H, "harness_rank4.cpp")],
            defs=["-DHARNESS_POLYGON", "-DTORCH_EXTENSION_NAME=ref_polygon_geo",
                  '-DREF_POLYGON_CPP="%s"' % os.path.join(REF_OPS, "polygon_geo/src/polygon_geo_cpu.cpp")],
            deps=[os.path.join(REF_OPS, "polygon_geo/src/polygon_geo_cpu.cpp")],
            libs=["-ltorch_python"]),  # the file's PYBIND11_MODULE block needs the Tensor casters
    }
    out = {}
    for name, j in jobs.items():
        target = os.path.join(OUT_REF, name)
        out[name] = target
        if not force and _newer(target, j["srcs"] + j["deps"]):
            continue
        objs = []
        for s in j["srcs"]:
            o = os.path.join(OUT_REF, name[:-3] + "_" + os.path.basename(s) + ".o")
            _run(["g++"] + cflags + j["defs"] + ["-c", s, "-o", o])
            objs.append(o)
        # -Bsymbolic: each .so binds its own copies of the reference's inline templates
        _run(["g++", "-shared", "-Wl,-Bsymbolic", "-o", target] + objs + ldflags + j.get("libs", []))
        for o in objs:
            os.remove(o)
    return out


if __name__ == "__main__":
    print(build_oracle(force="--force" in sys.argv))
    for k, v in build_ref(force="--force" in sys.argv).items():
        print(k, "->", v)
