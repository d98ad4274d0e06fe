"""Builds libvision_amd.so (gfx950) in-tree with hipcc.  `python vision_amd/build_ext.py` (run as a script: importing the package needs the library).

    python vision_amd/build_ext.py                      product library  -> vision_amd/lib/libvision_amd.so
    python vision_amd/build_ext.py --probe              diagnostics      -> vision_amd/lib/libvamd_probe.so
    python vision_amd/build_ext.py --variant NAME -DX=1 experiment build -> vision_amd/lib/libvision_amd_NAME.so

The product library is never compiled with extra preprocessor definitions: kernel experiments (among
them the timing-only `VAMD_ABLATE` variants, whose results are wrong by construction) go into a
separately named library that `VAMD_LIB=<path>` selects explicitly.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(OUT_DIR, "libvision_amd.so")
PROBE_LIB = os.path.join(OUT_DIR, "libvamd_probe.so")
SOURCES = ["elementwise.hip", "norm.hip", "conv_direct.hip", "conv_api.hip", "conv_mfma.hip", "conv_wgrad_halo.hip", "conv_thin.hip",
           "conv_thin_mfma.hip", "attention.hip", "vq.hip", "image.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable"]


def _stale(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile_and_link(sources, lib, obj_dir, defines=(), force=False, verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "vision_amd.h"))
    objs, procs = [], []
    for src in sources:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            cmd = [HIPCC] + FLAGS + list(defines) + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


def build(force=False, verbose=True):
    """The product library: fixed flags, no -D of any kind."""
    if any("VAMD_ABLATE" in str(v) for v in (os.environ.get("HIPCC", ""), os.environ.get("HIPCC_COMPILE_FLAGS_APPEND", ""),
                                             os.environ.get("CXXFLAGS", ""))):
        raise RuntimeError("VAMD_ABLATE must not reach the product library; use --variant NAME -DVAMD_ABLATE=n")
    return _compile_and_link(SOURCES, LIB, os.path.join(HERE, "build"), (), force, verbose)


def build_variant(name, defines, force=False, verbose=True):
    """An experiment build under its own name and object directory; -DVAMD_ABLATE=n additionally gets the
    acknowledgement macro without which conv_mfma.hip refuses to compile an ablation."""
    defines = list(defines)
    if any(d.startswith("-DVAMD_ABLATE") for d in defines):
        defines.append("-DVAMD_TIMING_ONLY_BUILD")
    lib = os.path.join(OUT_DIR, f"libvision_amd_{name}.so")
    return _compile_and_link(SOURCES, lib, os.path.join(HERE, "build", "variant_" + name), defines, force, verbose)


def build_probe(force=False, verbose=True):
    return _compile_and_link(["probe.hip"], PROBE_LIB, os.path.join(HERE, "build", "probe"), (), force, verbose)


if __name__ == "__main__":
    argv = sys.argv[1:]
    force = "--force" in argv
    if "--probe" in argv:
        print(build_probe(force))
    elif "--variant" in argv:
        name = argv[argv.index("--variant") + 1]
        print(build_variant(name, [a for a in argv if a.startswith("-D")], force))
    else:
        build(force)
        print(LIB)
