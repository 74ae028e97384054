"""Build + load libstain2stain_hip.so (the C ABI declared in include/stain2stain_hip.h).

The library is plain HIP compiled by hipcc for gfx950 and bound through ctypes; there is no torch
in its ABI.  There is deliberately NO fallback: if the library cannot be loaded every op raises.
"""
from __future__ import annotations

import ctypes
import os
import re
import subprocess
from typing import Dict, List, Optional

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libstain2stain_hip.so")
# scripts/ only: the same sources with -DS2S_ABLATE (superseded forward loops, result-changing timing ablations)
ABLATE_LIB_PATH = os.path.join(HERE, "libstain2stain_hip_ablate.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "stain2stain_hip.h")
SOURCES = ["conv3x3_mfma.hip", "conv3x3_wgrad_mfma.hip", "conv_small.hip", "conv_edge.hip", "norm_act.hip", "resample.hip",
           "flow.hip", "optim.hip", "input_pipeline.hip", "seg_loss.hip", "loss_variants.hip", "instnorm.hip", "pix2pix.hip",
           "runtime.hip"]

_lib: Optional[ctypes.CDLL] = None

ERRORS = {-1: "invalid shape / unsupported size", -2: "misaligned pointer", -3: "unsupported dtype",
          -4: "kernel launch failed", -5: "null pointer"}


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build(lib_path: str = None) -> bool:
    lib_path = lib_path or LIB_PATH
    if not os.path.exists(lib_path):
        return True
    t = os.path.getmtime(lib_path)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "common.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, jobs: int = 4, ablate: bool = False) -> str:
    """Compile every HIP source for gfx950 and link the shared library in-tree.  ``ablate``: the -DS2S_ABLATE variant
    (libstain2stain_hip_ablate.so: timing ablations whose results are wrong by construction; ``scripts/ablate_lib.py``
    builds and loads it explicitly -- the package itself never does, whatever the environment says)."""
    if ablate:
        if not force and not needs_build(ABLATE_LIB_PATH):
            return ABLATE_LIB_PATH
        return _build_to(ABLATE_LIB_PATH, os.path.join(HERE, "build_ablate"), ["-DS2S_ABLATE"], force, verbose, jobs)
    if not force and not needs_build():
        return LIB_PATH
    return _build_to(LIB_PATH, os.path.join(HERE, "build"), [], force, verbose, jobs)


def _build_to(lib_path: str, objdir: str, extra: List[str], force: bool, verbose: bool, jobs: int) -> str:
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", *extra]
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        srcp = os.path.join(CSRC, src)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(srcp)
                and os.path.getmtime(obj) > os.path.getmtime(os.path.join(CSRC, "common.h"))):
            continue
        cmd = [hipcc(), *flags, "-c", srcp, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        if len(procs) >= jobs:
            _drain(procs)
    _drain(procs)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, *objs]
    if verbose:
        print(" ".join(cmd))
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if out.returncode != 0:
        raise RuntimeError("link failed:\n" + out.stdout.decode())
    return lib_path


def _drain(procs: List) -> None:
    while procs:
        src, p = procs.pop(0)
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")


def declared_prototypes() -> Dict[str, tuple]:
    """{name: (restype, [argtypes])} parsed from include/stain2stain_hip.h (the single source of truth)."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos: Dict[str, tuple] = {}
    for ret, name, args in re.findall(r"\b(int|long)\s+(s2s_\w+)\s*\(([^)]*)\)\s*;", text):
        argtypes = []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            if "*" in a:
                argtypes.append(ctypes.c_void_p)
            elif a.startswith("int"):
                argtypes.append(ctypes.c_int)
            elif a.startswith("long"):
                argtypes.append(ctypes.c_long)
            elif a.startswith("float"):
                argtypes.append(ctypes.c_float)
            else:
                raise RuntimeError(f"unparsed parameter {a!r} in {name}")
        protos[name] = (ctypes.c_long if ret == "long" else ctypes.c_int, argtypes)
    return protos


def declared_symbols() -> List[str]:
    """Function names declared in include/stain2stain_hip.h."""
    return sorted(declared_prototypes())


def lib() -> ctypes.CDLL:
    """Load (never build implicitly on a GPU box unless the .so is missing) and return the library."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            import fcntl                   # one rank of a multi-process launch builds, the others wait for it
            with open(os.path.join(HERE, ".build.lock"), "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                try:
                    if not os.path.exists(LIB_PATH):
                        build()
                finally:
                    fcntl.flock(lock, fcntl.LOCK_UN)
        if _lib is None:
            try:
                _lib = ctypes.CDLL(LIB_PATH)
            except OSError as e:  # loud failure: there is no CPU / eager path behind these ops
                raise RuntimeError(f"stain2stain_amd: cannot load {LIB_PATH}: {e}") from e
        for name, (restype, argtypes) in declared_prototypes().items():
            fn = getattr(_lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
    return _lib


def load_library(path: str) -> ctypes.CDLL:
    """Bind ``path`` instead of the product library.  For scripts/ablate_lib.py only (the -DS2S_ABLATE build): it must be
    called before anything else has loaded the library, and it says loudly what it did."""
    global _lib
    if _lib is not None:
        raise RuntimeError("stain2stain_amd: a library is already loaded; load_library() must come first")
    import sys
    print(f"stain2stain_amd: LOADING {path} INSTEAD OF THE PRODUCT LIBRARY -- results of this process are not to be trusted",
          file=sys.stderr)
    _lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in declared_prototypes().items():
        fn = getattr(_lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"stain2stain_amd: {what} failed: {ERRORS.get(rc, rc)} (status {rc})")
