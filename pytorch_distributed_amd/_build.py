"""In-tree builder for ``pytorch_distributed_amd/_C.so``.

The native code has two halves that are compiled differently:

* ``csrc/kernels/*.hip`` — the hand-written gfx950 kernels. They include no
  PyTorch header, so each compiles in seconds with
  ``hipcc --offload-arch=gfx950``.
* ``csrc/**/*.cpp`` — the PyTorch-facing bindings and the C++ runtime
  (reducer, sharding units, loader, tracer). Host-only code, compiled with
  ``g++`` against the installed libtorch headers.

Objects are cached under ``build/obj`` keyed by a hash of the source, the
headers next to it and the flags, so a rebuild only recompiles what changed.
Nothing here needs a GPU: hipcc cross-compiles gfx950 code objects.

Run ``python -m pytorch_distributed_amd._build`` (``--force`` rebuilds all).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
REPO = PKG_DIR.parent
OBJ_DIR = REPO / "build" / "obj"
TARGET = PKG_DIR / "_C.so"
ARCH = "gfx950"
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))


def _torch_paths():
    import torch
    from torch.utils import cpp_extension as ce

    inc = [p for p in ce.include_paths() if os.path.isdir(p)] + [str(ROCM / "include")]
    torch_lib = str(Path(torch.__file__).resolve().parent / "lib")
    return inc, torch_lib, bool(torch._C._GLIBCXX_USE_CXX11_ABI)


def _hash(paths, flags) -> str:
    h = hashlib.sha256()
    for p in sorted(set(map(str, paths))):
        h.update(p.encode())
        h.update(Path(p).read_bytes())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def _compile(src: Path, cmd_prefix, flags, deps, force, verbose):
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    rel = src.relative_to(CSRC)
    tag = "asan_" if any("-fsanitize" in f for f in flags) else ""  # flavours keep separate objects
    obj = OBJ_DIR / (tag + "_".join(rel.parts) + ".o")
    stamp = obj.with_suffix(".hash")
    digest = _hash([src, *deps], [*cmd_prefix, *flags])
    if not force and obj.exists() and stamp.exists() and stamp.read_text() == digest:
        return obj, False
    cmd = [*cmd_prefix, *flags, "-c", str(src), "-o", str(obj)]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"compilation of {src} failed:\n{proc.stdout}\n{proc.stderr}")
    stamp.write_text(digest)
    return obj, True


def build(force: bool = False, verbose: bool = False, sanitize: bool = False) -> Path:
    """Compile every native source for gfx950 and link ``_C.so``. Returns its path.

    ``sanitize=True`` is the debugging flavour: the host-side C++ (bindings + runtime) is built
    with AddressSanitizer and UndefinedBehaviorSanitizer into ``build/asan/_C.so`` (the in-tree
    library is left alone; GPU code is never sanitized). Run it on the CPU with
    ``LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0
    PYTHONPATH=build/asan-overlay python -m pytest tests -m "not gpu"`` — see
    ``tools/run_asan_tests.sh``."""
    hipcc = shutil.which("hipcc") or str(ROCM / "bin" / "hipcc")
    cxx = shutil.which("g++") or "g++"
    inc, torch_lib, cxx11 = _torch_paths()
    py_inc = sysconfig.get_paths()["include"]

    kernel_srcs = sorted((CSRC / "kernels").glob("*.hip"))
    kernel_hdrs = sorted((CSRC / "kernels").glob("*.h"))
    host_srcs = sorted(p for p in CSRC.rglob("*.cpp"))
    host_hdrs = sorted(p for p in CSRC.rglob("*.h")) + sorted(CSRC.rglob("*.hpp"))

    hip_flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffast-math",
                 "-fno-finite-math-only", "-DNDEBUG", f"-I{CSRC}"]
    host_flags = ["-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
                  "-Wno-unknown-pragmas", "-Wno-sign-compare",
                  "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DUSE_C10D_GLOO", "-DUSE_C10D_NCCL",
                  "-DUSE_DISTRIBUTED", "-DUSE_RPC",
                  f"-D_GLIBCXX_USE_CXX11_ABI={int(cxx11)}", "-DTORCH_EXTENSION_NAME=_C",
                  "-DTORCH_API_INCLUDE_EXTENSION_H", f"-I{CSRC}", f"-I{py_inc}",
                  *[f"-isystem{p}" for p in inc]]

    target = TARGET
    if sanitize:
        host_flags = [f for f in host_flags if f != "-O2"] + ["-O1", "-g", "-fno-omit-frame-pointer",
                                                              "-fsanitize=address,undefined"]
        target = REPO / "build" / "asan" / "_C.so"
        target.parent.mkdir(parents=True, exist_ok=True)

    jobs = int(os.environ.get("MAX_JOBS", "0")) or min(8, os.cpu_count() or 1)
    tasks = [(s, [hipcc], hip_flags, kernel_hdrs) for s in kernel_srcs]
    tasks += [(s, [cxx], host_flags, host_hdrs) for s in host_srcs]
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        results = list(pool.map(lambda t: _compile(t[0], t[1], t[2], t[3], force, verbose), tasks))
    objs = [str(o) for o, _ in results]
    rebuilt = any(r for _, r in results)

    if sanitize:
        # g++ drives this link so that its own sanitizer runtimes are the ones pulled in; the
        # kernel objects only need the HIP runtime that PyTorch ships.
        link = [cxx, "-shared", "-fPIC", "-fsanitize=address,undefined", "-o", str(target), *objs,
                f"-L{torch_lib}", f"-Wl,-rpath,{torch_lib}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip",
                "-ltorch", "-ltorch_python", str(Path(torch_lib) / "libhipblaslt.so"),
                str(Path(torch_lib) / "libamdhip64.so"), "-lpthread", "-ldl"]
        if verbose:
            print("[build]", " ".join(link), flush=True)
        proc = subprocess.run(link, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"link failed:\n{proc.stdout}\n{proc.stderr}")
        return target
    if rebuilt or force or not TARGET.exists():
        link = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(TARGET), *objs,
                f"-L{torch_lib}", f"-Wl,-rpath,{torch_lib}", "-lc10", "-lc10_hip", "-ltorch_cpu",
                "-ltorch_hip", "-ltorch", "-ltorch_python",
                # the hipBLASLt and RCCL that PyTorch itself loaded (linking /opt/rocm's copies would
                # pull a second HIP runtime into the process)
                str(Path(torch_lib) / "libhipblaslt.so"), str(Path(torch_lib) / "librccl.so"), "-lpthread"]
        if verbose:
            print("[build]", " ".join(link), flush=True)
        proc = subprocess.run(link, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"link failed:\n{proc.stdout}\n{proc.stderr}")
    return TARGET


if __name__ == "__main__":
    out = build(force="--force" in sys.argv, verbose=True, sanitize="--sanitize" in sys.argv)
    print(f"built {out}")
