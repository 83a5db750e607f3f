"""Builds libqsv.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

The shared object is written next to this file so that it travels with the repository snapshot to the
GPU box; it is git-ignored.  There is no CPU fallback: if the build or the load fails the package raises.
"""

from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "libqsv.so"
PYHELP_PATH = PKG_DIR / "_qsvpyhelp.so"  # CPython-API helper of the Python layer (csrc/pyhelp.c), optional
SOURCES = ["kernels.hip", "qsv_api.hip", "plan.cpp", "split.cpp", "sort.hip"]
HEADERS = ["kernels.hpp", "plan.hpp", "split.hpp", "gate_loop_gen.inc", "../../include/qsv.h"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found; libqsv cannot be built")


def have_hipcc() -> bool:
    try:
        _hipcc()
        return True
    except RuntimeError:
        return False


def pyhelp_stale() -> bool:
    """The helper's signatures follow csrc/pyhelp.c: an older build must not be called with the newer argument lists."""
    src = CSRC / "pyhelp.c"
    return not PYHELP_PATH.exists() or PYHELP_PATH.stat().st_mtime < src.stat().st_mtime


def build_pyhelp(force: bool = False) -> "Path | None":
    """gcc csrc/pyhelp.c against this interpreter's headers; returns None when they (or gcc) are not there -- the Python
    layer then packs parameter vectors with array.fromlist."""
    import sysconfig

    src = CSRC / "pyhelp.c"
    if (PYHELP_PATH.exists() and not force and PYHELP_PATH.stat().st_mtime >= src.stat().st_mtime
            and (not LIB_PATH.exists() or PYHELP_PATH.stat().st_mtime >= LIB_PATH.stat().st_mtime)):
        return PYHELP_PATH
    include = sysconfig.get_paths().get("include")
    gcc = shutil.which("gcc")
    if not gcc or not include or not (Path(include) / "Python.h").exists():
        return None
    tmp = PYHELP_PATH.with_suffix(f".so.{os.getpid()}.tmp")  # (ranks that start together each write their own file)
    if not LIB_PATH.exists():
        return None  # the helper links against libqsv.so (it drives qsv_eval_begin / push / end itself)
    res = subprocess.run([gcc, "-O2", "-shared", "-fPIC", f"-I{include}", str(src), "-o", str(tmp), f"-L{PKG_DIR}", "-lqsv",
                          "-Wl,-rpath,$ORIGIN"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        tmp.unlink(missing_ok=True)
        return None
    os.replace(tmp, PYHELP_PATH)
    return PYHELP_PATH


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    built = LIB_PATH.stat().st_mtime
    return any((CSRC / f).resolve().stat().st_mtime > built for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False, defines: tuple = (), lib_path: Path = LIB_PATH) -> Path:
    """Compile every HIP source for gfx950 into queasars_amd/libqsv.so and return its path.

    ``defines`` / ``lib_path`` build a diagnostic variant next to it (scripts/stamps.py: -DQSV_STAMPS)."""
    if not force and not defines and not needs_build():
        build_pyhelp(False)
        return LIB_PATH
    obj_dir = PKG_DIR / ("build" if not defines else "build_" + "".join(c if c.isalnum() else "_" for c in "_".join(defines).lower()))
    obj_dir.mkdir(exist_ok=True)
    hipcc = _hipcc()
    common = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    common += [f"-D{d}" for d in defines]
    procs = []
    objs = []
    for src in SOURCES:
        obj = obj_dir / (src + ".o")
        objs.append(str(obj))
        cmd = [hipcc, *common, "-c", str(CSRC / src), "-o", str(obj)]
        if src.endswith(".cpp"):
            cmd.insert(1, "-x")
            cmd.insert(2, "hip")
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, proc in procs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    tmp = lib_path.with_suffix(f".so.{os.getpid()}.tmp")
    link = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(tmp), *objs]
    res = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout}")
    os.replace(tmp, lib_path)
    if not defines:
        build_pyhelp(True)
    return lib_path


if __name__ == "__main__":
    print(build(force=True, verbose=True))
