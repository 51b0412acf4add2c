"""Compile libmkb_hip.so for gfx950 with hipcc (in-tree; cross-compiles without a GPU).

    python -m mkb_amd.csrc.build [--force] [--save-temps]

Only files that changed are recompiled (object files under mkb_amd/csrc/_obj/).
"""
import os
import pathlib
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = pathlib.Path(__file__).resolve().parent
OUT = HERE.parent / "libmkb_hip.so"
OBJ = HERE / "_obj"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(HERE.glob("*.hip"))


def _needs(src, obj):
    if not obj.exists():
        return True
    newest = max([src.stat().st_mtime] + [h.stat().st_mtime for h in HERE.glob("*.h")]
                 + [(HERE.parent.parent / "include" / "mkb_hip.h").stat().st_mtime])
    return obj.stat().st_mtime < newest


def build(force=False, save_temps=False, verbose=True, extra_flags=(), out=None, objdir=None):
    """extra_flags / out / objdir build an experimental variant (tools/kbench.py) next to the product library."""
    global OBJ
    out = OUT if out is None else pathlib.Path(out)
    saved_obj = OBJ
    if objdir is not None:
        OBJ = pathlib.Path(objdir)
    try:
        return _build(force, save_temps, verbose, tuple(extra_flags), out)
    finally:
        OBJ = saved_obj


def _build(force, save_temps, verbose, extra_flags, OUT):
    OBJ.mkdir(exist_ok=True)
    jobs = []
    for src in sources():
        obj = OBJ / (src.stem + ".o")
        if force or _needs(src, obj):
            cmd = [HIPCC, *FLAGS, *extra_flags, "-c", str(src), "-o", str(obj)]
            if save_temps:
                cmd.insert(1, "-save-temps=obj")
            jobs.append(cmd)
    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd, cwd=str(OBJ))
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [str(OBJ / (s.stem + ".o")) for s in sources()]
    if jobs or not OUT.exists():
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT), *objs]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
