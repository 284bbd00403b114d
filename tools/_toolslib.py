"""Tools that flip tuning switches or knock-outs (KMG_ASSIGN_PPT, KMG_HOT_CELLS,
KMG_*_GRID, KMG_CUBE_REPL, KMG_CUBE_SMALL, KMG_DITHER_SORT, KMG_DITHER_STATS) need the TOOLS build of the library: the product
library does not read them.  use_tools_library() builds lib/libkmeans_hip_tools.so when it is missing or stale (make tools) and
points the binding at it (KMG_LIBRARY) -- call it before importing kmeans_gpu_amd, or pass its result in a child's environment."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "kmeans-gpu_amd")
TOOLS_LIB = os.path.join(PKG, "lib", "libkmeans_hip_tools.so")


def use_tools_library(env=None):
    subprocess.run(["make", "-j8", "-C", PKG, "tools"], check=True, stdout=subprocess.DEVNULL)
    target = os.environ if env is None else env
    target["KMG_LIBRARY"] = TOOLS_LIB
    return TOOLS_LIB
