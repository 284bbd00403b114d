"""ctypes binding of the CPU oracle (oracle/libkmg_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(ORACLE_DIR, "libkmg_oracle.so")

MODE_REPLACE, MODE_DITHER, MODE_MELD = 0, 1, 2


def build(force=False):
    src = [os.path.join(ORACLE_DIR, f) for f in ("kmg_oracle.c", "kmg_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _declare(L):
    u8p, u32p, f32p, i64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                             C.POINTER(C.c_float), C.POINTER(C.c_int64))
    L.orc_srgb_lut.argtypes = [f32p]
    L.orc_cbrt.argtypes = [C.c_float]; L.orc_cbrt.restype = C.c_float
    L.orc_pow_inv_2p4.argtypes = [C.c_float]; L.orc_pow_inv_2p4.restype = C.c_float
    L.orc_rgb_to_lab.argtypes = [u8p, C.c_uint64, f32p]
    L.orc_cie94.argtypes = [f32p, f32p]; L.orc_cie94.restype = C.c_float
    L.orc_cie94_key.argtypes = [f32p, f32p]; L.orc_cie94_key.restype = C.c_float
    L.orc_assign.argtypes = [f32p, C.c_uint64, f32p, C.c_uint32, C.c_int, u32p]
    L.orc_accumulate.argtypes = [f32p, u32p, C.c_uint64, C.c_uint32, i64p]
    L.orc_finalize.argtypes = [i64p, C.c_uint32, C.c_float, f32p]; L.orc_finalize.restype = C.c_uint32
    L.orc_lloyd.argtypes = [f32p, C.c_uint64, C.c_uint32, f32p, u32p, C.c_uint32, C.c_uint32, C.c_float]
    L.orc_lloyd.restype = C.c_uint32
    L.orc_rand.argtypes = [C.c_float]; L.orc_rand.restype = C.c_float
    L.orc_init_centroids.argtypes = [f32p, C.c_uint32, C.c_uint32, C.c_uint32, f32p]
    L.orc_resized_dims.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p]
    L.orc_resize.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u8p]
    L.orc_dither_threshold.argtypes = [f32p, C.c_uint32]; L.orc_dither_threshold.restype = C.c_float
    L.orc_dither.argtypes = [f32p, C.c_uint32, C.c_uint32, f32p, C.c_uint32, u32p]
    L.orc_meld.argtypes = [f32p, C.c_uint32, C.c_uint32, f32p, C.c_uint32, f32p]
    L.orc_lab_to_rgba8.argtypes = [f32p, C.c_uint64, u8p]
    L.orc_palette_srgb8_to_lab.argtypes = [u8p, f32p]
    L.orc_palette_lab_to_srgb8.argtypes = [f32p, u8p]
    L.orc_extract_palette_kmeans.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, f32p]
    L.orc_extract_palette_kmeans.restype = C.c_uint32
    L.orc_apply.argtypes = [u8p, C.c_uint32, C.c_uint32, f32p, C.c_uint32, C.c_int, u8p]
    L.orc_find.argtypes = [u8p, C.c_uint32, C.c_uint32, u8p, C.c_uint32, C.c_int, u8p]
    L.orc_reduce.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, u8p]
    L.orc_palette.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, u8p]
    L.orc_synth_uniform.argtypes = [C.c_uint64, C.c_uint64, u8p]
    L.orc_assign_accumulate_rgba.argtypes = [u8p, C.c_uint64, f32p, C.c_uint32, u32p, i64p]
    L.orc_num_threads.restype = C.c_int
    L.orc_set_num_threads.argtypes = [C.c_int]


def _rgba(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.shape[-1] == 4
    return a


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- thin numpy wrappers -------------------------------------------------------------

def srgb_lut():
    out = np.empty(256, np.float32)
    lib().orc_srgb_lut(_p(out, C.c_float))
    return out


def cbrt(x):
    return float(lib().orc_cbrt(C.c_float(x)))


def pow_inv_2p4(c):
    return float(lib().orc_pow_inv_2p4(C.c_float(c)))


def rgb_to_lab(rgba):
    rgba = _rgba(rgba).reshape(-1, 4)
    out = np.empty((rgba.shape[0], 3), np.float32)
    lib().orc_rgb_to_lab(_p(rgba, C.c_uint8), rgba.shape[0], _p(out, C.c_float))
    return out


def cie94(one, second):
    a, b = _f32(one), _f32(second)
    return float(lib().orc_cie94(_p(a, C.c_float), _p(b, C.c_float)))


def cie94_key(one, second):
    a, b = _f32(one), _f32(second)
    return float(lib().orc_cie94_key(_p(a, C.c_float), _p(b, C.c_float)))


def centroids4(lab):
    lab = _f32(lab).reshape(-1, lab.shape[-1])
    if lab.shape[1] == 4:
        return lab
    out = np.ones((lab.shape[0], 4), np.float32)
    out[:, :3] = lab
    return out


def assign(lab3, cent4, literal=False):
    """literal: False / 0 = literal distance, hoisted terms (fast); True / 1 = orc_cie94 per pair; 2 = squared key only"""
    lab3 = _f32(lab3).reshape(-1, 3); cent4 = centroids4(cent4)
    out = np.empty(lab3.shape[0], np.uint32)
    lib().orc_assign(_p(lab3, C.c_float), lab3.shape[0], _p(cent4, C.c_float), cent4.shape[0],
                     int(literal), _p(out, C.c_uint32))
    return out


def accumulate(lab3, labels, k):
    lab3 = _f32(lab3).reshape(-1, 3)
    labels = np.ascontiguousarray(labels, np.uint32)
    acc = np.zeros((k, 4), np.int64)
    lib().orc_accumulate(_p(lab3, C.c_float), _p(labels, C.c_uint32), lab3.shape[0], k, _p(acc, C.c_int64))
    return acc


def finalize(acc, cent4, convergence=1.0):
    acc = np.ascontiguousarray(acc, np.int64); cent4 = centroids4(cent4).copy()
    n = lib().orc_finalize(_p(acc, C.c_int64), cent4.shape[0], C.c_float(convergence), _p(cent4, C.c_float))
    return cent4, int(n)


def lloyd(lab3, cent4, max_iterations=128, check_period=8, convergence=1.0):
    lab3 = _f32(lab3).reshape(-1, 3); cent4 = centroids4(cent4).copy()
    labels = np.empty(lab3.shape[0], np.uint32)
    it = lib().orc_lloyd(_p(lab3, C.c_float), lab3.shape[0], cent4.shape[0], _p(cent4, C.c_float),
                         _p(labels, C.c_uint32), max_iterations, check_period, C.c_float(convergence))
    return cent4, labels, int(it)


def rand(seed):
    return float(lib().orc_rand(C.c_float(seed)))


def init_centroids(lab3, w, h, k):
    lab3 = _f32(lab3).reshape(-1, 3)
    assert lab3.shape[0] == w * h
    out = np.zeros((k, 4), np.float32)
    lib().orc_init_centroids(_p(lab3, C.c_float), w, h, k, _p(out, C.c_float))
    return out


def resized_dims(w, h, max_size=256):
    nw, nh = C.c_uint32(), C.c_uint32()
    lib().orc_resized_dims(w, h, max_size, C.byref(nw), C.byref(nh))
    return nw.value, nh.value


def resize(rgba, nw, nh):
    rgba = _rgba(rgba); h, w = rgba.shape[:2]
    out = np.empty((nh, nw, 4), np.uint8)
    lib().orc_resize(_p(rgba, C.c_uint8), w, h, nw, nh, _p(out, C.c_uint8))
    return out


def dither_threshold(cent4):
    cent4 = centroids4(cent4)
    return float(lib().orc_dither_threshold(_p(cent4, C.c_float), cent4.shape[0]))


def dither(lab3, w, h, cent4):
    lab3 = _f32(lab3).reshape(-1, 3); cent4 = centroids4(cent4)
    out = np.empty(w * h, np.uint32)
    lib().orc_dither(_p(lab3, C.c_float), w, h, _p(cent4, C.c_float), cent4.shape[0], _p(out, C.c_uint32))
    return out


def meld(lab3, w, h, cent4):
    lab3 = _f32(lab3).reshape(-1, 3); cent4 = centroids4(cent4)
    out = np.empty((w * h, 3), np.float32)
    lib().orc_meld(_p(lab3, C.c_float), w, h, _p(cent4, C.c_float), cent4.shape[0], _p(out, C.c_float))
    return out


def lab_to_rgba8(lab3):
    lab3 = _f32(lab3).reshape(-1, 3)
    out = np.empty((lab3.shape[0], 4), np.uint8)
    lib().orc_lab_to_rgba8(_p(lab3, C.c_float), lab3.shape[0], _p(out, C.c_uint8))
    return out


def palette_srgb8_to_lab(rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8).reshape(-1)[:3].copy()
    out = np.empty(3, np.float32)
    lib().orc_palette_srgb8_to_lab(_p(rgb, C.c_uint8), _p(out, C.c_float))
    return out


def palette_lab_to_srgb8(lab):
    lab = _f32(lab).reshape(-1)[:3].copy()
    out = np.empty(3, np.uint8)
    lib().orc_palette_lab_to_srgb8(_p(lab, C.c_float), _p(out, C.c_uint8))
    return out


def extract_palette_kmeans(rgba, k, shrink_max_dim=256):
    rgba = _rgba(rgba); h, w = rgba.shape[:2]
    out = np.zeros((k, 4), np.float32)
    it = lib().orc_extract_palette_kmeans(_p(rgba, C.c_uint8), w, h, k, shrink_max_dim, _p(out, C.c_float))
    return out, int(it)


def apply(rgba, cent4, mode):
    rgba = _rgba(rgba); h, w = rgba.shape[:2]; cent4 = centroids4(cent4)
    out = np.empty_like(rgba)
    lib().orc_apply(_p(rgba, C.c_uint8), w, h, _p(cent4, C.c_float), cent4.shape[0], mode, _p(out, C.c_uint8))
    return out


def find(rgba, palette_rgba, mode):
    rgba = _rgba(rgba); h, w = rgba.shape[:2]
    pal = _rgba(palette_rgba).reshape(-1, 4)
    out = np.empty_like(rgba)
    lib().orc_find(_p(rgba, C.c_uint8), w, h, _p(pal, C.c_uint8), pal.shape[0], mode, _p(out, C.c_uint8))
    return out


def reduce(rgba, k, mode):
    rgba = _rgba(rgba); h, w = rgba.shape[:2]
    out = np.empty_like(rgba)
    lib().orc_reduce(_p(rgba, C.c_uint8), w, h, k, mode, _p(out, C.c_uint8))
    return out


def palette(rgba, k):
    rgba = _rgba(rgba); h, w = rgba.shape[:2]
    out = np.empty((k, 4), np.uint8)
    lib().orc_palette(_p(rgba, C.c_uint8), w, h, k, _p(out, C.c_uint8))
    return out


def synth_uniform(seed, n):
    out = np.empty((n, 4), np.uint8)
    lib().orc_synth_uniform(seed, n, _p(out, C.c_uint8))
    return out


def assign_accumulate_rgba(rgba, cent4):
    rgba = _rgba(rgba).reshape(-1, 4); cent4 = centroids4(cent4)
    k = cent4.shape[0]
    labels = np.empty(rgba.shape[0], np.uint32)
    acc = np.zeros((k, 4), np.int64)
    lib().orc_assign_accumulate_rgba(_p(rgba, C.c_uint8), rgba.shape[0], _p(cent4, C.c_float), k,
                                     _p(labels, C.c_uint32), _p(acc, C.c_int64))
    return labels, acc


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


# ---- Algorithm::Octree --------------------------------------------------------------------
def _declare_octree(L):
    u8p = C.POINTER(C.c_uint8)
    L.orc_octree_palette.argtypes = [u8p, C.c_uint64, C.c_uint32, u8p]; L.orc_octree_palette.restype = C.c_uint32
    L.orc_palette_octree.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, u8p]; L.orc_palette_octree.restype = C.c_uint32
    L.orc_reduce_octree.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, u8p]


def octree_palette(pixels, color_count):
    _declare_octree(lib())
    px = _rgba(pixels).reshape(-1, 4)
    out = np.empty((max(min(int(color_count), px.shape[0]), 1), 4), np.uint8)
    n = lib().orc_octree_palette(_p(px, C.c_uint8), px.shape[0], int(color_count), _p(out, C.c_uint8))
    return out[:n].copy()


def palette_octree(rgba, k):
    _declare_octree(lib())
    rgba = _rgba(rgba); h, w = rgba.shape[:2]
    out = np.empty((max(int(k), 1), 4), np.uint8)
    n = lib().orc_palette_octree(_p(rgba, C.c_uint8), w, h, int(k), _p(out, C.c_uint8))
    return out[:n].copy()


def reduce_octree(rgba, k, mode):
    _declare_octree(lib())
    rgba = _rgba(rgba); h, w = rgba.shape[:2]
    out = np.empty_like(rgba)
    lib().orc_reduce_octree(_p(rgba, C.c_uint8), w, h, int(k), int(mode), _p(out, C.c_uint8))
    return out
