"""kmeans_gpu_amd -- Python host binding of libkmeans_hip.so (ctypes over the C ABI of
include/kmeans_hip.h).

It mirrors the reference crate's public surface (core/src/lib.rs:24-165):

    ImageProcessor().palette(color_count, image, algo)  -> list of RGBA8
    ImageProcessor().find(image, colors, reduce_mode)   -> image
    ImageProcessor().reduce(color_count, image, algo, reduce_mode) -> image

Images are numpy uint8 arrays of shape (height, width, 4) (tightly packed RGBA8,
core/src/image.rs:20-48).  `Lloyd` exposes the device-pointer building blocks used by the
multi-GPU layer (Group / GroupLloyd over kmg_group_*) and by bench.py; PyTorch only
supplies device memory, streams and torch.distributed there.

There is no CPU fallback: if the shared library is missing or no HIP device is usable, the
constructors raise.
"""
import ctypes as C
import enum
import os

import numpy as np

__all__ = ["ImageProcessor", "Algorithm", "ReduceMode", "Lloyd", "ApplyPlan", "Group", "GroupLloyd", "GroupOptions", "KmgError", "lib",
           "library_path", "GROUP_FORCE_COLLECTIVES", "GROUP_LOOPBACK", "GROUP_CELLS", "GROUP_OVERLAP", "GROUP_FUSED_UPDATE",
           "resized_dims", "palette_to_centroids", "centroids_to_palette", "dither_threshold",
           "default_options", "Options"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG_ROOT = os.path.dirname(os.path.dirname(_HERE))          # .../kmeans-gpu_amd
# KMG_LIBRARY: another build of the same ABI -- tools/ loads lib/libkmeans_hip_tools.so (make tools: tuning switches and
# knock-outs compiled in) through it; the product library reads none of those switches
_LIB_PATH = os.environ.get("KMG_LIBRARY") or os.path.join(_PKG_ROOT, "lib", "libkmeans_hip.so")


class _RawDeviceArray:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _alias_tensor(ptr, n, typestr):
    """torch tensor over device memory the library owns (no copy): valid while the owning object lives"""
    import torch
    return torch.as_tensor(_RawDeviceArray(ptr, n, typestr), device="cuda")


class KmgError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"kmeans_hip error {status}: {message}")
        self.status = status


class Algorithm(enum.IntEnum):          # core/src/lib.rs:215-219
    Kmeans = 0
    Octree = 1


class ReduceMode(enum.IntEnum):         # core/src/lib.rs:234-239
    Replace = 0
    Dither = 1
    Meld = 2


class Options(C.Structure):             # include/kmeans_hip.h kmg_options
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("shrink_max_dim", C.c_uint32),
                ("max_iterations", C.c_uint32), ("check_period", C.c_uint32), ("convergence", C.c_float),
                ("strategy", C.c_int32)]


# kmg_options.strategy (include/kmeans_hip.h KMG_STRATEGY_*): results are identical either way, only the time differs
STRATEGY_AUTO, STRATEGY_SCAN, STRATEGY_TABLE, STRATEGY_MASK_WORDS = 0, 1, 2, 4
_STRATEGY_NAMES = {"auto": STRATEGY_AUTO, "scan": STRATEGY_SCAN, "brute": STRATEGY_SCAN, "table": STRATEGY_TABLE}
_default_strategy = STRATEGY_AUTO
_live = None            # weak set of the live ImageProcessor / Group objects (set_strategy reaches them)


def _strategy_value(strategy):
    if isinstance(strategy, str):
        v = 0
        for part in strategy.replace("|", "+").split("+"):
            part = part.strip().lower()
            if part == "mask_words":
                v |= STRATEGY_MASK_WORDS
            elif part in _STRATEGY_NAMES:
                v |= _STRATEGY_NAMES[part]
            else:
                raise ValueError(f"unknown strategy {strategy!r}")
        return v
    return int(strategy)


def set_strategy(strategy):
    """kmg_processor_set_strategy on every live processor of this process (ImageProcessor objects and the members of Group
    objects) and the default of the ones created later: "auto" (the library's cost models), "scan" (alias "brute": per-pixel
    scans), "table" (colour table / candidate lists), optionally "+mask_words".  What the tests and tools flip between runs."""
    global _default_strategy
    _default_strategy = _strategy_value(strategy)
    for obj in list(_live or ()):
        obj.set_strategy(_default_strategy)


def _register(obj):
    global _live
    if _live is None:
        import weakref
        _live = weakref.WeakSet()
    _live.add(obj)


MAX_DEVICES = 16                        # KMG_MAX_DEVICES
UNIQUE_ID_BYTES = 128                   # KMG_UNIQUE_ID_BYTES
GROUP_FORCE_COLLECTIVES, GROUP_LOOPBACK = 1, 2                  # kmg_group_options.flags
GROUP_CELLS, GROUP_OVERLAP, GROUP_FUSED_UPDATE = 1, 2, 4        # kmg_group_lloyd_bind flags


class GroupOptions(C.Structure):        # include/kmeans_hip.h kmg_group_options
    _fields_ = [("struct_size", C.c_uint32), ("n_devices", C.c_uint32), ("devices", C.c_int32 * MAX_DEVICES),
                ("flags", C.c_uint32), ("processor", Options)]


def library_path():
    return _LIB_PATH


_lib = None

# every symbol include/kmeans_hip.h declares
SYMBOLS = [
    "kmg_last_error", "kmg_version", "kmg_host_alloc", "kmg_host_free", "kmg_default_options", "kmg_processor_create",
    "kmg_processor_create_ex", "kmg_processor_destroy", "kmg_processor_set_strategy", "kmg_palette", "kmg_find", "kmg_reduce",
    "kmg_palette_to_centroids", "kmg_centroids_to_palette", "kmg_octree_palette", "kmg_dev_rgb_to_lab",
    "kmg_resized_dims",
    "kmg_dev_resize", "kmg_lloyd_create", "kmg_lloyd_destroy", "kmg_lloyd_set_centroids",
    "kmg_lloyd_get_centroids", "kmg_lloyd_init_centroids", "kmg_lloyd_init_step", "kmg_lloyd_init_pick_band",
    "kmg_lloyd_set_centroid_rgba", "kmg_init_first_key", "kmg_lloyd_assign_accumulate",
    "kmg_lloyd_assign_partials", "kmg_lloyd_reduce_partials", "kmg_lloyd_labels", "kmg_lloyd_reserve_cus", "kmg_lloyd_bind_image",
    "kmg_lloyd_unbind_image", "kmg_debug_bound_image", "kmg_lloyd_prepare", "kmg_debug_check_table", "kmg_debug_table_stats", "kmg_debug_check_pairs", "kmg_debug_check_dither_masks", "kmg_debug_check_meld_masks", "kmg_kernel_name",
    "kmg_lloyd_profile", "kmg_lloyd_profile_read",
    "kmg_lloyd_update", "kmg_lloyd_assign_update", "kmg_lloyd_set_cell_share", "kmg_lloyd_labels_from_tables",
    "kmg_lloyd_table_buffers", "kmg_lloyd_accumulate_into", "kmg_lloyd_labels_from_tables_update", "kmg_lloyd_histogram_buffer", "kmg_lloyd_rebuild_from_histogram", "kmg_debug_block_counts", "kmg_debug_idle_blocks", "kmg_debug_encode_table_check", "kmg_debug_division_check", "kmg_lloyd_converged_count", "kmg_lloyd_iterate", "kmg_lloyd_flush", "kmg_lloyd_run", "kmg_dev_apply", "kmg_apply_plan_create", "kmg_apply_plan_run", "kmg_apply_plan_destroy",
    "kmg_dither_threshold",
    "kmg_default_group_options", "kmg_group_create", "kmg_group_unique_id", "kmg_group_create_rank", "kmg_group_destroy",
    "kmg_group_info", "kmg_group_processor", "kmg_group_stream", "kmg_group_palette", "kmg_group_find", "kmg_group_reduce",
    "kmg_group_reduce_batch", "kmg_group_lloyd_create", "kmg_group_lloyd_destroy", "kmg_group_lloyd_bind",
    "kmg_group_lloyd_set_centroids", "kmg_group_lloyd_get_centroids", "kmg_group_lloyd_init", "kmg_group_lloyd_prime",
    "kmg_group_lloyd_step", "kmg_group_lloyd_sync", "kmg_group_lloyd_run", "kmg_group_lloyd_member", "kmg_group_lloyd_create_batch", "kmg_group_lloyd_bind_batch", "kmg_group_lloyd_set_centroids_image",
    "kmg_group_lloyd_get_centroids_image", "kmg_group_lloyd_run_batch",
]


def lib():
    """Load libkmeans_hip.so (built in-tree by `make -C kmeans-gpu_amd`).  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f"{_LIB_PATH} not found: build it with `make -C {_PKG_ROOT}` "
                          "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                          "There is no fallback implementation.")
    # When PyTorch shares the process (tests, bench.py, the sharded driver) it must be imported
    # BEFORE libkmeans_hip.so is loaded: torch bundles its own libamdhip64, and two different HIP
    # runtimes in one process cannot both own the GPU ("no ROCm-capable device is detected").
    # With torch's runtime already mapped, our DT_NEEDED libamdhip64 resolves to that same copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_LIB_PATH)
    vp, u8p, u32p, f32p, i64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    L.kmg_last_error.restype = C.c_char_p
    L.kmg_version.restype = C.c_char_p
    L.kmg_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.kmg_host_free.argtypes = [vp]
    L.kmg_host_free.restype = None
    L.kmg_default_options.argtypes = [C.POINTER(Options)]
    L.kmg_default_options.restype = None
    L.kmg_processor_create.argtypes = [C.POINTER(vp)]
    L.kmg_processor_create_ex.argtypes = [C.POINTER(Options), C.POINTER(vp)]
    L.kmg_processor_destroy.argtypes = [vp]
    L.kmg_processor_destroy.restype = None
    L.kmg_palette.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, u8p, C.POINTER(C.c_uint32)]
    L.kmg_find.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, u8p, C.c_uint32, C.c_int, u8p]
    L.kmg_reduce.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, u8p]
    L.kmg_palette_to_centroids.argtypes = [u8p, C.c_uint32, f32p]
    L.kmg_centroids_to_palette.argtypes = [f32p, C.c_uint32, u8p]
    L.kmg_octree_palette.argtypes = [u8p, C.c_uint64, C.c_uint32, u8p, C.POINTER(C.c_uint32)]
    L.kmg_dev_rgb_to_lab.argtypes = [vp, u8p, C.c_uint64, f32p, vp]
    L.kmg_resized_dims.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.kmg_resized_dims.restype = None
    L.kmg_dev_resize.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u8p, vp]
    L.kmg_lloyd_create.argtypes = [vp, C.c_uint32, C.POINTER(vp)]
    L.kmg_lloyd_destroy.argtypes = [vp]
    L.kmg_lloyd_destroy.restype = None
    L.kmg_lloyd_set_centroids.argtypes = [vp, f32p, vp]
    L.kmg_lloyd_get_centroids.argtypes = [vp, f32p, vp]
    L.kmg_lloyd_init_centroids.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, vp]
    L.kmg_lloyd_init_step.argtypes = [vp, u8p, C.c_uint64, C.c_uint64, C.c_uint32, vp, vp]
    L.kmg_lloyd_init_pick_band.argtypes = [vp, u8p, C.c_uint64, C.c_uint64, vp, vp, vp]
    L.kmg_lloyd_set_centroid_rgba.argtypes = [vp, C.c_uint32, vp, vp]
    L.kmg_init_first_key.argtypes = [C.c_uint32, C.c_uint32]
    L.kmg_init_first_key.restype = C.c_uint64
    L.kmg_lloyd_assign_accumulate.argtypes = [vp, u8p, C.c_uint64, u32p, i64p, vp]
    L.kmg_lloyd_assign_partials.argtypes = [vp, u8p, C.c_uint64, u32p, vp]
    L.kmg_lloyd_reduce_partials.argtypes = [vp, C.c_uint64, i64p, vp]
    L.kmg_lloyd_labels.argtypes = [vp, u8p, C.c_uint64, u32p, vp]
    L.kmg_lloyd_reserve_cus.argtypes = [vp, C.c_uint32]
    L.kmg_lloyd_bind_image.argtypes = [vp, u8p, C.c_uint64, vp]
    L.kmg_debug_table_stats.argtypes = [vp, C.POINTER(C.c_uint64), vp]
    L.kmg_debug_check_pairs.argtypes = [vp, C.POINTER(C.c_uint64), vp]
    L.kmg_debug_bound_image.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.kmg_debug_check_dither_masks.argtypes = [vp, f32p, C.c_uint32, C.POINTER(C.c_uint64), vp]
    L.kmg_debug_check_meld_masks.argtypes = [vp, f32p, C.c_uint32, C.POINTER(C.c_uint64), vp]
    L.kmg_kernel_name.argtypes = [C.c_int]
    L.kmg_kernel_name.restype = C.c_char_p
    L.kmg_lloyd_profile.argtypes = [vp, C.c_int]
    L.kmg_lloyd_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.kmg_lloyd_unbind_image.argtypes = [vp]
    L.kmg_lloyd_prepare.argtypes = [vp, u8p, C.c_uint64, C.c_int, C.POINTER(C.c_int), vp]
    L.kmg_debug_check_table.argtypes = [vp, C.POINTER(C.c_uint64), vp]
    L.kmg_lloyd_update.argtypes = [vp, i64p, vp]
    L.kmg_lloyd_assign_update.argtypes = [vp, u8p, C.c_uint64, u32p, i64p, C.c_int, vp]
    L.kmg_lloyd_set_cell_share.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.kmg_lloyd_labels_from_tables.argtypes = [vp, u8p, C.c_uint64, u32p, vp]
    L.kmg_debug_block_counts.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.kmg_debug_idle_blocks.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.kmg_debug_encode_table_check.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.kmg_debug_division_check.argtypes = [vp, C.c_float, C.POINTER(C.c_uint64)]
    L.kmg_lloyd_histogram_buffer.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.kmg_lloyd_rebuild_from_histogram.argtypes = [vp, C.c_uint64, vp]
    L.kmg_lloyd_table_buffers.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.kmg_lloyd_converged_count.argtypes = [vp, C.POINTER(C.c_uint32), vp]
    L.kmg_lloyd_accumulate_into.argtypes = [vp, u8p, C.c_uint64, i64p, vp]
    L.kmg_lloyd_labels_from_tables_update.argtypes = [vp, u8p, C.c_uint64, u32p, i64p, vp]
    L.kmg_lloyd_iterate.argtypes = [vp, u8p, C.c_uint64, u32p, i64p, C.c_int, vp]
    L.kmg_lloyd_flush.argtypes = [vp, vp]
    L.kmg_lloyd_run.argtypes = [vp, u8p, C.c_uint64, u32p, C.POINTER(C.c_uint32), vp]
    L.kmg_dev_apply.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, C.c_uint32, f32p, C.c_uint32, C.c_int, u8p, vp]
    L.kmg_apply_plan_create.argtypes = [vp, f32p, C.c_uint32, C.c_int, C.c_uint64, vp, C.POINTER(vp)]
    L.kmg_apply_plan_run.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, C.c_uint32, u8p, vp]
    L.kmg_apply_plan_destroy.argtypes = [vp, C.c_int]
    L.kmg_apply_plan_destroy.restype = None
    L.kmg_dither_threshold.argtypes = [f32p, C.c_uint32, C.POINTER(C.c_float)]
    L.kmg_processor_set_strategy.argtypes = [vp, C.c_int]
    L.kmg_default_group_options.argtypes = [C.POINTER(GroupOptions)]
    L.kmg_default_group_options.restype = None
    L.kmg_group_create.argtypes = [C.POINTER(GroupOptions), C.POINTER(vp)]
    L.kmg_group_unique_id.argtypes = [vp]
    L.kmg_group_create_rank.argtypes = [C.POINTER(GroupOptions), vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.kmg_group_destroy.argtypes = [vp]
    L.kmg_group_destroy.restype = None
    L.kmg_group_info.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    L.kmg_group_processor.argtypes = [vp, C.c_uint32]
    L.kmg_group_processor.restype = vp
    L.kmg_group_stream.argtypes = [vp, C.c_uint32]
    L.kmg_group_stream.restype = vp
    L.kmg_group_palette.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, u8p, C.POINTER(C.c_uint32)]
    L.kmg_group_find.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, u8p, C.c_uint32, C.c_int, u8p]
    L.kmg_group_reduce.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, u8p]
    L.kmg_group_reduce_batch.argtypes = [vp, C.c_uint32, C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32,
                                         C.c_int, C.c_int, C.POINTER(vp)]
    L.kmg_group_lloyd_create.argtypes = [vp, C.c_uint32, C.POINTER(vp)]
    L.kmg_group_lloyd_destroy.argtypes = [vp]
    L.kmg_group_lloyd_destroy.restype = None
    L.kmg_group_lloyd_bind.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32,
                                       C.POINTER(vp), C.c_uint32]
    L.kmg_group_lloyd_set_centroids.argtypes = [vp, f32p]
    L.kmg_group_lloyd_get_centroids.argtypes = [vp, f32p]
    for name in ("init", "prime", "step", "sync"):
        getattr(L, "kmg_group_lloyd_" + name).argtypes = [vp]
    L.kmg_group_lloyd_run.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.kmg_group_lloyd_member.argtypes = [vp, C.c_uint32, C.POINTER(C.c_int)]
    L.kmg_group_lloyd_member.restype = vp
    L.kmg_group_lloyd_create_batch.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.kmg_group_lloyd_bind_batch.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                             C.POINTER(C.c_uint32), C.POINTER(vp), C.c_uint32]
    L.kmg_group_lloyd_set_centroids_image.argtypes = [vp, C.c_uint32, f32p]
    L.kmg_group_lloyd_get_centroids_image.argtypes = [vp, C.c_uint32, f32p]
    L.kmg_group_lloyd_run_batch.argtypes = [vp, C.POINTER(C.c_uint32)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise KmgError(rc, lib().kmg_last_error().decode("utf-8", "replace"))


def _np_ptr(a):
    return C.c_void_p(a.ctypes.data)


def default_options():
    o = Options()
    lib().kmg_default_options(C.byref(o))
    return o


def resized_dims(width, height, max_size=256):
    """InputTexture::resized dimension rule (core/src/structures.rs:79-89)."""
    nw, nh = C.c_uint32(), C.c_uint32()
    lib().kmg_resized_dims(width, height, max_size, C.byref(nw), C.byref(nh))
    return nw.value, nh.value


def palette_to_centroids(colors):
    """CentroidsBuffer::fixed_centroids (core/src/structures.rs:523-553): RGBA8 -> (L,a,b,1)."""
    pal = np.ascontiguousarray(colors, np.uint8).reshape(-1, 4)
    out = np.empty((pal.shape[0], 4), np.float32)
    _check(lib().kmg_palette_to_centroids(_np_ptr(pal), pal.shape[0], _np_ptr(out)))
    return out


def centroids_to_palette(centroids4):
    """CentroidsBuffer::pull_values (core/src/structures.rs:581-617)."""
    c = np.ascontiguousarray(centroids4, np.float32).reshape(-1, 4)
    out = np.empty((c.shape[0], 4), np.uint8)
    _check(lib().kmg_centroids_to_palette(_np_ptr(c), c.shape[0], _np_ptr(out)))
    return out


def octree_palette(pixels, color_count):
    """ColorTree::{add_color, reduce} (core/src/octree.rs): the reference's CPU octree quantiser."""
    px = np.ascontiguousarray(pixels, np.uint8).reshape(-1, 4)
    out = np.empty((max(min(int(color_count), px.shape[0]), 1), 4), np.uint8)
    cnt = C.c_uint32()
    _check(lib().kmg_octree_palette(_np_ptr(px), px.shape[0], int(color_count), _np_ptr(out), C.byref(cnt)))
    return out[:cnt.value].copy()


def dither_threshold(centroids4):
    c = np.ascontiguousarray(centroids4, np.float32).reshape(-1, 4)
    t = C.c_float()
    _check(lib().kmg_dither_threshold(_np_ptr(c), c.shape[0], C.byref(t)))
    return t.value


def _image(image):
    a = np.ascontiguousarray(image, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("image must be a (height, width, 4) uint8 RGBA array")
    return a


class _PinnedPool:
    """Result arrays of large images in page-locked host memory (kmg_host_alloc): a fresh pageable array of 256 MiB costs the
    call tens of milliseconds of page faults, a pinned block is copied to by DMA.  A block returns to the pool when the array
    that wraps it is garbage collected and serves the next result of the same size; at most `limit` bytes are kept."""

    def __init__(self, limit=1 << 30, threshold=16 << 20):
        self.free, self.kept, self.limit, self.threshold = {}, 0, limit, threshold

    def _give_back(self, ptr, nbytes):
        if self.kept + nbytes <= self.limit:
            self.free.setdefault(nbytes, []).append(ptr)
            self.kept += nbytes
        else:
            lib().kmg_host_free(C.c_void_p(ptr))

    def array(self, shape):
        import weakref
        nbytes = int(np.prod(shape))
        if nbytes < self.threshold or os.environ.get("KMG_PINNED_RESULTS", "1") == "0":
            return None
        blocks = self.free.get(nbytes)
        if blocks:
            ptr = blocks.pop()
            self.kept -= nbytes
        else:
            p = C.c_void_p()
            if lib().kmg_host_alloc(nbytes, C.byref(p)) != 0 or not p.value:
                return None
            ptr = p.value
        buf = (C.c_uint8 * nbytes).from_address(ptr)
        weakref.finalize(buf, self._give_back, ptr, nbytes)      # (the numpy array keeps `buf` alive as its base)
        return np.frombuffer(buf, dtype=np.uint8).reshape(shape)


_pinned = _PinnedPool()


def _result(img, out):
    if out is None:
        pinned = _pinned.array(img.shape)
        return pinned if pinned is not None else np.empty_like(img)
    if out.dtype != np.uint8 or out.shape != img.shape or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous uint8 array of the image's shape")
    return out


class ImageProcessor:
    """Mirror of `kmeans_color_gpu::ImageProcessor` (core/src/lib.rs:24-165)."""

    def __init__(self, device=-1, shrink_max_dim=256, max_iterations=128, check_period=8,
                 convergence=1.0, strategy=None):
        self._h = C.c_void_p()
        o = default_options()
        o.device = device
        o.shrink_max_dim = shrink_max_dim
        o.max_iterations = max_iterations
        o.check_period = check_period
        o.convergence = convergence
        o.strategy = _default_strategy if strategy is None else _strategy_value(strategy)
        _check(lib().kmg_processor_create_ex(C.byref(o), C.byref(self._h)))
        self.options = o
        _register(self)

    def set_strategy(self, strategy):
        """kmg_processor_set_strategy: "auto" | "scan" | "table" [+ "mask_words"] (or the KMG_STRATEGY_* bits)"""
        if self._h.value:
            _check(lib().kmg_processor_set_strategy(self._h, _strategy_value(strategy)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().kmg_processor_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._h

    # lib.rs:67-77
    def palette(self, color_count, image, algo=Algorithm.Kmeans):
        img = _image(image)
        h, w = img.shape[:2]
        out = np.empty((max(int(color_count), 1), 4), np.uint8)     # octree returns <= color_count colours
        cnt = C.c_uint32()
        _check(lib().kmg_palette(self._h, _np_ptr(img), w, h, int(color_count), int(algo), _np_ptr(out), C.byref(cnt)))
        return out[:cnt.value].copy()

    # lib.rs:79-114
    def find(self, image, colors, reduce_mode=ReduceMode.Replace, out=None):
        """out: optional (height, width, 4) uint8 array that receives the result (the C ABI writes into the caller's buffer;
        the reference returns a fresh Vec, which is what out=None does)"""
        img = _image(image)
        h, w = img.shape[:2]
        pal = np.ascontiguousarray(colors, np.uint8).reshape(-1, 4)
        out = _result(img, out)
        _check(lib().kmg_find(self._h, _np_ptr(img), w, h, _np_ptr(pal), pal.shape[0], int(reduce_mode), _np_ptr(out)))
        return out

    # lib.rs:116-164
    def reduce(self, color_count, image, algo=Algorithm.Kmeans, reduce_mode=ReduceMode.Replace, out=None):
        img = _image(image)
        h, w = img.shape[:2]
        out = _result(img, out)
        _check(lib().kmg_reduce(self._h, _np_ptr(img), w, h, int(color_count), int(algo), int(reduce_mode), _np_ptr(out)))
        return out

    # ---- device-pointer helpers (torch tensors supply the memory) -------------------------
    def rgb_to_lab(self, d_rgba, n_pixels, d_lab3, stream=0):
        _check(lib().kmg_dev_rgb_to_lab(self._h, C.c_void_p(d_rgba), n_pixels, C.c_void_p(d_lab3), C.c_void_p(stream)))

    def resize(self, d_rgba, width, height, new_width, new_height, d_out, stream=0):
        _check(lib().kmg_dev_resize(self._h, C.c_void_p(d_rgba), width, height, new_width, new_height,
                                    C.c_void_p(d_out), C.c_void_p(stream)))

    def apply(self, d_rgba, width, rows, row0, centroids4, mode, d_out, stream=0):
        c = np.ascontiguousarray(centroids4, np.float32).reshape(-1, 4)
        _check(lib().kmg_dev_apply(self._h, C.c_void_p(d_rgba), width, rows, row0, _np_ptr(c), c.shape[0],
                                   int(mode), C.c_void_p(d_out), C.c_void_p(stream)))

    def apply_plan(self, centroids4, mode, n_pixels_hint, stream=0):
        """the output pass as a plan (kmg_apply_plan_*): tables built once, then `run` per row band, asynchronously"""
        return ApplyPlan(self, centroids4, mode, n_pixels_hint, stream)

    def debug_block_counts(self):
        """(device blocks allocated with hipMalloc so far, blocks handed out again)"""
        out = (C.c_uint64 * 2)()
        _check(lib().kmg_debug_block_counts(self._h, out))
        return int(out[0]), int(out[1])

    def debug_idle_blocks(self):
        """(blocks the processor holds idle right now, their bytes)"""
        out = (C.c_uint64 * 2)()
        _check(lib().kmg_debug_idle_blocks(self._h, out))
        return int(out[0]), int(out[1])

    def debug_encode_table_check(self):
        """floats whose sRGB8 byte from the meld pass's threshold table differs from the encode's own byte (all floats tried)"""
        out = C.c_uint64(0)
        _check(lib().kmg_debug_encode_table_check(self._h, C.byref(out)))
        return int(out.value)

    def debug_division_check(self, c):
        """(mismatches, smallest |x| bits, largest |x| bits) of the device's x / c against IEEE division over every float x"""
        out = (C.c_uint64 * 3)()
        _check(lib().kmg_debug_division_check(self._h, C.c_float(c), out))
        return int(out[0]), int(out[1]), int(out[2])

    def debug_check_dither_masks(self, centroids4, stream=0):
        """exhaustive check of the pruned dither pass's candidate masks; returns the violation count"""
        c = np.ascontiguousarray(centroids4, np.float32).reshape(-1, 4)
        v = C.c_uint64()
        _check(lib().kmg_debug_check_dither_masks(self._h, _np_ptr(c), c.shape[0], C.byref(v), C.c_void_p(stream)))
        return int(v.value)

    def debug_check_meld_masks(self, centroids4, stream=0):
        """exhaustive check of the pruned meld pass's candidate masks; returns the violation count"""
        c = np.ascontiguousarray(centroids4, np.float32).reshape(-1, 4)
        v = C.c_uint64()
        _check(lib().kmg_debug_check_meld_masks(self._h, _np_ptr(c), c.shape[0], C.byref(v), C.c_void_p(stream)))
        return int(v.value)


class ApplyPlan:
    """kmg_apply_plan_*: the output pass for one centroid table, band by band, without host synchronisation"""

    def __init__(self, processor, centroids4, mode, n_pixels_hint, stream=0):
        c = np.ascontiguousarray(centroids4, np.float32).reshape(-1, 4)
        self._h = C.c_void_p()
        _check(lib().kmg_apply_plan_create(processor.handle, _np_ptr(c), c.shape[0], int(mode), int(n_pixels_hint), C.c_void_p(stream),
                                           C.byref(self._h)))

    def run(self, d_rgba, width, rows, row0, d_out, stream=0):
        _check(lib().kmg_apply_plan_run(self._h, C.c_void_p(d_rgba), width, rows, row0, C.c_void_p(d_out), C.c_void_p(stream)))

    def close(self, synchronise=True):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().kmg_apply_plan_destroy(self._h, int(bool(synchronise)))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Lloyd:
    """One Lloyd problem (an image or a row band of it) on one device: kmg_lloyd_* of the C ABI.
    Pointers are raw device addresses (e.g. torch.Tensor.data_ptr()); `stream` a hipStream_t."""

    def __init__(self, processor, k):
        self._p = processor
        self.k = int(k)
        self._h = C.c_void_p()
        _check(lib().kmg_lloyd_create(processor.handle, self.k, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().kmg_lloyd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_centroids(self, centroids4, stream=0):
        c = np.ascontiguousarray(centroids4, np.float32).reshape(self.k, 4)
        _check(lib().kmg_lloyd_set_centroids(self._h, _np_ptr(c), C.c_void_p(stream)))

    def get_centroids(self, stream=0):
        out = np.empty((self.k, 4), np.float32)
        _check(lib().kmg_lloyd_get_centroids(self._h, _np_ptr(out), C.c_void_p(stream)))
        return out

    def init_centroids(self, d_rgba, width, height, stream=0):
        _check(lib().kmg_lloyd_init_centroids(self._h, C.c_void_p(d_rgba), width, height, C.c_void_p(stream)))

    # sharded (row band) initialisation steps -- what kmg_group_lloyd_init drives (tests/sharded_harness.py sharded_init)
    def init_step(self, d_rgba, n_local, first_index, j, d_key, stream=0):
        _check(lib().kmg_lloyd_init_step(self._h, C.c_void_p(d_rgba or None), n_local, first_index, j,
                                         C.c_void_p(d_key), C.c_void_p(stream)))

    def init_pick_band(self, d_rgba, n_local, first_index, d_key, d_colour2, stream=0):
        _check(lib().kmg_lloyd_init_pick_band(self._h, C.c_void_p(d_rgba or None), n_local, first_index,
                                              C.c_void_p(d_key), C.c_void_p(d_colour2), C.c_void_p(stream)))

    def set_centroid_rgba(self, j, d_colour, stream=0):
        _check(lib().kmg_lloyd_set_centroid_rgba(self._h, j, C.c_void_p(d_colour), C.c_void_p(stream)))

    @staticmethod
    def init_first_key(width, height):
        return int(lib().kmg_init_first_key(width, height))

    def assign_accumulate(self, d_rgba, n_pixels, d_labels, d_acc4, stream=0):
        _check(lib().kmg_lloyd_assign_accumulate(self._h, C.c_void_p(d_rgba), n_pixels,
                                                 C.c_void_p(d_labels or None), C.c_void_p(d_acc4 or None),
                                                 C.c_void_p(stream)))

    def assign_partials(self, d_rgba, n_pixels, d_labels, stream=0):
        _check(lib().kmg_lloyd_assign_partials(self._h, C.c_void_p(d_rgba), n_pixels,
                                               C.c_void_p(d_labels or None), C.c_void_p(stream)))

    def labels(self, d_rgba, n_pixels, d_labels, stream=0):
        """labels only, for the current centroid table"""
        _check(lib().kmg_lloyd_labels(self._h, C.c_void_p(d_rgba), n_pixels, C.c_void_p(d_labels), C.c_void_p(stream)))

    def reserve_cus(self, n_cus):
        """leave n_cus compute units without a label-pass workgroup, for a collective that runs beside the pass"""
        _check(lib().kmg_lloyd_reserve_cus(self._h, int(n_cus)))

    def reduce_partials(self, n_pixels, d_acc4, stream=0):
        _check(lib().kmg_lloyd_reduce_partials(self._h, n_pixels, C.c_void_p(d_acc4), C.c_void_p(stream)))

    def bind_image(self, d_rgba, n_pixels, stream=0):
        """build the colour table of this image once; later passes on it use the table"""
        _check(lib().kmg_lloyd_bind_image(self._h, C.c_void_p(d_rgba), n_pixels, C.c_void_p(stream)))

    def prepare(self, d_rgba, n_pixels, want_labels=True, stream=0):
        """one-time per-image preparation; returns the chosen strategy ("scan" or "table")"""
        st = C.c_int()
        _check(lib().kmg_lloyd_prepare(self._h, C.c_void_p(d_rgba), n_pixels, int(bool(want_labels)),
                                       C.byref(st), C.c_void_p(stream)))
        return "table" if st.value == 1 else "scan"

    KERNEL_IDS = {"k_assign": 0, "k_reduce_partials": 1, "k_update": 2, "k_cell_candidates": 3, "k_cube": 4,
                  "k_labels": 5}

    def profile(self, enable=True):
        """start / stop per-launch HIP-event timing: True = every kernel, False = stop, or an iterable
        of kernel names (KERNEL_IDS) to time only those"""
        if enable is True:
            mask = -1
        elif not enable:
            mask = 0
        else:
            mask = 0
            for name in enable:
                mask |= 1 << self.KERNEL_IDS[name]
        _check(lib().kmg_lloyd_profile(self._h, mask))

    def profile_read(self):
        """{kernel name: (total ms, launches)} since the last read; synchronises the events"""
        n = 6
        ms = (C.c_double * n)()
        cnt = (C.c_uint32 * n)()
        _check(lib().kmg_lloyd_profile_read(self._h, ms, cnt))
        return {lib().kmg_kernel_name(i).decode(): (ms[i], cnt[i]) for i in range(n) if cnt[i]}

    def unbind_image(self):
        _check(lib().kmg_lloyd_unbind_image(self._h))

    def debug_check_table(self, stream=0):
        """exhaustive check over all 2^24 colours: (bound violations, arg-mins missing from the cell masks,
        per-colour labels that differ from the brute-force arg-min)"""
        out = (C.c_uint64 * 3)()
        _check(lib().kmg_debug_check_table(self._h, out, C.c_void_p(stream)))
        return int(out[0]), int(out[1]), int(out[2])

    def debug_table_stats(self, stream=0):
        out = (C.c_uint64 * 14)()
        _check(lib().kmg_debug_table_stats(self._h, out, C.c_void_p(stream)))
        names = ["occupied_cells", "candidates_total", "cells_one_candidate", "max_candidates",
                 "cells_one_label", "occupied_sub_cells", "sub_cells_one_label", "distinct_colours",
                 "sub_cells_decided", "sub_cells_scanned", "scan_candidates", "cells_unlisted",
                 "candidates_pruned", "sub_cells_pruned_to_one"]
        return dict(zip(names, (int(v) for v in out)))

    def debug_bound_image(self):
        """(occupied cells, hot cells) of the bound image"""
        out = (C.c_uint64 * 2)()
        _check(lib().kmg_debug_bound_image(self._h, out))
        return int(out[0]), int(out[1])

    def debug_check_pairs(self, stream=0):
        """(mismatching colours, pixels resolved by the LDS pair entries, pixels) of the last table pass"""
        out = (C.c_uint64 * 3)()
        _check(lib().kmg_debug_check_pairs(self._h, out, C.c_void_p(stream)))
        return int(out[0]), int(out[1]), int(out[2])

    def update(self, d_acc4, stream=0):
        _check(lib().kmg_lloyd_update(self._h, C.c_void_p(d_acc4), C.c_void_p(stream)))

    def set_cell_share(self, part, parts, stream=0):
        """cell-sharded cube pass: later assign passes of the bound image visit share `part` of `parts` of its occupied cells"""
        _check(lib().kmg_lloyd_set_cell_share(self._h, int(part), int(parts), C.c_void_p(stream)))

    def accumulate_into(self, d_rgba, n_pixels, d_acc4, stream=0):
        """the cube pass of the bound image ADDS its sums to d_acc4 as it stands (no hand-over launch)"""
        _check(lib().kmg_lloyd_accumulate_into(self._h, C.c_void_p(d_rgba), int(n_pixels), C.c_void_p(d_acc4), C.c_void_p(stream)))

    def labels_from_tables_update(self, d_rgba, n_pixels, d_labels, d_acc4, stream=0):
        """label map from the tables as they stand + kmg_lloyd_update from d_acc4, which is left zero (one launch)"""
        _check(lib().kmg_lloyd_labels_from_tables_update(self._h, C.c_void_p(d_rgba), int(n_pixels), C.c_void_p(d_labels), C.c_void_p(d_acc4),
                                                         C.c_void_p(stream)))

    def labels_from_tables(self, d_rgba, n_pixels, d_labels, stream=0):
        """the label pass with the label tables as they stand, on any pixels whose colours occur in the bound image"""
        _check(lib().kmg_lloyd_labels_from_tables(self._h, C.c_void_p(d_rgba), n_pixels, C.c_void_p(d_labels), C.c_void_p(stream)))

    def histogram_buffer(self):
        """(pointer, bytes) of the bound image's colour histogram (2^24 u32, cell-major colour order)"""
        a, na = C.c_void_p(), C.c_uint64()
        _check(lib().kmg_lloyd_histogram_buffer(self._h, C.byref(a), C.byref(na)))
        return a.value, na.value

    def rebuild_from_histogram(self, n_pixels, stream=0):
        """re-derive the binding from the (all-reduced) histogram, which now counts n_pixels pixels"""
        _check(lib().kmg_lloyd_rebuild_from_histogram(self._h, int(n_pixels), C.c_void_p(stream)))

    def histogram_tensor(self):
        """the bound image's colour histogram as a torch tensor that ALIASES the library's buffer (int32 [2^24]) -- for the
        all-reduce of a cell-sharded loop (kmg_group_lloyd_* with KMG_GROUP_CELLS; tests/sharded_harness.py)"""
        ptr, nbytes = self.histogram_buffer()
        return _alias_tensor(ptr, nbytes // 4, "<i4")

    def table_tensors(self):
        """(per-colour labels uint8 [2^24], cell entries int32 [32768]) of the bound image, k <= 256, as torch tensors that
        ALIAS the library's label tables -- for the all-gather of a cell-sharded loop"""
        lab, nlab, ent, nent = self.table_buffers()
        if nlab != 1 << 24:
            raise KmgError("table_tensors: the cell-sharded loop needs k <= 256")
        n_cells, n_sub = 32768, 32768 * 8
        return _alias_tensor(lab, nlab, "|u1"), _alias_tensor(ent + 2 * (n_sub + n_cells), n_cells, "<i4")

    def table_buffers(self):
        """(per-colour label table pointer, bytes, cell entry table pointer, bytes) of the bound image"""
        a, na, b, nb = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
        _check(lib().kmg_lloyd_table_buffers(self._h, C.byref(a), C.byref(na), C.byref(b), C.byref(nb)))
        return a.value, na.value, b.value, nb.value

    def assign_update(self, d_rgba, n_pixels, d_labels, d_acc4, do_update=True, stream=0):
        """assign (labels optional, sums into d_acc4), then -- do_update -- the centroid update from those sums"""
        _check(lib().kmg_lloyd_assign_update(self._h, C.c_void_p(d_rgba), n_pixels, C.c_void_p(d_labels or None),
                                             C.c_void_p(d_acc4), int(bool(do_update)), C.c_void_p(stream)))

    def iterate(self, d_rgba, n_pixels, d_labels, d_acc4, update_first=True, stream=0):
        """one Lloyd iteration, asynchronous; the label map is complete after flush() (or a device sync)"""
        _check(lib().kmg_lloyd_iterate(self._h, C.c_void_p(d_rgba), n_pixels, C.c_void_p(d_labels or None),
                                       C.c_void_p(d_acc4), int(bool(update_first)), C.c_void_p(stream)))

    def flush(self, stream=0):
        """make `stream` wait for the label passes that iterate() left running on the library's side stream"""
        _check(lib().kmg_lloyd_flush(self._h, C.c_void_p(stream)))

    def converged_count(self, stream=0):
        n = C.c_uint32()
        _check(lib().kmg_lloyd_converged_count(self._h, C.byref(n), C.c_void_p(stream)))
        return n.value

    def run(self, d_rgba, n_pixels, d_labels=0, stream=0):
        it = C.c_uint32()
        _check(lib().kmg_lloyd_run(self._h, C.c_void_p(d_rgba), n_pixels, C.c_void_p(d_labels or None),
                                   C.byref(it), C.c_void_p(stream)))
        return it.value


class _BorrowedProcessor(ImageProcessor):
    """a member processor of a Group: the group owns it"""

    def __init__(self, handle):                      # pylint: disable=super-init-not-called
        self._h = C.c_void_p(handle)

    def close(self):
        self._h = C.c_void_p()


class _BorrowedLloyd(Lloyd):
    """a member kmg_lloyd of a GroupLloyd (profiling, statistics): the group owns it"""

    def __init__(self, handle, k):                   # pylint: disable=super-init-not-called
        self._h = C.c_void_p(handle)
        self.k = int(k)

    def close(self):
        self._h = C.c_void_p()


class Group:
    """kmg_group_*: ImageProcessor::new (core/src/lib.rs:38-65) over a device LIST -- one processor, compute stream and RCCL
    rank per device.  `devices` = HIP ordinals of this process's ranks (None = every visible device).  One process per GPU:
    rank 0 makes `unique_id()`, the host runtime hands it to every process, each passes it with its `first_rank` and `world`.
    palette / find / reduce mirror ImageProcessor's and give the same bytes."""

    def __init__(self, devices=None, flags=0, unique_id=None, first_rank=0, world=None, shrink_max_dim=256, max_iterations=128,
                 check_period=8, convergence=1.0, strategy=None):
        o = GroupOptions()
        lib().kmg_default_group_options(C.byref(o))
        if devices is not None:
            devices = [int(d) for d in devices]
            if len(devices) > MAX_DEVICES:
                raise ValueError("too many devices")
            o.n_devices = len(devices)
            for i, d in enumerate(devices):
                o.devices[i] = d
        o.flags = int(flags)
        o.processor.shrink_max_dim = shrink_max_dim
        o.processor.max_iterations = max_iterations
        o.processor.check_period = check_period
        o.processor.convergence = convergence
        o.processor.strategy = _default_strategy if strategy is None else _strategy_value(strategy)
        self._h = C.c_void_p()
        if unique_id is None:
            _check(lib().kmg_group_create(C.byref(o), C.byref(self._h)))
        else:
            if world is None:
                raise ValueError("Group(unique_id=...) is one process of a multi-process world: pass world (and first_rank)")
            uid = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
            _check(lib().kmg_group_create_rank(C.byref(o), uid, int(first_rank), int(world), C.byref(self._h)))
        a, b, c, v = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_int()
        _check(lib().kmg_group_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(v)))
        self.n_local, self.first_rank, self.world, self.rccl_version = a.value, b.value, c.value, v.value
        self.options = o
        _register(self)

    def set_strategy(self, strategy):
        """kmg_processor_set_strategy on every member processor"""
        if not self._h.value:
            return
        L = lib()
        for i in range(self.n_local):
            _check(L.kmg_processor_set_strategy(L.kmg_group_processor(self._h, i), _strategy_value(strategy)))

    @staticmethod
    def unique_id():
        """ncclGetUniqueId through the library (loads RCCL): 128 bytes for every process of the job"""
        buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
        _check(lib().kmg_group_unique_id(buf))
        return bytes(buf)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().kmg_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._h

    def processor(self, i=0):
        return _BorrowedProcessor(lib().kmg_group_processor(self._h, int(i)))

    def stream(self, i=0):
        """local device i's compute stream (hipStream_t as an integer): everything the group enqueues there runs on it"""
        return int(lib().kmg_group_stream(self._h, int(i)) or 0)

    def palette(self, color_count, image, algo=Algorithm.Kmeans):
        img = _image(image)
        h, w = img.shape[:2]
        out = np.empty((max(int(color_count), 1), 4), np.uint8)
        cnt = C.c_uint32()
        _check(lib().kmg_group_palette(self._h, _np_ptr(img), w, h, int(color_count), int(algo), _np_ptr(out), C.byref(cnt)))
        return out[:cnt.value].copy()

    def find(self, image, colors, reduce_mode=ReduceMode.Replace, out=None):
        img = _image(image)
        h, w = img.shape[:2]
        pal = np.ascontiguousarray(colors, np.uint8).reshape(-1, 4)
        out = _result(img, out)
        _check(lib().kmg_group_find(self._h, _np_ptr(img), w, h, _np_ptr(pal), pal.shape[0], int(reduce_mode), _np_ptr(out)))
        return out

    def reduce(self, color_count, image, algo=Algorithm.Kmeans, reduce_mode=ReduceMode.Replace, out=None):
        img = _image(image)
        h, w = img.shape[:2]
        out = _result(img, out)
        _check(lib().kmg_group_reduce(self._h, _np_ptr(img), w, h, int(color_count), int(algo), int(reduce_mode), _np_ptr(out)))
        return out

    def reduce_batch(self, color_count, images, algo=Algorithm.Kmeans, reduce_mode=ReduceMode.Replace):
        """whole images per device, no collective (BASELINE config 4 as placed); returns the list of results"""
        imgs = [_image(im) for im in images]
        outs = [_result(im, None) for im in imgs]          # (large results from the pool of page-locked blocks, as reduce() does)
        n = len(imgs)
        src = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        dst = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        ws = (C.c_uint32 * n)(*[im.shape[1] for im in imgs])
        hs = (C.c_uint32 * n)(*[im.shape[0] for im in imgs])
        _check(lib().kmg_group_reduce_batch(self._h, n, src, ws, hs, int(color_count), int(algo), int(reduce_mode), dst))
        return outs


class GroupLloyd:
    """kmg_group_lloyd_*: one Lloyd problem -- or a BATCH of n_images problems, each image tiled over all ranks -- over row bands
    resident on the group's devices (modules.rs:763-840 + ONE RCCL all-reduce of the n_images x k x 4 int64 sums per iteration).
    Pointers are raw device addresses."""

    def __init__(self, group, k, n_images=1):
        self._g = group
        self.k = int(k)
        self.n_images = int(n_images)
        self._h = C.c_void_p()
        if self.n_images == 1:
            _check(lib().kmg_group_lloyd_create(group.handle, self.k, C.byref(self._h)))
        else:
            _check(lib().kmg_group_lloyd_create_batch(group.handle, self.k, self.n_images, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().kmg_group_lloyd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bind(self, d_rgba, row0, rows, width, height, d_labels=None, flags=0):
        n = self._g.n_local
        if not (len(d_rgba) == len(row0) == len(rows) == n):
            raise ValueError("one band per local device")
        px = (C.c_void_p * n)(*[int(p) or None for p in d_rgba])
        r0 = (C.c_uint32 * n)(*[int(v) for v in row0])
        rs = (C.c_uint32 * n)(*[int(v) for v in rows])
        lab = (C.c_void_p * n)(*[int(p) or None for p in d_labels]) if d_labels is not None else None
        _check(lib().kmg_group_lloyd_bind(self._h, px, r0, rs, int(width), int(height), lab, int(flags)))

    def bind_batch(self, d_rgba, row0, rows, widths, heights, d_labels=None, flags=0):
        """d_rgba / row0 / rows / d_labels: [image][local device]; widths / heights: per image (or one number for all)"""
        n, m = self._g.n_local, self.n_images
        if isinstance(widths, int):
            widths = [widths] * m
        if isinstance(heights, int):
            heights = [heights] * m
        flat = lambda a: [v for per_image in a for v in per_image]
        if not (len(d_rgba) == len(row0) == len(rows) == m) or any(len(x) != n for x in d_rgba):
            raise ValueError("one band per image and local device")
        px = (C.c_void_p * (n * m))(*[int(p) or None for p in flat(d_rgba)])
        r0 = (C.c_uint32 * (n * m))(*[int(v) for v in flat(row0)])
        rs = (C.c_uint32 * (n * m))(*[int(v) for v in flat(rows)])
        ws = (C.c_uint32 * m)(*[int(v) for v in widths])
        hs = (C.c_uint32 * m)(*[int(v) for v in heights])
        lab = (C.c_void_p * (n * m))(*[int(p) or None for p in flat(d_labels)]) if d_labels is not None else None
        _check(lib().kmg_group_lloyd_bind_batch(self._h, px, r0, rs, ws, hs, lab, int(flags)))

    def set_centroids(self, centroids4, image=None):
        c = np.ascontiguousarray(centroids4, np.float32).reshape(self.k, 4)
        if image is None:
            _check(lib().kmg_group_lloyd_set_centroids(self._h, _np_ptr(c)))
        else:
            _check(lib().kmg_group_lloyd_set_centroids_image(self._h, int(image), _np_ptr(c)))

    def get_centroids(self, image=None):
        out = np.empty((self.k, 4), np.float32)
        if image is None:
            _check(lib().kmg_group_lloyd_get_centroids(self._h, _np_ptr(out)))
        else:
            _check(lib().kmg_group_lloyd_get_centroids_image(self._h, int(image), _np_ptr(out)))
        return out

    def run_batch(self):
        """every image to its own convergence; the iteration each one stopped at"""
        its = (C.c_uint32 * self.n_images)()
        _check(lib().kmg_group_lloyd_run_batch(self._h, its))
        return [int(v) for v in its]

    def init(self):
        _check(lib().kmg_group_lloyd_init(self._h))

    def prime(self):
        _check(lib().kmg_group_lloyd_prime(self._h))

    def step(self):
        _check(lib().kmg_group_lloyd_step(self._h))

    def sync(self):
        _check(lib().kmg_group_lloyd_sync(self._h))

    def run(self):
        it = C.c_uint32()
        _check(lib().kmg_group_lloyd_run(self._h, C.byref(it)))
        return it.value

    def member(self, i=0):
        """(Lloyd of local device i, "table" | "scan")"""
        st = C.c_int()
        h = lib().kmg_group_lloyd_member(self._h, int(i), C.byref(st))
        return _BorrowedLloyd(h, self.k), ("table" if st.value == 1 else "scan")
