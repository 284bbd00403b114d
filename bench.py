#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Lloyd-iteration hot path on MI355X.

Metric (BASELINE.json): pixels/sec per Lloyd iteration at 8192x8192, k = 256, plus the fraction of
the HBM roofline.  A "step" is ONE Lloyd iteration over the whole (sharded) image: centroid update
from the global accumulators, a label for every pixel (written to HBM, 4 B/px, every iteration --
like the reference's find_centroid dispatch) and the per-cluster sums of the new assignment, and
for N > 1 the RCCL all-reduce of the k x 4 int64 accumulators.

N = 1: synthetic 8192x8192 RGBA (splitmix64 seed 0x5EED0003), k = 256.
N > 1 (default --scaling strong): the SAME 8192x8192 image over the N GPUs -- row bands of 8192/N rows for the label map and,
       with the colour table, the cube pass sharded by cells (KMG_GROUP_CELLS: histogram all-reduce once, per iteration
       the k x 4 all-reduce and an all-gather of the label tables).  The weak-scaling figure (one 8192-row band per GPU, one
       k-means problem over the 8192 x 8192 N image) is measured right after and reported under `extra.weak_scaling_*`.
N > 1, --scaling weak: only that.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))

WIDTH = 8192
ROWS_PER_GPU = 8192
K = 256
ALGORITHMIC_BYTES_PER_PIXEL = 8          # 4 B RGBA8 read + 4 B u32 label write
HBM_PEAK_GBPS = 8000.0                   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
HBM_ACHIEVABLE_GBPS = 6290.0             # same guide: 6.29 TB/s measured with a float4 copy
FP32_VECTOR_PEAK_TFLOPS = 157.3          # same guide: peak FP32 (vector)
FLOP_PER_PAIR, FLOP_PER_PIXEL = 17, 50   # SURVEY.md 8d "algorithmic flops": literal CIE94 per (pixel, centroid) + Lab conversion


def algorithmic_bytes(kernel, n_pixels, k):
    """per-launch algorithmic bytes of each kernel of the path"""
    if kernel in ("k_assign", "k_labels"):
        return ALGORITHMIC_BYTES_PER_PIXEL * n_pixels        # pixel in, label out
    if kernel == "k_cube":
        return (1 << 24) * (4 + (1 if k <= 256 else 2))      # colour counts in, colour labels out
    return None


def synthetic_image(kind, n_pixels, first, k, seed):
    """uint8 (n, 4) tensor on the GPU: `uniform` (headline), `blobs` (k Gaussian clusters, sigma 12),
    `photo` (the 768x513 fixture tiled to the width with a slow gradient so that tiles differ)"""
    import numpy as np
    import torch
    from kmeans_gpu_amd import synth
    if kind == "uniform":
        return synth.uniform_rgba_torch(seed, n_pixels, first=first, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(seed & 0x7FFFFFFF)
    if kind == "blobs":
        centres = torch.randint(0, 256, (k, 3), generator=g, device="cuda", dtype=torch.int32)
        out = torch.empty((n_pixels, 4), dtype=torch.uint8, device="cuda")
        for c0 in range(0, n_pixels, 1 << 24):
            m = min(1 << 24, n_pixels - c0)
            which = torch.randint(0, k, (m,), generator=g, device="cuda")
            px = centres[which].to(torch.float32) + 12.0 * torch.randn((m, 3), generator=g, device="cuda")
            out[c0:c0 + m, :3] = px.round().clamp_(0, 255).to(torch.uint8)
            out[c0:c0 + m, 3] = 255
        return out
    from PIL import Image
    tokyo = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "tokyo.png")).convert("RGBA"))
    rows = n_pixels // WIDTH
    big = np.tile(tokyo, (rows // tokyo.shape[0] + 1, WIDTH // tokyo.shape[1] + 1, 1))[:rows, :WIDTH].copy()
    yy, xx = np.mgrid[0:rows, 0:WIDTH]
    big[..., 0] = np.clip(big[..., 0].astype(int) + (xx >> 9), 0, 255).astype(np.uint8)
    big[..., 1] = np.clip(big[..., 1].astype(int) + (yy >> 9), 0, 255).astype(np.uint8)
    return torch.from_numpy(big.reshape(-1, 4)).cuda()


def _step(s, strategy, rgba, n_pixels, labels, acc, stream):
    """one Lloyd iteration of a single-GPU problem with its label map: assign, then the update (on the assign pass's last launch with
    the colour table -- kmg_lloyd_assign_update; the per-pixel scan does the same in its own launches)"""
    s.assign_update(rgba.data_ptr(), n_pixels, labels.data_ptr(), acc.data_ptr(), True, stream)


def other_distributions(proc, k, n_pixels, stream, steps=10):
    """SURVEY 8d secondary inputs, not part of `value`: the same iteration on `blobs` and on a tiled
    photograph, with the share of the pixels the label pass resolves from its LDS table alone."""
    import numpy as np
    import torch
    import kmeans_gpu_amd as kg
    extra = {}
    for kind in ("blobs", "photo"):
        rgba = synthetic_image(kind, n_pixels, 0, k, 0x5EED0B10)
        labels = torch.empty(n_pixels, dtype=torch.int32, device="cuda")
        acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
        sel = rgba[(torch.arange(k, device="cuda") * (n_pixels // k))].contiguous()
        lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
        proc.rgb_to_lab(sel.data_ptr(), k, lab.data_ptr(), stream)
        torch.cuda.synchronize()
        cent = np.ones((k, 4), np.float32)
        cent[:, :3] = lab.cpu().numpy()
        s = kg.Lloyd(proc, k)
        s.set_centroids(cent, stream)
        strategy = s.prepare(rgba.data_ptr(), n_pixels, True, stream)
        for _ in range(4):
            _step(s, strategy, rgba, n_pixels, labels, acc, stream)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            _step(s, strategy, rgba, n_pixels, labels, acc, stream)
        torch.cuda.synchronize()
        extra[f"{kind}_ms_per_step"] = (time.perf_counter() - t) / steps * 1e3
        extra[f"{kind}_strategy"] = strategy
        if strategy == "table" and k <= 256:
            s.assign_accumulate(rgba.data_ptr(), n_pixels, 0, acc.data_ptr(), stream)    # label tables of the current centroids
            _, resolved, total = s.debug_check_pairs(stream)
            extra[f"{kind}_pixels_resolved_in_lds"] = resolved / max(total, 1)
        s.close()
        del rgba, labels
    return extra


def cfg4_rank_share(proc, k, n_pixels, stream, steps=5):
    """BASELINE config 4 (16 images of 8192 x 8192 over 8 GPUs) as this build places it: WHOLE images per GPU, zero
    collectives (kmg_group_reduce_batch's split).  One rank's share = two images; time of one Lloyd iteration of both, label maps
    included.  Not part of `value`."""
    import numpy as np
    import torch
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    mine = [0, 8]                                                # image i on device i % 8
    images = [synth.uniform_rgba_torch(0x5EED0400 + i, n_pixels, device="cuda") for i in mine]
    labels = [torch.empty(n_pixels, dtype=torch.int32, device="cuda") for _ in mine]
    accs = [torch.zeros((k, 4), dtype=torch.int64, device="cuda") for _ in mine]
    lloyds, strategies = [], []
    for img in images:
        sel = img[(torch.arange(k, device="cuda") * (n_pixels // k))].contiguous()
        lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
        proc.rgb_to_lab(sel.data_ptr(), k, lab.data_ptr(), stream)
        torch.cuda.synchronize()
        cent = np.ones((k, 4), np.float32)
        cent[:, :3] = lab.cpu().numpy()
        s = kg.Lloyd(proc, k)
        s.set_centroids(cent, stream)
        strategies.append(s.prepare(img.data_ptr(), n_pixels, True, stream))
        lloyds.append(s)

    def iteration():
        for s, how, img, lab, acc in zip(lloyds, strategies, images, labels, accs):
            _step(s, how, img, n_pixels, lab, acc, stream)
    iteration(); iteration()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        iteration()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / steps * 1e3
    for s in lloyds:
        s.close()
    return {"cfg4_images_per_rank": len(mine), "cfg4_rank_share_ms_per_iteration": ms,
            "cfg4_collectives_per_iteration": 0}


def cfg4_native_batch(k, width, height, images=2):
    """BASELINE config 4 through the C ABI's own batch call: kmg_group_reduce_batch of one rank's share (two whole 8192 x 8192
    images on this GPU: upload, initialisation and Lloyd loop at FULL resolution, dither output pass, download) -- host buffers
    in and out, PCIe included, never part of `value`.  On a node the group spans the 8 devices and takes 16 images in the same
    call, the devices side by side."""
    import numpy as np
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    imgs = [synth.uniform_rgba_numpy(0x5EED0400 + i, width * height).reshape(height, width, 4) for i in range(images)]
    with kg.Group(devices=[0], shrink_max_dim=0) as g:
        times = []
        for _ in range(3):
            t = time.perf_counter()
            outs = g.reduce_batch(k, imgs, reduce_mode=kg.ReduceMode.Dither)
            times.append((time.perf_counter() - t) * 1e3)
        assert all(int(o[..., 3].min()) == 255 for o in outs)
    return {"cfg4_native_batch_images": images, "cfg4_native_batch_host_to_host_cold_ms": times[0],
            "cfg4_native_batch_host_to_host_warm_ms": min(times[1:])}


def cfg4_tiled_rank_share(k, width, height, steps=3, images=16, world=8, rank=3):
    """BASELINE config 4 split the way north_star words it, through the C ABI (kmg_group_lloyd_create_batch / _bind_batch): every image
    tiled over the 8 GPUs in row bands, ONE all-reduce of images x k x 4 int64 per iteration.  This is rank 3's share -- its 16 bands
    of 8192 x 1024 -- in a group of one rank with KMG_GROUP_FORCE_COLLECTIVES: the all-reduce is a real ncclAllReduce (of one rank:
    the other seven ranks' sums are missing from the numbers, not from the clock).  Time of one iteration of the rank's share, label
    maps included.  Not part of `value`."""
    import numpy as np
    import torch
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    r0, r1 = rank * height // world, (rank + 1) * height // world
    n_band, n = (r1 - r0) * width, width * height
    with _StdoutToStderr():
        group = kg.Group(devices=[torch.cuda.current_device()], flags=kg.GROUP_FORCE_COLLECTIVES, shrink_max_dim=0)
    proc = group.processor(0)
    stream = torch.cuda.current_stream().cuda_stream
    gl = kg.GroupLloyd(group, k, n_images=images)
    bands, labels = [], []
    for i in range(images):
        bands.append(synth.uniform_rgba_torch(0x5EED0400 + i, n_band, first=r0 * width, device="cuda"))
        labels.append(torch.empty(n_band, dtype=torch.int32, device="cuda"))
    gl.bind_batch([[b.data_ptr()] for b in bands], [[r0]] * images, [[r1 - r0]] * images, width, height, [[l.data_ptr()] for l in labels])
    for i in range(images):
        sel = synth.uniform_rgba_at(0x5EED0400 + i, np.arange(k, dtype=np.uint64) * np.uint64(n // k))
        d_sel = torch.from_numpy(sel).cuda()
        lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
        proc.rgb_to_lab(d_sel.data_ptr(), k, lab.data_ptr(), stream)
        torch.cuda.synchronize()
        cent = np.ones((k, 4), np.float32)
        cent[:, :3] = lab.cpu().numpy()
        gl.set_centroids(cent, image=i)
    with _StdoutToStderr():
        gl.prime()
        gl.step()
        gl.sync()
    t = time.perf_counter()
    for _ in range(steps):
        gl.step()
    gl.sync()
    ms = (time.perf_counter() - t) / steps * 1e3
    gl.close()
    group.close()
    return {"cfg4_tiled_rank_share_ms_per_iteration": ms, "cfg4_tiled_bands_per_rank": images,
            "cfg4_tiled_collectives_per_iteration": 1,
            "cfg4_tiled_driver": "kmg_group_lloyd_create_batch / _bind_batch / _step (C ABI), one ncclAllReduce of 16 x k x 4 int64 per iteration"}


CFG2_WIDTH, CFG2_HEIGHT, CFG2_K = 4096, 4096, 16
VALU_LANE_OPS_PEAK = 78.6e12             # fp32 vector lane-operations per second (157.3 TF/s counts an FMA as two)


def cfg2_timing(proc, stream, steps=20, strategies=("table", "scan"), profile_kernels=True, default_strategy="auto"):
    """BASELINE config 2: synthetic 4096 x 4096 (seed 0x5EED0002), k = 16, assign + update only -- the configuration of
    tests/test_gpu_table.py::test_cfg2_full_size_assign_update (centroids = shader Lab of the pixels at j * floor(N/k)).
    One step = one assignment with its label map and sums + the centroid update.  Both strategies are timed (the library's
    cost model picks one; `cfg2_strategy_auto` says which); the reported step is the auto choice's.  Not part of `value`."""
    import numpy as np
    import torch
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    n, k = CFG2_WIDTH * CFG2_HEIGHT, CFG2_K
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG2, n, device="cuda")
    labels = torch.empty(n, dtype=torch.int32, device="cuda")
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    sel = synth.uniform_rgba_at(synth.SEED_CFG2, np.arange(k, dtype=np.uint64) * np.uint64(n // k))
    d_sel = torch.from_numpy(sel).cuda()
    lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
    proc.rgb_to_lab(d_sel.data_ptr(), k, lab.data_ptr(), stream)
    torch.cuda.synchronize()
    cent = np.ones((k, 4), np.float32)
    cent[:, :3] = lab.cpu().numpy()
    out = {}
    try:
        kg.set_strategy("auto")
        s = kg.Lloyd(proc, k)
        s.set_centroids(cent, stream)
        out["cfg2_strategy_auto"] = s.prepare(rgba.data_ptr(), n, True, stream)
        s.close()
        for strategy in strategies:
            kg.set_strategy(strategy)
            s = kg.Lloyd(proc, k)
            s.set_centroids(cent, stream)
            torch.cuda.synchronize()
            t = time.perf_counter()
            s.prepare(rgba.data_ptr(), n, True, stream)
            torch.cuda.synchronize()
            out[f"cfg2_{strategy}_prepare_ms"] = (time.perf_counter() - t) * 1e3

            def step():
                s.assign_update(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), True, stream)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            out[f"cfg2_{strategy}_ms_per_step"] = (time.perf_counter() - t) / steps * 1e3
            if profile_kernels:
                s.profile(True)
                for _ in range(min(steps, 10)):
                    step()
                torch.cuda.synchronize()
                out[f"cfg2_{strategy}_kernels_ms"] = {nm: ms / cnt for nm, (ms, cnt) in s.profile_read().items()}
                s.profile(False)
            s.close()
    finally:
        kg.set_strategy(default_strategy)
    auto = out["cfg2_strategy_auto"]
    if f"cfg2_{auto}_ms_per_step" in out:
        out["cfg2_ms_per_step"] = out[f"cfg2_{auto}_ms_per_step"]
    del rgba, labels
    return out


def _bound(hbm_frac, valu_frac):
    """the roof a kernel sits under: the larger of its two fractions -- unless both are low: then neither bytes nor vector issue
    is what it waits for but chains of dependent instructions / memory round trips at the wave counts its registers allow"""
    if max(hbm_frac, valu_frac) < 0.35:
        return "issue"
    return "valu" if valu_frac > hbm_frac else "hbm"


def attach_valu_roof(table, ms_of, tj):
    """The roof that binds, per kernel of the line: `valu` = vector wave-instructions one launch executes (SQ_INSTS_VALU of
    a rocprofv3 PMC pass of the same loop, profiles/traffic.json `valu_wave_instructions`) x 64 lanes over the kernel's time
    and the fp32 vector issue peak (78.6 T lane-operations/s = 157.3 TF/s with an FMA as two); `bound` = the larger of the
    two fractions -- "hbm" or "valu"; "issue" when both are below 0.35 -- and "requests" for the label pass, whose time goes to divergent gathers the vector
    memory pipeline retires at ~1 lane per 2 clocks per CU (profiles/NOTES.md)."""
    counts = tj.get("valu_wave_instructions", {})
    for nm, row in table.items():
        if not isinstance(row, dict) or "valu" in row or nm not in counts or nm not in ms_of or not ms_of[nm]:
            continue
        row["valu_wave_instructions"] = counts[nm]
        row["valu"] = counts[nm] * 64 / (ms_of[nm] * 1e-3) / VALU_LANE_OPS_PEAK
        row["bound"] = "requests" if nm == "k_labels" else _bound(row.get("frac", 0.0), row["valu"])
        row["valu_source"] = tj.get("valu_source", "profiles/traffic.json")


def cfg2_roofline(extra, tj):
    """both roofs of SURVEY 8d for config 2: 8 B/px over the step against HBM, and the vector instructions the step's
    kernels execute (profiles/traffic.json `valu_wave_instructions`, rocprofv3 SQ_INSTS_VALU of the same loop) x 64 lanes
    over the fp32 vector issue peak"""
    if "cfg2_ms_per_step" not in extra:
        return None
    n = CFG2_WIDTH * CFG2_HEIGHT
    ms = extra["cfg2_ms_per_step"]
    r = {"algorithmic_bytes_per_launch": ALGORITHMIC_BYTES_PER_PIXEL * n,
         "achieved_GBps": ALGORITHMIC_BYTES_PER_PIXEL * n / (ms * 1e-3) / 1e9,
         "frac": ALGORITHMIC_BYTES_PER_PIXEL * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
         "hbm_floor_ms": ALGORITHMIC_BYTES_PER_PIXEL * n / (HBM_PEAK_GBPS * 1e9) * 1e3,
         "valu_floor_ms_literal_scan": n * (FLOP_PER_PAIR * CFG2_K + FLOP_PER_PIXEL) / (FP32_VECTOR_PEAK_TFLOPS * 1e12) * 1e3,
         "strategy": extra.get("cfg2_strategy_auto"),
         "note": "whole step (all launches of assign + update), host clock over the loop; 4096x4096, k=16, seed 0x5EED0002"}
    valu = tj.get("valu_wave_instructions", {}).get("cfg2_step")
    if valu:
        r["valu_wave_instructions"] = valu
        r["valu"] = valu * 64 / (ms * 1e-3) / VALU_LANE_OPS_PEAK
        r["bound"] = _bound(r["frac"], r["valu"])
        r["traffic"] = tj.get("bytes_per_launch", {}).get("cfg2_step")
    return r


def reduce_end_to_end(proc, rgba, width, height, k):
    """kmg_reduce (lib.rs:116-164) of the full-resolution image from and to HOST buffers: upload, initialisation, Lloyd loop,
    dither output pass, download -- PCIe included, never part of `value`.  The second call (warm processor: streams, blocks
    and pool memory are there) is the one reported."""
    import kmeans_gpu_amd as kg
    import numpy as np
    host = rgba.cpu().numpy().reshape(height, width, 4)
    times = []
    for _ in range(3):
        t = time.perf_counter()
        proc.reduce(k, host, reduce_mode=kg.ReduceMode.Dither)
        times.append((time.perf_counter() - t) * 1e3)
    # the same call into a result buffer the caller keeps (what the C ABI does; the figures above include creating -- and, from
    # the second call on, unmapping -- a fresh 256 MiB numpy array per call, as the reference's Vec-returning API would)
    out = np.empty_like(host)
    into = []
    for _ in range(3):
        t = time.perf_counter()
        proc.reduce(k, host, reduce_mode=kg.ReduceMode.Dither, out=out)
        into.append((time.perf_counter() - t) * 1e3)
    return {"reduce_host_to_host_cold_ms": times[0], "reduce_host_to_host_warm_ms": min(times[1:]),
            "reduce_host_to_host_into_callers_buffer_ms": min(into[1:])}


def default_call_timing():
    """BASELINE config 1 and the reference's DEFAULT call (lib.rs:116-164 with structures.rs:67-89: shrink to <= 256, init, Lloyd
    loop, output pass) from and to host buffers on the reference's own test image: wall time per warm call.  Launch-bound --
    a 256 x 171 working image -- and never part of `value`."""
    import numpy as np
    import kmeans_gpu_amd as kg
    from PIL import Image
    tokyo = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "tokyo.png")).convert("RGBA"))
    p = kg.ImageProcessor()                   # the reference's defaults
    out = {}

    def warm(fn, reps=5):
        ts = []
        for _ in range(reps):
            t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t) * 1e3)
        return min(ts[1:])
    out["cfg1_reduce_tokyo_k8_replace_ms"] = warm(lambda: p.reduce(8, tokyo, reduce_mode=kg.ReduceMode.Replace))
    out["default_palette_tokyo_k256_ms"] = warm(lambda: p.palette(256, tokyo))
    out["default_reduce_tokyo_k256_dither_ms"] = warm(lambda: p.reduce(256, tokyo, reduce_mode=kg.ReduceMode.Dither))
    p.close()
    return out


def output_pass_timing(proc, rgba, n_pixels, stream, sh=None, steps=3, prep_ms=None):
    """Not part of `value`: the other kernel family of the path, BASELINE config 5 -- find + ordered
    dither with the 64-entry resurrect_64 palette on the same 8192x8192 pixels (8 B/px algorithmic:
    4 B in, 4 B RGBA8 out), and the iteration without the per-pixel label map."""
    import numpy as np
    import torch
    import kmeans_gpu_amd as kg
    extra = {}
    try:
        from PIL import Image
        px = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "resurrect_64.png")).convert("RGBA")).reshape(-1, 4)
        pal = np.array(sorted(set(map(tuple, px))), np.uint8)
        cent = kg.palette_to_centroids(pal)
        out = torch.empty((n_pixels, 4), dtype=torch.uint8, device="cuda")
        for mode, name in ((kg.ReduceMode.Dither, "find_dither"), (kg.ReduceMode.Replace, "find_replace")):
            proc.apply(rgba.data_ptr(), WIDTH, n_pixels // WIDTH, 0, cent, mode, out.data_ptr(), stream)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(steps):
                proc.apply(rgba.data_ptr(), WIDTH, n_pixels // WIDTH, 0, cent, mode, out.data_ptr(), stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / steps * 1e3
            extra[f"{name}_k{len(pal)}_ms"] = ms
            extra[f"{name}_k{len(pal)}_hbm_frac"] = ALGORITHMIC_BYTES_PER_PIXEL * n_pixels / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
        if sh is not None:
            if sh.split_labels and k_of(sh) <= 256:
                sh.backend.assign_accumulate(rgba.data_ptr(), n_pixels, 0, sh.acc.data_ptr(), stream)   # label tables of the current centroids
                _, resolved, total = sh.backend.debug_check_pairs(stream)
                extra["uniform_pixels_resolved_in_lds"] = resolved / max(total, 1)

            # the iteration without its label map (update + sums).  kmg_lloyd_run's own iterations are cheaper still: the update
            # rides on the assign pass and the label pass's pair entries are derived once, before the final label map
            # (cfg3_lloyd_and_labels_ms / cfg3_iterations)
            def sums_only(iters):
                for _ in range(iters):
                    sh.backend.update(sh.acc.data_ptr(), stream)
                    sh.backend.assign_accumulate(rgba.data_ptr(), n_pixels, 0, sh.acc.data_ptr(), stream)
            sums_only(2)
            torch.cuda.synchronize()
            t = time.perf_counter()
            sums_only(10)
            torch.cuda.synchronize()
            extra["iteration_without_label_map_ms"] = (time.perf_counter() - t) / 10 * 1e3
        # BASELINE config 3 end to end: reference init at full resolution, Lloyd to convergence, label
        # map, dither output pass (the library's own loop, kmg_lloyd_run)
        k3 = sh.k if sh is not None else 256
        labels = torch.empty(n_pixels, dtype=torch.int32, device="cuda")
        s3 = kg.Lloyd(proc, k3)

        def stage(fn):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3, r
        s3.init_centroids(rgba.data_ptr(), WIDTH, n_pixels // WIDTH, stream)          # allocations + binding
        extra["cfg3_init_ms"], _ = stage(lambda: s3.init_centroids(rgba.data_ptr(), WIDTH, n_pixels // WIDTH, stream))
        extra["cfg3_lloyd_and_labels_ms"], extra["cfg3_iterations"] = stage(
            lambda: s3.run(rgba.data_ptr(), n_pixels, labels.data_ptr(), stream))
        cent3 = s3.get_centroids(stream)
        # (the first call at this k takes its scratch block: reported apart, like the init's first call)
        extra["cfg3_dither_cold_ms"], _ = stage(lambda: proc.apply(rgba.data_ptr(), WIDTH, n_pixels // WIDTH, 0, cent3,
                                                                   kg.ReduceMode.Dither, out.data_ptr(), stream))
        extra["cfg3_dither_ms"], _ = stage(lambda: proc.apply(rgba.data_ptr(), WIDTH, n_pixels // WIDTH, 0, cent3,
                                                              kg.ReduceMode.Dither, out.data_ptr(), stream))
        extra["cfg3_bind_ms"] = prep_ms
        # (the initialisation binds the image itself -- histogram, tie keys -- and the loop reuses that binding: cfg3_bind_ms is
        # part of cfg3_init_ms, not a term of the sum)
        extra["cfg3_total_ms"] = extra["cfg3_init_ms"] + extra["cfg3_lloyd_and_labels_ms"] + extra["cfg3_dither_ms"]
        extra["cfg3_total_note"] = ("BASELINE config 3 end to end on a resident image, warm processor: reference init at full resolution "
                                    "(its own colour histogram of the image included: cfg3_bind_ms is what a bind alone costs) + Lloyd to "
                                    "convergence with the final label map + dither output pass")
        s3.close()
        del labels, out
        extra.update(other_distributions(proc, k3, n_pixels, stream, steps=max(steps, 2) * 3))
        extra.update(cfg4_rank_share(proc, k3, n_pixels, stream, steps=steps))
        extra.update(cfg4_tiled_rank_share(k3, WIDTH, n_pixels // WIDTH, steps=steps))
        if n_pixels == WIDTH * ROWS_PER_GPU:
            extra.update(cfg4_native_batch(k3, WIDTH, n_pixels // WIDTH))
        extra.update(reduce_end_to_end(proc, rgba, WIDTH, n_pixels // WIDTH, k3))
        extra.update(cfg2_timing(proc, stream, steps=max(steps, 2) * 5, default_strategy=kg._default_strategy))
        extra.update(default_call_timing())
    except Exception as e:      # the extras must never break the benchmark line (tests/test_gpu_bench.py fails on it instead)
        import traceback
        extra["error"] = repr(e) + " | " + traceback.format_exc().strip().splitlines()[-3].strip()
    return extra


def k_of(sh):
    return int(sh.k)


def cpu_baseline(k, centroids4, seed, target_seconds=10.0, one_thread_seconds=5.0):
    """The CPU oracle (a port of the reference's WGSL; the reference itself needs Rust + Vulkan) on a bounded sample of the same
    workload: all host threads (`value`) and ONE thread (`one_thread`) -- the same pass, the same first rows."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from kmeans_gpu_amd import synth
    threads = O.num_threads()

    def measure(n_threads, seconds, probe_px):
        O.set_num_threads(n_threads)
        probe = synth.uniform_rgba_numpy(seed, probe_px)
        O.assign_accumulate_rgba(probe, centroids4)                 # (first call: the thread team starts)
        t = time.perf_counter()
        O.assign_accumulate_rgba(probe, centroids4)
        rate = probe_px / max(time.perf_counter() - t, 1e-4)
        n = int(min(WIDTH * ROWS_PER_GPU, max(probe_px, rate * seconds)))
        n -= n % WIDTH if n > WIDTH else 0
        px = synth.uniform_rgba_numpy(seed, n)
        t = time.perf_counter()
        O.assign_accumulate_rgba(px, centroids4)
        return n, time.perf_counter() - t
    try:
        n1, dt1 = measure(1, one_thread_seconds, 8192)
        n, dt = measure(threads, target_seconds, 64 * 8192 if threads > 8 else 8 * 8192)
    finally:
        O.set_num_threads(threads)
    what = "one assign+accumulate pass (per-pixel scan, literal CIE94 arg-min)"
    return {"value": n / dt, "unit": "pixels/s", "cores": threads, "kind": "port",
            "sample": f"first {n} pixels ({n // WIDTH} rows of {WIDTH}) of the same image, k={k}, {what}, {dt:.2f} s",
            "one_thread": {"value": n1 / dt1, "unit": "pixels/s", "cores": 1,
                           "sample": f"first {n1} pixels ({n1 // WIDTH} rows of {WIDTH}) of the same image, k={k}, {what}, {dt1:.2f} s"},
            "build": "oracle/Makefile: gcc -O3 -march=x86-64-v3 -ffp-contract=off -fno-fast-math -fopenmp"}


# speed-up of ONE 8192 x 8192, k = 256 image over N GPUs expected from this design (per-rank emulation on one
# GPU, tools/strong_cells_per_rank.py, profiles/r06w_strong_cells_per_rank.json: the worst rank's iteration, fused form; the all-gather of
# the label tables / the all-reduce are stand-ins there -- with a 2 MiB-per-peer all-gather over xGMI the cells estimate at N = 8 is ~2.8x)
EXPECTED_SPEEDUP = {"cells": {2: 1.68, 4: 2.64, 8: 3.66}, "bands": {2: 1.35, 4: 1.60, 8: 1.83}}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a FRESH child (torch.distributed.run, one process per
    GPU) before this process has made any HIP call, relay rank 0's JSON line, leave with the child's exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stderr.write(r.stdout)
    sys.exit(r.returncode if r.returncode else (0 if lines else 1))


class _StdoutToStderr:
    """RCCL prints a version banner on STDOUT when a communicator is created or first used: stdout carries the JSON line only"""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)


class _Loop:
    """what the extra measurements need of the timed loop: rank 0's kmg_lloyd, a k x 4 int64 buffer, the strategy"""

    def __init__(self, backend, acc, split_labels, k):
        self.backend, self.acc, self.split_labels, self.k = backend, acc, split_labels, int(k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--k", type=int, default=K)
    ap.add_argument("--rows", type=int, default=None, help="rows per GPU (default 8192; --scaling strong: 8192 / gpus)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="N > 1: strong (default) = the BASELINE 8192x8192 image split over the GPUs; weak = one 8192-row band "
                         "per GPU (reported under `extra` by the default run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extra measurements (use under rocprofv3 to keep kernel averages clean)")
    ap.add_argument("--overlap", action="store_true",
                    help="N > 1: the all-reduce of the sums on a second stream beside the label pass (KMG_GROUP_OVERLAP); default: "
                         "both ways are tried before the timed region and the faster one is measured")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: the all-reduce in line on the compute stream")
    ap.add_argument("--cells", action="store_true",
                    help="N > 1: shard the cube pass by cells of the colour cube as well (KMG_GROUP_CELLS: histogram all-reduce "
                         "once, all-gather of the label tables per iteration)")
    ap.add_argument("--no-cells", action="store_true", help="N > 1: row bands only")
    ap.add_argument("--separate-update", action="store_true",
                    help="centroid update as a launch of its own (k_update + memset) instead of on the assign pass's last launch")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on a box with ONE GPU: the N ranks are started as usual, but rank 0 hosts all N ranks of the "
                         "group on cuda:0 (KMG_GROUP_LOOPBACK: RCCL refuses two ranks on a device) -- exercises the launcher, the "
                         "N-rank loop of the library and this file's N > 1 code; its timings mean nothing")
    ap.add_argument("--force-dist", action="store_true",
                    help="load RCCL and run every collective even with one rank (KMG_GROUP_FORCE_COLLECTIVES)")
    ap.add_argument("--fail-first-candidate", action="store_true",
                    help="testing aid, N > 1: the first sharding candidate tried before the timed region raises, as a failed "
                         "collective would -- the ranks must drop the group, make a new one and measure another candidate")
    ap.add_argument("--only", choices=["cfg2"], default=None,
                    help="run one secondary configuration alone and print its JSON (for rocprofv3 legs): cfg2 = 4096x4096, k=16")
    ap.add_argument("--strategy", choices=["auto", "scan", "table"], default="auto",
                    help="per-pixel scan, colour table, or the library's cost model (default)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])                  # (never returns; nothing has touched the GPU yet)

    import numpy as np
    import torch
    import torch.distributed as dist
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if world > 1:
        # the processes' bootstrap (the communicator's unique id, barriers, the MAX of the ranks' times) goes over gloo on the
        # host; the data path's collectives are RCCL calls inside libkmeans_hip (kmg_group_*)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # one node: RCCL's bootstrap sockets (the unique id carries an address) stay on the loopback interface -- the container's
        # hostname may not resolve and no other interface is needed; the data path is xGMI / shared memory either way
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo")
    if args.rehearse and rank != 0:
        dist.barrier()                                        # rank 0 hosts every rank of the rehearsal
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path exists)")
    if world > torch.cuda.device_count() and not args.rehearse:
        # (every rank sees this before any communicator is created: nobody is left waiting in ncclCommInitRank)
        raise SystemExit(f"--gpus {world} needs {world} devices, this node shows {torch.cuda.device_count()} "
                         "(RCCL takes one rank per device; --rehearse runs the N-rank path on one GPU)")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    sync_ranks = world > 1 and not args.rehearse              # this process meets the others at barriers
    n_here = world if args.rehearse else 1                    # ranks of the group this process hosts
    first_rank = 0 if args.rehearse else rank

    k = args.k
    # N > 1 measures what BASELINE.json's north_star claims: the ONE 8192x8192 image over 1 / 2 / 4 / 8 GPUs (strong scaling);
    # the weak-scaling figure goes to `extra`
    args.scaling = args.scaling or ("strong" if world > 1 else "weak")
    if args.rows is not None:
        rows = args.rows
    elif args.scaling == "strong":
        rows = ROWS_PER_GPU // world
    else:
        rows = ROWS_PER_GPU
    n_local = WIDTH * rows
    height = rows * world
    seed = synth.SEED_CFG3
    kg.set_strategy(args.strategy)                            # (kmg_options.strategy of every processor this run creates)

    # ---- the group: ImageProcessor::new over this job's devices (include/kmeans_hip.h kmg_group_*) ----
    def make_group():
        with _StdoutToStderr():
            if args.rehearse:
                return kg.Group(devices=[0] * world, flags=kg.GROUP_LOOPBACK, shrink_max_dim=0)
            if world > 1 or args.force_dist:
                uid = [kg.Group.unique_id() if rank == 0 else None]
                if world > 1:
                    dist.broadcast_object_list(uid, src=0)
                return kg.Group(devices=[local_rank], unique_id=uid[0], first_rank=rank, world=world, shrink_max_dim=0,
                                flags=kg.GROUP_FORCE_COLLECTIVES if args.force_dist else 0)
            return kg.Group(devices=[local_rank], shrink_max_dim=0)
    group = make_group()
    proc = group.processor(0)
    collective_backend = ("loopback through device memory (rehearsal on one GPU)" if args.rehearse else
                          f"RCCL {group.rccl_version} (ncclAllReduce inside libkmeans_hip, dlopen'ed)" if group.rccl_version else
                          "none (one rank)")
    if args.only == "cfg2":
        st = torch.cuda.current_stream().cuda_stream
        which = ("table", "scan") if args.strategy == "auto" else (args.strategy,)
        res = cfg2_timing(proc, st, steps=args.steps, strategies=which, profile_kernels=not args.no_extras)
        group.close()
        print(json.dumps(res), flush=True)
        return
    bands = [synth.uniform_rgba_torch(seed, n_local, first=(first_rank + i) * n_local, device="cuda") for i in range(n_here)]
    label_maps = [torch.empty(n_local, dtype=torch.int32, device="cuda") for _ in range(n_here)]
    rgba, labels = bands[0], label_maps[0]

    # initial centroids: shader Lab of the pixels at linear index j * floor(N/k) of band 0
    n_first = WIDTH * ROWS_PER_GPU if (args.scaling == "strong" and args.rows is None) else n_local
    sel = synth.uniform_rgba_at(seed, np.arange(k, dtype=np.uint64) * np.uint64(n_first // k))
    d_sel = torch.from_numpy(sel).cuda()
    lab = torch.empty((k, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    proc.rgb_to_lab(d_sel.data_ptr(), k, lab.data_ptr(), stream)
    torch.cuda.synchronize()
    cent = np.ones((k, 4), np.float32)
    cent[:, :3] = lab.cpu().numpy()

    # one-time per-image preparation (the counterpart of the reference's one-time Lab conversion
    # pass, operations.rs:63-71): outside the per-iteration timing, reported separately -- on a cold processor (first
    # image: every block is a fresh hipMalloc, the static colour tables are built) and on a warm one (the second
    # image of a frame loop / of a rank's share of a batch: the first image's blocks are reused)
    def prepared():
        s = kg.Lloyd(proc, k)
        s.set_centroids(cent, stream)
        torch.cuda.synchronize()
        t = time.perf_counter()
        how = s.prepare(rgba.data_ptr(), n_local, True, stream)
        torch.cuda.synchronize()
        s.close()
        return how, time.perf_counter() - t
    strategy, t_prep_cold = prepared()
    _, t_prep = prepared()

    gl = kg.GroupLloyd(group, k)
    fused = world == 1 and not args.force_dist and not args.separate_update

    def bind_loop(gl_, flags, band_list, label_list, rows_, height_, centroids):
        gl_.bind([b.data_ptr() for b in band_list], [(first_rank + i) * rows_ for i in range(n_here)], [rows_] * n_here, WIDTH, height_,
                 [l.data_ptr() for l in label_list], flags)
        gl_.set_centroids(centroids)
        with _StdoutToStderr():
            gl_.prime()
            for _ in range(args.warmup):
                gl_.step()
            gl_.sync()

    def max_over_ranks(seconds):
        if not sync_ranks:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def fence(gl_):
        gl_.sync()
        if sync_ranks:
            dist.barrier()

    def timed(gl_, steps):
        fence(gl_)
        t = time.perf_counter()
        for _ in range(steps):
            gl_.step()
        # this rank's steps are done when its stream has drained; the job's time is the MAX over the ranks of these local times
        gl_.sync()
        return time.perf_counter() - t

    multi = world > 1
    cells_possible = multi and k <= 256 and (strategy == "table" or args.cells)
    picked = None
    if multi:
        # One image over N GPUs has two shapes -- row bands with a cell-sharded cube pass (the cube pass shrinks with N, the
        # label tables are all-gathered every iteration: 16 MiB / N per rank) or row bands alone (every rank labels the whole
        # cube for its band) -- and the k x 4 all-reduce can run in line or beside the label pass.  Which wins depends on the
        # node's fabric, which no run of this project has met yet: the candidates are timed here, outside the timed region (a
        # few iterations each, MAX over the ranks), and the fastest is the loop that is measured (config.sharding_choice).
        variants = {}
        if cells_possible and not args.no_cells and (args.scaling == "strong" or args.cells):
            # (the fused form: the cube pass adds into the accumulators, the band's label pass updates from the all-reduced sums --
            # two launches per iteration and rank instead of four)
            variants["cells"] = kg.GROUP_CELLS | kg.GROUP_FUSED_UPDATE
        if not args.cells:
            if not args.overlap:
                variants["bands"] = 0
            if not args.no_overlap:
                variants["bands, all-reduce beside the label pass"] = kg.GROUP_OVERLAP
        trial, failed = {}, {}
        for name, fl in variants.items():
            # (a candidate that raises on this rank counts as infinitely slow on all of them: the MAX over the ranks carries it, so
            # every rank drops it together and the next candidate is tried with the ranks still in step)
            try:
                bind_loop(gl, fl, bands, label_maps, rows, height, cent)
                if args.fail_first_candidate and not trial:
                    raise RuntimeError("--fail-first-candidate")
                t_local = timed(gl, 5)
            except Exception as e:                             # noqa: BLE001 -- reported below, not swallowed
                t_local = float("inf")
                failed[name] = f"{type(e).__name__}: {e}"
            trial[name] = max_over_ranks(t_local) / 5 * 1e3
            if trial[name] == float("inf"):
                # A rank whose collective or launch failed has marked its group broken and aborted its communicator
                # (include/kmeans_hip.h): every rank knows by now (the MAX above), so all of them drop the group together and make
                # a new one -- a new unique id over gloo, new communicators -- before the next candidate is tried.
                try:
                    gl.close()
                    group.close()
                except Exception:                              # noqa: BLE001 -- a broken group may refuse; it is dropped either way
                    pass
                group = make_group()
                proc = group.processor(0)
                gl = kg.GroupLloyd(group, k)
        best = min(trial, key=trial.get)                     # (the same on every rank: the times were all-reduced)
        if trial[best] == float("inf"):
            raise RuntimeError(f"no sharding of the image over {world} GPUs ran: {failed or 'another rank failed'}")
        flags = variants[best]
        if len(variants) > 1 or failed:
            picked = {"picked": best, **{f"{nm}_ms_per_step": (ms if ms != float("inf") else None) for nm, ms in trial.items()}}
            if failed:
                picked["failed_on_this_rank"] = failed
    else:
        flags = kg.GROUP_FUSED_UPDATE if fused else 0
    bind_loop(gl, flags, bands, label_maps, rows, height, cent)
    lloyd, strategy = gl.member(0)
    cells = bool(flags & kg.GROUP_CELLS)

    # HIP events around the heavy launches, on the launch stream.  Inside the timed region only the kernel the roofline is
    # quoted on carries events (every timed launch puts two event records = ~5 us of idle GPU between the kernels); the other
    # heavy kernel is timed in a short loop of its own right after the timed one.
    timed_kernels = ["k_assign", "k_labels"] if strategy == "table" else ["k_assign"]
    lloyd.profile(timed_kernels)
    elapsed = timed(gl, args.steps)
    fence(gl)
    prof = lloyd.profile_read()
    lloyd.profile(False)
    if strategy == "table":
        lloyd.profile(["k_cube"])
        for _ in range(min(args.steps, 10)):
            gl.step()
        gl.sync()
        prof.update(lloyd.profile_read())
        lloyd.profile(False)
    elapsed = max_over_ranks(elapsed)

    # N > 1: (a) the weak-scaling figure of the same loop (one 8192-row band per GPU, one k-means problem over the 8192 x 8192 N
    # image, row bands + the k x 4 all-reduce) -- same bracket, MAX over the ranks; (b) on rank 0 alone, the SAME 8192 x 8192
    # image on one GPU, so that the line carries the speed-up measured in this very run.  Both under `extra`.
    weak, one_gpu_ms = None, None
    if multi and args.scaling == "strong" and args.rows is None and not args.no_extras:
        n_w = WIDTH * ROWS_PER_GPU
        bands_w = [synth.uniform_rgba_torch(seed, n_w, first=(first_rank + i) * n_w, device="cuda") for i in range(n_here)]
        labels_w = [torch.empty(n_w, dtype=torch.int32, device="cuda") for _ in range(n_here)]
        sel_w = synth.uniform_rgba_at(seed, np.arange(k, dtype=np.uint64) * np.uint64(n_w // k))
        d_sel_w = torch.from_numpy(sel_w).cuda()
        proc.rgb_to_lab(d_sel_w.data_ptr(), k, lab.data_ptr(), stream)
        torch.cuda.synchronize()
        cent_w = np.ones((k, 4), np.float32)
        cent_w[:, :3] = lab.cpu().numpy()
        glw = kg.GroupLloyd(group, k)
        bind_loop(glw, flags & kg.GROUP_OVERLAP, bands_w, labels_w, ROWS_PER_GPU, ROWS_PER_GPU * world, cent_w)
        el_w = max_over_ranks(timed(glw, args.steps))
        weak = {"weak_scaling_value": n_w * world * args.steps / el_w, "weak_scaling_ms_per_step": el_w * 1e3 / args.steps,
                "weak_scaling_workload": f"{WIDTH}x{ROWS_PER_GPU * world} (one {ROWS_PER_GPU}-row band per GPU), row bands",
                "weak_scaling_strategy": glw.member(0)[1]}
        if rank == 0:
            # (the first band of the weak problem IS the BASELINE image)
            one = kg.Lloyd(proc, k)
            one.set_centroids(cent_w, stream)
            one.prepare(bands_w[0].data_ptr(), n_w, True, stream)
            acc1 = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            for i in range(args.warmup + args.steps):
                if i == args.warmup:
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                one.assign_update(bands_w[0].data_ptr(), n_w, labels_w[0].data_ptr(), acc1.data_ptr(), True, stream)
            torch.cuda.synchronize()
            one_gpu_ms = (time.perf_counter() - t1) / args.steps * 1e3
            one.close()
        glw.close()
        del bands_w, labels_w
        if sync_ranks:
            dist.barrier()

    # what a plain device-to-device copy of the same 2 x 4 B/px reaches on THIS box, for the "achievable" column
    copy_gbps = None
    if rank == 0:
        scratch = torch.empty_like(labels)
        for _ in range(2):
            scratch.copy_(labels)
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        for _ in range(5):
            scratch.copy_(labels)
        torch.cuda.synchronize()
        copy_gbps = 2.0 * labels.numel() * 4 * 5 / (time.perf_counter() - t_c) / 1e9
        del scratch

    if rank == 0:
        total_pixels = n_local * world
        ms_per_step = elapsed * 1e3 / args.steps
        kernels = {name: {"ms_per_launch": ms / cnt, "launches": cnt} for name, (ms, cnt) in prof.items()}
        dominant = max((nm for nm in prof if algorithmic_bytes(nm, n_local, k) is not None),
                       key=lambda name: prof[name][0])
        k_ms = kernels[dominant]["ms_per_launch"]
        abytes = algorithmic_bytes(dominant, n_local, k)
        achieved = abytes / (k_ms * 1e-3) / 1e9
        step_gbps = ALGORITHMIC_BYTES_PER_PIXEL * total_pixels / (ms_per_step * 1e-3) / 1e9
        traffic, traffic_source, tj = None, None, {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and k == K and rows == ROWS_PER_GPU:
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj.get("bytes_per_launch", {}).get(dominant)
            traffic_source = tj.get("source", {}).get(dominant, "profiles/traffic.json (rocprofv3 PMC passes, not this run)")
        flops = total_pixels * (FLOP_PER_PAIR * k + FLOP_PER_PIXEL)
        design_floor_ms = None
        if copy_gbps and "k_cube" in kernels and world == 1:
            design_floor_ms = ALGORITHMIC_BYTES_PER_PIXEL * total_pixels / (copy_gbps * 1e9) * 1e3 + kernels["k_cube"]["ms_per_launch"]
        scaling_note = {}
        if multi:
            shape = "cells" if cells else "bands"
            scaling_note = {"expected_speedup": EXPECTED_SPEEDUP[shape].get(world),
                            "expected_speedup_source": "DESIGN.md section 6: per-rank emulation on one GPU (tools/strong_cells_per_rank.py), "
                                                       "collectives not included; north_star asks for >= 6x at 8 GPUs, this design does "
                                                       "not expect it (at N = 8 a rank's iteration is 0.06 ms of which ~0.035 ms do not shrink with the share: "
                                                       "profiles/r06_share_kernels.csv)",
                            "measured_speedup": (one_gpu_ms / ms_per_step) if one_gpu_ms else None,
                            "one_gpu_ms_per_step_same_run": one_gpu_ms}
        out = {
            "metric": f"pixels/sec per Lloyd iteration ({WIDTH}x{height}, k={k})",
            "value": total_pixels * args.steps / elapsed,
            "unit": "pixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic uniform RGBA {WIDTH}x{height} (seed 0x5EED0003), "
                                   f"k={k}, one Lloyd iteration = update + assign (labels written every "
                                   f"iteration) + accumulate"
                                   + (" + RCCL all-reduce of k x 4 int64" if world > 1 else ""),
                       "width": WIDTH, "height": height, "k": k,
                       "sharding": f"row bands, {rows} rows per GPU" + (", cube pass sharded by cells" if cells else "")
                                   + (", all-reduce beside the label pass" if flags & kg.GROUP_OVERLAP else ""),
                       "driver": "kmg_group_lloyd_* of libkmeans_hip (C ABI): one rank per process, collectives inside the library",
                       "collective_backend": collective_backend, "collective_ranks": group.world,
                       "strategy": strategy, "prepare_ms": t_prep * 1e3, "prepare_cold_ms": t_prep_cold * 1e3,
                       "label_pass": "before the next iteration",
                       "update": ("by the band's label pass, from the all-reduced sums (kmg_lloyd_labels_from_tables_update)" if cells else
                                  "by the last launch of the assign pass (kmg_lloyd_assign_update)") if flags & kg.GROUP_FUSED_UPDATE
                                 else "k_update launch",
                       **scaling_note,
                       **({"sharding_choice": picked} if picked else {})},
            # BASELINE.json quotes "% HBM roofline" on the assign+update LOOP: `achieved` / `frac` are the whole iteration's (8 B/px over
            # ms_per_step: every launch, gap and collective); the dominant kernel's own figures are under kernel_*
            "roofline": {"bound": "hbm",
                         "scope": "the whole Lloyd iteration (all launches of assign + update): 8 B/px x pixels / ms_per_step",
                         "achieved": step_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": step_gbps / HBM_PEAK_GBPS,
                         "peak_achievable": HBM_ACHIEVABLE_GBPS, "frac_achievable": step_gbps / HBM_ACHIEVABLE_GBPS,
                         "copy_measured": copy_gbps,      # torch copy_ of the label map on this box, read + write, GB/s
                         "frac_of_copy": (step_gbps / copy_gbps) if copy_gbps else None,
                         "algorithmic_bytes_per_step": ALGORITHMIC_BYTES_PER_PIXEL * total_pixels,
                         # the floor of THIS design: the label pass cannot beat a device copy of the same
                         # 8 B/px, and the cube pass moves none of them -- copy (measured on this box, above) + k_cube (measured,
                         # below).  What the step takes beyond it is gather requests, launch boundaries and the tail.
                         "design_floor_ms": design_floor_ms,
                         "design_floor_source": "8 B/px / roofline.copy_measured + kernels.k_cube.ms_per_launch, both measured in this run",
                         "design_floor_frac": (ALGORITHMIC_BYTES_PER_PIXEL * total_pixels / (design_floor_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS)
                                              if design_floor_ms else None,
                         "kernel": dominant,
                         "kernel_bound": "requests" if dominant == "k_labels" else ("valu" if dominant == "k_assign" else "hbm"),
                         "kernel_bound_note": "k_labels: a device copy plus divergent gather requests that the vector memory pipeline retires "
                                              "at ~1 lane per 2 clocks per CU, hit or miss (profiles/NOTES.md); the fraction is still taken "
                                              "against the HBM roof, the metric BASELINE.json names",
                         "kernel_ms": k_ms, "kernel_algorithmic_bytes_per_launch": abytes,
                         "kernel_achieved": achieved, "kernel_frac": achieved / HBM_PEAK_GBPS,
                         "kernel_frac_achievable": achieved / HBM_ACHIEVABLE_GBPS,
                         "traffic": traffic, "traffic_source": traffic_source, "traffic_age": tj.get("commit"),
                         "traffic_note": "counter bytes of the dominant kernel per launch (its algorithmic bytes: kernel_algorithmic_bytes_per_launch)",
                         # SURVEY 8d's second roof, for orientation only: the time the LITERAL per-pixel scan's flops
                         # (17 k + 50 per pixel) would need at the fp32 vector peak, against the measured step -- a ratio, not a
                         # utilisation (the colour table does not execute those flops).  What the kernels really issue is in
                         # kernels_roofline.*.valu
                         "literal_scan_valu_floor_ms": flops / (FP32_VECTOR_PEAK_TFLOPS * 1e12) * 1e3,
                         "literal_scan_valu_floor_over_step": flops / (ms_per_step * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
                         "valu_peak_tflops": FP32_VECTOR_PEAK_TFLOPS},
            "kernels": kernels,
            # every timed kernel against the HBM roof, from its own algorithmic bytes (k_cube: 2^24 counts in, 2^24 labels out)
            "kernels_roofline": {nm: {"algorithmic_bytes_per_launch": algorithmic_bytes(nm, n_local, k),
                                      "achieved_GBps": algorithmic_bytes(nm, n_local, k) / (v["ms_per_launch"] * 1e-3) / 1e9,
                                      "frac": algorithmic_bytes(nm, n_local, k) / (v["ms_per_launch"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                      "traffic": (tj.get("bytes_per_launch", {}).get(nm) if traffic is not None else None)}
                                 for nm, v in kernels.items() if algorithmic_bytes(nm, n_local, k) is not None},
            "kernels_note": "HIP events on the launch stream; the dominant kernel inside the timed region, k_cube in a loop of "
                            "its own right after it (same state, same launches)",
        }
        attach_valu_roof(out["kernels_roofline"], {nm: v["ms_per_launch"] for nm, v in kernels.items()}, tj)
        if world == 1 and rows == ROWS_PER_GPU and not args.no_extras:
            acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
            out["extra"] = output_pass_timing(proc, rgba, n_local, stream, _Loop(lloyd, acc, strategy == "table", k), prep_ms=t_prep * 1e3)
            for name in ("find_dither_k64", "find_replace_k64"):          # the output passes of BASELINE config 5, same roof
                if name + "_ms" in out["extra"]:
                    ms = out["extra"][name + "_ms"]
                    out["kernels_roofline"][name] = {
                        "algorithmic_bytes_per_launch": ALGORITHMIC_BYTES_PER_PIXEL * n_local,
                        "achieved_GBps": ALGORITHMIC_BYTES_PER_PIXEL * n_local / (ms * 1e-3) / 1e9,
                        "frac": ALGORITHMIC_BYTES_PER_PIXEL * n_local / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "traffic": tj.get("bytes_per_launch", {}).get(name),
                        "note": "whole kmg_dev_apply call (all its launches), host clock over 3 calls"}
            c2 = cfg2_roofline(out["extra"], tj)
            if c2 is not None:
                out["kernels_roofline"]["cfg2"] = c2
            attach_valu_roof(out["kernels_roofline"], {nm[:-3]: ms for nm, ms in out["extra"].items() if nm.endswith("_ms")}, tj)
        if weak is not None:
            out.setdefault("extra", {}).update(weak)
        if not args.no_cpu_baseline:                              # (rank 0, beside every N: north_star)
            out["cpu_baseline"] = cpu_baseline(k, cent, seed)
        line = json.dumps(out)
    else:
        line = None

    gl.close()
    group.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        # RCCL writes its version banner to the C stdout buffer; flush it first so that the JSON
        # line is the last thing on stdout
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(line, flush=True)


if __name__ == "__main__":
    main()
