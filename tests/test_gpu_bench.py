"""bench.py's extra measurements at reduced size (-m gpu): bench.py swallows an exception of the extras into
`extra.error` so that the benchmark line always appears -- here such an exception fails the suite instead."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_bench_extras_run_without_error(torch_cuda, oracle):
    import bench
    import kmeans_gpu_amd as kg
    from kmeans_gpu_amd import synth
    torch = torch_cuda
    st = torch.cuda.current_stream().cuda_stream
    rows, k = 512, 256
    n = bench.WIDTH * rows
    proc = kg.ImageProcessor(shrink_max_dim=0)
    rgba = synth.uniform_rgba_torch(synth.SEED_CFG3, n, device="cuda")
    labels = torch.empty(n, dtype=torch.int32, device="cuda")
    sel = synth.uniform_rgba_at(synth.SEED_CFG3, np.arange(k, dtype=np.uint64) * np.uint64(n // k))
    cent = oracle.centroids4(oracle.rgb_to_lab(sel))
    lloyd = kg.Lloyd(proc, k)
    lloyd.set_centroids(cent, st)
    strategy = lloyd.prepare(rgba.data_ptr(), n, True, st)
    acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
    lloyd.assign_update(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), True, st)
    lloyd.assign_update(rgba.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), True, st)
    extra = bench.output_pass_timing(proc, rgba, n, st, bench._Loop(lloyd, acc, strategy == "table", k), steps=1)
    assert "error" not in extra, extra["error"]
    for key in ("find_dither_k64_ms", "find_replace_k64_ms", "iteration_without_label_map_ms", "cfg3_init_ms",
                "cfg3_lloyd_and_labels_ms", "cfg3_dither_ms", "blobs_ms_per_step", "photo_ms_per_step",
                "cfg4_rank_share_ms_per_iteration", "cfg4_tiled_rank_share_ms_per_iteration", "reduce_host_to_host_warm_ms",
                "cfg2_ms_per_step", "cfg1_reduce_tokyo_k8_replace_ms", "default_palette_tokyo_k256_ms"):
        assert key in extra and extra[key] > 0, key
    lloyd.close()
    proc.close()
