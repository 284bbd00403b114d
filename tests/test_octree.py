"""Algorithm::Octree (core/src/octree.rs): the product's host implementation (C++, ordered set) against
the oracle's independent restatement (plain C, linear scans), the reference's own unit test
(octree.rs:250-311), and -- on the GPU box -- palette / reduce end to end."""
import numpy as np
import pytest

from conftest import sorted_palette


def test_reference_unit_test_46_colours_to_8(oracle):
    """octree.rs:250-311 test_add_color: the 46 apollo colours reduce to exactly 8"""
    import kmeans_gpu_amd as kg
    px = sorted_palette("apollo-1x.png")
    assert len(px) == 46 and tuple(px[0]) == (9, 10, 20, 255) and tuple(px[-1]) == (235, 237, 233, 255)
    got = kg.octree_palette(px, 8)
    assert len(got) == 8
    assert np.array_equal(got, oracle.octree_palette(px, 8))


@pytest.mark.parametrize("n,k", [(1, 1), (1, 5), (7, 3), (300, 1), (300, 2), (5000, 16), (5000, 64), (16384, 256), (2000, 4000)])
def test_product_octree_equals_oracle(oracle, n, k):
    import kmeans_gpu_amd as kg
    rng = np.random.default_rng(n * 31 + k)
    centres = rng.integers(0, 256, (12, 3))
    px = np.full((n, 4), 255, np.uint8)
    px[:, :3] = np.clip(centres[rng.integers(0, 12, n)] + rng.normal(0, 9, (n, 3)), 0, 255).astype(np.uint8)
    got, want = kg.octree_palette(px, k), oracle.octree_palette(px, k)
    assert np.array_equal(got, want)
    assert 1 <= len(got) <= k
    assert [tuple(c) for c in got] == sorted(set(tuple(c) for c in got))      # sorted, deduplicated
    flat = oracle.synth_uniform(n + k, n)
    assert np.array_equal(kg.octree_palette(flat, k), oracle.octree_palette(flat, k))


def test_octree_few_distinct_colours(oracle):
    import kmeans_gpu_amd as kg
    px = np.tile(np.array([[10, 20, 30, 255], [10, 20, 31, 255], [200, 100, 0, 255]], np.uint8), (50, 1))
    for k in (1, 2, 3, 10):
        got = kg.octree_palette(px, k)
        assert np.array_equal(got, oracle.octree_palette(px, k))
    assert len(kg.octree_palette(px, 10)) == 3


@pytest.mark.gpu
@pytest.mark.parametrize("k", [2, 8, 16])
def test_palette_and_reduce_with_octree(processor, oracle, tokyo, k):
    import kmeans_gpu_amd as kg
    got = processor.palette(k, tokyo, kg.Algorithm.Octree)
    want = oracle.palette_octree(tokyo, k)
    assert np.array_equal(got, want)
    for mode in (0, 1):
        assert np.array_equal(processor.reduce(k, tokyo, kg.Algorithm.Octree, mode), oracle.reduce_octree(tokyo, k, mode))


@pytest.mark.gpu
def test_octree_small_image_is_not_resized(processor, oracle):
    img = oracle.synth_uniform(3, 100 * 128).reshape(100, 128, 4)        # <= 128: used as is (lib.rs:295)
    import kmeans_gpu_amd as kg
    assert np.array_equal(processor.palette(6, img, kg.Algorithm.Octree), oracle.palette_octree(img, 6))
