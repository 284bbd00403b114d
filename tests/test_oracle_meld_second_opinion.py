"""The reference commits no golden image of `-m meld` (samples.sh:3-8), so the oracle's restatement of mix_colors.wgsl main_meld
is pinned by no reference fixture.  A second, INDEPENDENT transcription -- written here straight from the WGSL text in scalar
binary32 numpy arithmetic, with its own distance_cie94 (functions/delta_e.wgsl:1-22) and the shader's control flow exactly as
written (the distances to `closest` / `second_closest` are RECOMPUTED at every comparison, mix_colors.wgsl:37-43) -- must agree
with the C oracle bit for bit.  Two restatements by different routes agreeing is not the reference's vector, but it removes
transcription slips (argument order of the asymmetric CIE94, the sentinel, strict `<`, the blend factor's operands)."""
import numpy as np
import pytest

f32 = np.float32


def _cie94(one, second):
    """functions/delta_e.wgsl:1-22, operation by operation in binary32"""
    K1, K2 = f32(0.045), f32(0.015)
    dL = one[0] - second[0]
    da = one[1] - second[1]
    db = one[2] - second[2]
    C1 = np.sqrt(one[1] * one[1] + one[2] * one[2])
    C2 = np.sqrt(second[1] * second[1] + second[2] * second[2])
    dC = C1 - C2
    dH = np.sqrt(np.maximum((da * da) + (db * db) - (dC * dC), f32(0.0)))
    SL = f32(1.0)
    SC = f32(1.0) + K1 * C1
    SH = f32(1.0) + K2 * C1
    return np.sqrt((dL / SL) * (dL / SL) + (dC / SC) * (dC / SC) + (dH / SH) * (dH / SH))


def _meld_pixel(color, cents):
    """mix_colors.wgsl:29-48 two_closest_colors + :85-90 meld (main_meld :127-131 handles count == 1 before)"""
    closest = np.full(3, 10000.0, f32)
    second = np.full(3, 10000.0, f32)
    for temp in cents:
        d = _cie94(color, temp)
        if d < _cie94(color, closest):
            second = closest
            closest = temp
        elif d < _cie94(color, second):
            second = temp
    factor = _cie94(color, second) / _cie94(closest, second)
    return factor * closest + (f32(1.0) - factor) * second


@pytest.mark.parametrize("seed,k", [(1, 2), (2, 3), (3, 7), (4, 16), (5, 40)])
def test_oracle_meld_equals_an_independent_transcription_of_the_shader(oracle, seed, k):
    rng = np.random.default_rng(seed)
    w, h = 23, 9
    rgba = rng.integers(0, 256, (w * h, 4), dtype=np.uint8)
    if seed == 3:
        rgba[:40] = rgba[0]                                   # repeated colours
    lab = oracle.rgb_to_lab(rgba)
    pal = rng.integers(0, 256, (k, 4), dtype=np.uint8)
    if seed == 4:
        pal[3] = pal[2]                                       # a duplicate centroid: distance 0 between the two closest
        rgba[5, :3] = pal[2, :3]
        lab = oracle.rgb_to_lab(rgba)
    cent = oracle.centroids4(oracle.rgb_to_lab(pal))          # shader Lab of the palette: any centroid table will do
    got = oracle.meld(lab, w, h, cent)
    with np.errstate(divide="ignore", invalid="ignore"):
        want = np.stack([_meld_pixel(lab[i].astype(f32), [c[:3].astype(f32) for c in cent]) for i in range(w * h)])
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), f"{int((~same).any(1).sum())} of {w * h} pixels differ"


def test_oracle_meld_single_colour(oracle):
    """main_meld with count == 1: the centroid itself (mix_colors.wgsl:127-131)"""
    lab = oracle.rgb_to_lab(np.array([[10, 200, 30, 255], [0, 0, 0, 255]], np.uint8))
    cent = oracle.centroids4(np.array([[50.0, 10.0, -20.0]], np.float32))
    assert np.array_equal(oracle.meld(lab, 2, 1, cent), np.repeat(cent[:, :3], 2, 0))
