#!/usr/bin/env python3
"""What would richer LDS entries of the label pass resolve?  (run on the GPU box: python tools/pair_schemes.py > gpurun_out/pair_schemes.txt)

The label of every colour of the cube (the per-pixel scan over a 4096 x 4096 image holding all 2^24 colours) after 12 Lloyd
iterations of the benchmark workload (and of the tiled photograph), analysed per 8x8x8 cell with torch:
  * cells by number of labels, and the share of the PIXELS in them;
  * two-label cells: share of their pixels inside the slab for the direction the cube pass picks today (centre of mass at
    half-cell resolution, components -2..2), for the best of the 125 directions, and for the best direction with components -3..3;
  * cells with three or more labels: what a second entry (the most frequent label split off by the first plane, the next two
    separated by a second plane) would leave to the per-colour gather.
Weights: the image's own colour histogram, so every figure is a share of the pixels of that image."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kmeans-gpu_amd", "python"))
sys.path.insert(0, ROOT)
import numpy as np, torch
import kmeans_gpu_amd as kg
from kmeans_gpu_amd import synth
import bench

dev = "cuda"
proc = kg.ImageProcessor(shrink_max_dim=0)
st = torch.cuda.current_stream().cuda_stream
n = 8192 * 8192
k = int(sys.argv[1]) if len(sys.argv) > 1 else 256

# all 2^24 colours as an image: pixel i = colour i (r | g << 8 | b << 16)
idx = torch.arange(1 << 24, device=dev, dtype=torch.int32)
cube = (idx | (255 << 24)).view(torch.uint8).reshape(-1, 4).contiguous()


def directions(m):
    r = torch.arange(-m, m + 1, device=dev)
    d = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), -1).reshape(-1, 3)
    return d[(d != 0).any(1)]


xyz = torch.stack(torch.meshgrid(*(torch.arange(8, device=dev),) * 3, indexing="ij"), -1).reshape(512, 3)   # (r&7, g&7, b&7)


def slab_share(lab_cells, w_cells, a, b, dirs, chunk=512):
    """two-label cells: for every cell the smallest weighted share of colours in [tlo, thi] over `dirs`.
    lab_cells [C, 512] labels, w_cells [C, 512] weights, a / b [C] the two labels.  Returns [C] best slab weight."""
    best = torch.full((lab_cells.shape[0],), float("inf"), device=dev)
    p_all = (xyz.float() @ dirs.float().T).T.contiguous()            # [D, 512]
    occ = w_cells > 0
    for c0 in range(0, lab_cells.shape[0], chunk):
        L = lab_cells[c0:c0 + chunk]; W = w_cells[c0:c0 + chunk]; O = occ[c0:c0 + chunk]
        A = a[c0:c0 + chunk, None]; B = b[c0:c0 + chunk, None]
        notA = (L != A) & O; notB = (L != B) & O
        P = p_all[None]                                              # [1, D, 512]
        big = 1e9
        tlo = torch.where(notA[:, None, :], P, torch.full_like(P, big)).amin(-1)           # lowest p of a colour that is not A
        thi = torch.where(notB[:, None, :], P, torch.full_like(P, -big)).amax(-1)          # highest p of a colour that is not B
        inslab = (P >= tlo[..., None]) & (P <= thi[..., None]) & O[:, None, :]
        s = (inslab * W[:, None, :]).sum(-1)                         # [c, D]
        best[c0:c0 + chunk] = s.amin(1)
    return best


def analyse(name, rgba):
    s = kg.Lloyd(proc, k)
    sel = rgba[(torch.arange(k, device=dev) * (n // k))].contiguous()
    lab = torch.empty((k, 3), dtype=torch.float32, device=dev)
    proc.rgb_to_lab(sel.data_ptr(), k, lab.data_ptr(), st)
    torch.cuda.synchronize()
    cent = np.ones((k, 4), np.float32); cent[:, :3] = lab.cpu().numpy()
    s.set_centroids(cent, st)
    s.prepare(rgba.data_ptr(), n, True, st)
    acc = torch.zeros((k, 4), dtype=torch.int64, device=dev)
    for it in range(12):
        s.assign_update(rgba.data_ptr(), n, 0, acc.data_ptr(), True, st)
    s.assign_accumulate(rgba.data_ptr(), n, 0, acc.data_ptr(), st)
    bad, resolved, total = s.debug_check_pairs(st)
    cent = s.get_centroids(st)
    s.close()
    # labels of all colours under these centroids (per-pixel scan of the cube image)
    kg.set_strategy("scan")
    b = kg.Lloyd(proc, k); b.set_centroids(cent, st)
    labels = torch.zeros(1 << 24, dtype=torch.int32, device=dev)
    b.assign_accumulate(cube.data_ptr(), 1 << 24, labels.data_ptr(), 0, st)
    torch.cuda.synchronize(); b.close()
    kg.set_strategy("auto")
    # histogram of the image over r | g << 8 | b << 16
    col = (rgba.view(torch.int32).reshape(-1) & 0xFFFFFF).long()
    hist = torch.bincount(col, minlength=1 << 24).float()
    # [b, g, r] -> cells [32,32,32] x [8,8,8] with inner order (r&7, g&7, b&7) to match xyz
    def cells(t):
        t = t.reshape(32, 8, 32, 8, 32, 8)                           # b_hi b_lo g_hi g_lo r_hi r_lo
        return t.permute(4, 2, 0, 5, 3, 1).reshape(32768, 512)       # cell (r,g,b hi), colour (r,g,b lo)
    L = cells(labels); W = cells(hist)
    tot = W.sum()
    occ = W > 0
    Ls = torch.where(occ, L, torch.full_like(L, -1))
    srt = Ls.sort(1).values
    new = (srt[:, 1:] != srt[:, :-1]) & (srt[:, 1:] >= 0)
    nlab = new.sum(1) + (srt[:, 0] >= 0).int()
    print(f"== {name}: k={k}, library: {1 - resolved / total:.4f} of the pixels gather today (mismatches {bad})")
    for c in (0, 1, 2, 3, 4):
        m = nlab == c if c < 4 else nlab >= 4
        print(f"   cells with {c}{'+' if c == 4 else ''} labels: {int(m.sum()):6d}  pixels {float(W[m].sum() / tot):.4f}")
    # most frequent labels per cell (weighted by pixels)
    onehot_w = torch.zeros((32768, k), device=dev)
    onehot_w.scatter_add_(1, L.long(), W)
    top = onehot_w.topk(3, dim=1)
    two = nlab == 2
    if two.any():
        a, b_ = top.indices[two, 0].int(), top.indices[two, 1].int()
        w2 = float(W[two].sum())
        for nm, dirs in (("125 directions (-2..2), best", directions(2)), ("components -3..3, best", directions(3))):
            sl = slab_share(L[two], W[two], a, b_, dirs)
            print(f"   two-label cells, {nm}: slab = {float(sl.sum()) / w2:.4f} of their pixels = {float(sl.sum() / tot):.4f} of all")
    three = nlab >= 3
    if three.any():
        w3 = float(W[three].sum())
        Lc, Wc = L[three], W[three]
        A = top.indices[three, 0].int(); B = top.indices[three, 1].int(); Cc = top.indices[three, 2].int()
        # first plane: A against the rest, only the A side resolved (today's w = 7 entries): colours with p < tlo
        dirs = directions(2)
        p_all = (xyz.float() @ dirs.float().T).T.contiguous()
        res1 = torch.zeros(Lc.shape[0], device=dev); res2 = torch.zeros(Lc.shape[0], device=dev)
        for c0 in range(0, Lc.shape[0], 256):
            l = Lc[c0:c0 + 256]; w = Wc[c0:c0 + 256]; o = w > 0
            a = A[c0:c0 + 256, None]; b2 = B[c0:c0 + 256, None]; c2 = Cc[c0:c0 + 256, None]
            P = p_all[None]
            notA = (l != a) & o
            tlo = torch.where(notA[:, None, :], P, torch.full_like(P, 1e9)).amin(-1)
            sideA = (P < tlo[..., None]) & o[:, None, :]
            r1 = (sideA * w[:, None, :]).sum(-1)                      # resolved by the A side, per direction
            res1[c0:c0 + 256] = r1.amax(1)
            # second plane over the colours NOT on the A side of the best first plane: B below, C above
            bestd = r1.argmax(1)
            rest = o & ~sideA[torch.arange(l.shape[0]), bestd]
            notB = (l != b2) & rest; notC = (l != c2) & rest
            t_lo = torch.where(notB[:, None, :], P, torch.full_like(P, 1e9)).amin(-1)
            t_hi = torch.where(notC[:, None, :], P, torch.full_like(P, -1e9)).amax(-1)
            sideB = (P < t_lo[..., None]) & rest[:, None, :]
            sideC = (P > t_hi[..., None]) & rest[:, None, :]
            r2 = ((sideB | sideC) * w[:, None, :]).sum(-1)
            res2[c0:c0 + 256] = r2.amax(1)
        # ONE plane, both sides: p < tlo -> A, p > thi -> B (thi = highest p of a colour that is not B), for the 125 directions
        # and for the 18 directions with components -1..1 and at most two of them non-zero (p <= 14: two 4-bit thresholds)
        for nm, dirs in (("125 directions", directions(2)),
                         ("18 directions (4-bit thresholds)", directions(1)[(directions(1) != 0).sum(1) <= 2])):
            p_all = (xyz.float() @ dirs.float().T).T.contiguous()
            p_all = p_all - p_all.amin(1, keepdim=True)
            both = torch.zeros(Lc.shape[0], device=dev)
            for c0 in range(0, Lc.shape[0], 256):
                l = Lc[c0:c0 + 256]; w = Wc[c0:c0 + 256]; o = w > 0
                a = A[c0:c0 + 256, None]; b2 = B[c0:c0 + 256, None]
                P = p_all[None]
                tlo = torch.where(((l != a) & o)[:, None, :], P, torch.full_like(P, 1e9)).amin(-1)
                thi = torch.where(((l != b2) & o)[:, None, :], P, torch.full_like(P, -1e9)).amax(-1)
                res = (((P < tlo[..., None]) | (P > thi[..., None])) & o[:, None, :]) * w[:, None, :]
                both[c0:c0 + 256] = res.sum(-1).amax(1)
            print(f"   cells with 3+ labels, ONE plane with an A side and a B side, {nm}: resolves {float(both.sum()) / w3:.4f} of their pixels; "
                  f"left to the gather {float((w3 - both.sum()) / tot):.4f} of all pixels")
        print(f"   cells with 3+ labels: A side of the best first plane resolves {float(res1.sum()) / w3:.4f} of their pixels; "
              f"a second plane (B | C) {float(res2.sum()) / w3:.4f} more; left to the gather {1 - float((res1 + res2).sum()) / w3:.4f} "
              f"= {float((w3 - (res1 + res2).sum()) / tot):.4f} of all pixels (today's entries leave {float((w3 - res1.sum()) / tot):.4f} at best)")


analyse("uniform", synth.uniform_rgba_torch(synth.SEED_CFG3, n, device=dev))
analyse("photo", bench.synthetic_image("photo", n, 0, k, 0x5EED0B10))
