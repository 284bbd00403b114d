import sys
sys.path.insert(0, "kmeans-gpu_amd/python"); sys.path.insert(0, "tests")
import numpy as np, torch
import kmeans_gpu_amd as kg, oracle_lib as O
p = kg.ImageProcessor(shrink_max_dim=0)
st = torch.cuda.current_stream().cuda_stream
n = 300_000
rgba = O.synth_uniform(5, n)
d = torch.from_numpy(rgba).cuda()
for k in (512, 1024, 1365, 2048, 3000, 3392, 4096):
    cent = O.centroids4(O.rgb_to_lab(rgba[:k]))
    wl, wa = O.assign_accumulate_rgba(rgba, cent)
    for bind in (False, True):
        try:
            s = kg.Lloyd(p, k); s.set_centroids(cent, st)
            if bind: s.bind_image(d.data_ptr(), n, st)
            labels = torch.zeros(n, dtype=torch.int32, device="cuda"); acc = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
            s.assign_accumulate(d.data_ptr(), n, labels.data_ptr(), acc.data_ptr(), st); torch.cuda.synchronize()
            ok = np.array_equal(labels.cpu().numpy().view(np.uint32), wl) and np.array_equal(acc.cpu().numpy(), wa)
            print(k, "table" if bind else "scan", "OK" if ok else "MISMATCH"); s.close()
        except Exception as e:
            print(k, "table" if bind else "scan", "ERR", str(e)[:120])
    try:
        out = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
        p.apply(d.data_ptr(), 1000, 300, 0, cent, 1, out.data_ptr(), st); torch.cuda.synchronize()
        print(k, "apply dither ok")
    except Exception as e:
        print(k, "apply ERR", str(e)[:120])
