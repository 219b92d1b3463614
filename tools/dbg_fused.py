import importlib, sys, numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("openmp-parallel-computing_amd")
from oracle import pyoracle as O
api = pkg.hostapi; api.init(0)
for (w,h) in ((1024,40),(2048,70),(64,5)):
    img = O.frame_rand(w,h,3,seed=3)
    d = torch.from_numpy(img).cuda()
    want = O.sobel(O.gray_to_plane(img,0))
    got = api.gray_sobel_fused(d,0).cpu().numpy()
    bad = np.argwhere(got != want)
    print(w,h,'mismatches',len(bad))
    if len(bad):
        ys = np.unique(bad[:,0]); xs = np.unique(bad[:,1] % 16)
        print(' rows', ys[:20], ' x%16', xs, ' x range', bad[:,1].min(), bad[:,1].max())
        for (y,x) in bad[:6]: print('  ', y, x, got[y,x], want[y,x])
