import numpy as np, sys
sys.path.insert(0,'.')
import implicit_amd.gpu as gpu
f=256; rng=np.random.default_rng(7)
Yh=rng.random((26744,f),dtype=np.float32)*0.01
Y=gpu.Matrix(Yh); norms=gpu.calculate_norms(Y); knn=gpu.KnnQuery()
ids,d=knn.topk(Y, Y[0:1000], 100, item_norms=norms)
nh=norms.to_numpy().reshape(-1)
for r in (671, 345, 10):
    s=(Yh@Yh[r])/nh
    o=np.argsort(-s, kind='stable')
    print('row',r,'scores 98..102', s[o[97:103]], 'gaps', np.diff(s[o[97:103]]), 'gpu ids tail', ids[r,97:100], 'np', o[97:100], file=sys.stderr)
    # tile maxima: count tiles whose max >= 100th largest tile max
    tm=np.array([s[i:i+64].max() for i in range(0,26744,64)])
    tau=np.sort(tm)[-100]
    print('  tiles>=tau', (tm>=tau).sum(), 'scores>=tau', (s>=tau).sum(), 'k-th score', s[o[99]], 'tau', tau, file=sys.stderr)
