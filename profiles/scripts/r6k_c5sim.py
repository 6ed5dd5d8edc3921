import numpy as np, time, sys
sys.path.insert(0,'.')
import implicit_amd.gpu as gpu
f=256; rng=np.random.default_rng(7)
Y=gpu.Matrix(rng.random((26744,f),dtype=np.float32)*0.01)
norms=gpu.calculate_norms(Y); knn=gpu.KnnQuery()
for trial in range(2):
    ids,d=knn.topk(Y, Y[0:1000], 100, item_norms=norms)
Yt=gpu.Matrix((rng.standard_normal((26744,f))*0.1).astype(np.float32))
n2=gpu.calculate_norms(Yt)
print("normal factors", file=sys.stderr)
ids,d=knn.topk(Yt, Yt[0:1000], 100, item_norms=n2)
