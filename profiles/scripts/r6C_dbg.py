import sys, numpy as np
sys.path.insert(0,'.')
import implicit_amd.gpu as gpu
rng=np.random.default_rng(3)
ni,f,k=26744,256,100
Y=(rng.random((ni,f),dtype=np.float32)*0.01) if len(sys.argv)>1 else (rng.standard_normal((ni,f))*0.1).astype(np.float32)
n=np.linalg.norm(Y,axis=1).astype(np.float32)
knn=gpu.KnnQuery()
I=gpu.Matrix(Y); N=gpu.Matrix(n.reshape(1,-1))
Q=Y[:1000]
ids,d=knn.topk(I,gpu.Matrix(Q),k,item_norms=N)
s=(Q[:4]@Y.T)/n[None,:]
srt=np.sort(s,axis=1)[:,::-1]
print("true 100th score rows0-3", srt[:,99], "std of row 0", s[0].std(), "fraction above its tau guess", (s[0]>np.sort(s[0])[::-1][99]-3*s[0].std()).mean())
sub=s[:, np.concatenate([np.arange(b*128,min(ni,b*128+128)) for b in range(0,(ni+127)//128,4)])]
print("subset size", sub.shape, "100th of subset", np.sort(sub,axis=1)[:,::-1][:,99])
gm=[np.sort(np.array([sub[0,t::256].max() for t in range(256)]))[::-1][99]]
print("100th largest of 256 strided group maxima row0", gm)
print("ids ok", (ids[:4,:5]))
