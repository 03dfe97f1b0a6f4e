import numpy as np, torch, sys
sys.path.insert(0,'.')
from meshflow_amd import synthetic, ops
from oracle import clib, meshflow_oracle as mo
H,W,R,C=64,96,4,4
frames,disp,hom=synthetic.clip(5,H,W,R,C,seed=H+W,kind='noise',jitter_sigma=1.0)
stab=mo.stabilized_vertex_displacements(W,H,0,disp,hom,3,10)
dev=torch.device('cuda:0')
t=ops.cell_table(torch.from_numpy(disp).to(dev),torch.from_numpy(stab).to(dev),W,H,R,C)
out=ops.warp(torch.from_numpy(frames).to(dev),t).cpu().numpy()
for f in range(5):
    tab,_=clib.cell_table(W,H,R,C,disp[f],stab[f])
    want,wc=clib.warp_frame(frames[f],R,C,tab)
    d=np.argwhere((out[f]!=want).any(axis=2))
    print(f,len(d),d[:12].tolist())
    if len(d):
        y,x=d[0]; print(out[f][y,x],want[y,x])
