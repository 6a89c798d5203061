import numpy as np, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from upliftingtabletennis_amd import calib, synth
g=np.load('/root/repo/tests/golden/calib64.npz')
n=64
kps=np.stack([g['calib64/%d/keypoints'%ci] for ci in range(n)])
mint,mext,ninl=calib.calibrate_cameras(kps)
T=synth.TABLE_POINTS
def obj(kp,Mi,Me):
    vis=kp[:,2]==1
    return np.linalg.norm(calib.reproject(T[vis],Mi,Me)-kp[vis,:2],axis=1)
rows=[]
for ci in range(n):
    rMi,rMe=g['calib64/%d/Mint'%ci],g['calib64/%d/Mext'%ci]
    ed,er=obj(kps[ci],mint[ci],mext[ci]),obj(kps[ci],rMi,rMe)
    idv,irf=ed<3.5,er<3.5
    d=np.linalg.norm(calib.reproject(T,mint[ci],mext[ci])-calib.reproject(T,rMi,rMe),axis=1)
    tMi,tMe=g['calib64/%d/Mint_true'%ci],g['calib64/%d/Mext_true'%ci]
    dt_dev=np.linalg.norm(calib.reproject(T,mint[ci],mext[ci])-calib.reproject(T,tMi,tMe),axis=1).max()
    dt_ref=np.linalg.norm(calib.reproject(T,rMi,rMe)-calib.reproject(T,tMi,tMe),axis=1).max()
    rows.append((ci,int(idv.sum()),int(irf.sum()),bool(np.array_equal(idv,irf)),ed[idv&irf].sum(),er[idv&irf].sum(),d.max(),abs(mint[ci][0,0]-rMi[0,0])/rMi[0,0],dt_dev,dt_ref, ed[idv].sum(), er[irf].sum()))
rows=np.array(rows,dtype=object)
for r in rows:
    if (not r[3]) or r[6]>0.8: print('cam %d inl dev %d ref %d same %s obj dev %.4f ref %.4f dmax %.3f frel %.4f | vs truth: dev %.2f ref %.2f | own-inlier obj dev %.3f ref %.3f'%tuple(r))
d=np.array([r[6] for r in rows],float); same=np.array([r[3] for r in rows],bool)
print('same inliers',same.sum(),'dmax percentiles (all)',np.percentile(d,[50,90,100]),'(same-inlier cams)',np.percentile(d[same],[50,90,100]))
ro=np.array([r[4]/max(r[5],1e-12) for r in rows],float); print('obj ratio pct',np.percentile(ro,[0,50,90,100]), 'share<=1',(ro<=1+1e-6).mean())
fr=np.array([r[7] for r in rows],float); print('frel pct',np.percentile(fr,[50,90,100]))
tv=np.array([[r[8],r[9]] for r in rows],float); print('vs truth: dev median %.3f max %.3f | ref median %.3f max %.3f'%(np.median(tv[:,0]),tv[:,0].max(),np.median(tv[:,1]),tv[:,1].max()))
