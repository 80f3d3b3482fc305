"""CPU study (round 3): how far is the Ritz vector at a convergence check from the final one, entry-wise, against what the
Lanczos quantities say (resid / gap in 2-norm, x max|ev| per entry, difference of consecutive checks)?  python tests/tools/err_study.py n mode seed"""
import json, os, sys
import numpy as np, scipy.sparse as sp
from scipy.linalg import eigh_tridiagonal
from scipy.sparse.csgraph import connected_components
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
from oracle import ncuts_ref
from oracle.gen_fullsize import MODES, chunk_for
import gpu_model as gm
n=int(sys.argv[1]); mode=sys.argv[2]; seed=int(sys.argv[3])
cfg=MODES[mode]; ch=chunk_for(n,mode,seed)
A=ncuts_ref.affinity_sparse(ch["points"],ch["tarl"],ch["dino"],alpha=cfg["alpha"],theta=cfg["theta"],gamma=cfg["gamma"]); T=cfg["T"]
def solve(w, ids):
    n=w.shape[0]; d=np.asarray(w.sum(axis=0)).ravel()+1.0; s=1/np.sqrt(d)
    Wm=(sp.diags(s)@(w+sp.identity(n))@sp.diags(s)).tocsr(); u1=np.sqrt(d/d.sum())
    v=gm.start_vector(ids); v-=u1*(u1@v); v/=np.linalg.norm(v); V=[v]; al=[];be=[]; vp=np.zeros(n); bp=0.0
    checks=[]
    for j in range(min(4000,n-1)):
        y=Wm@v; a=v@y; wv=y-a*v-bp*vp; g=u1@wv; b=np.sqrt(max(wv@wv-g*g,0.0)); wv=wv-g*u1
        al.append(a); be.append(b); m=j+1
        if m%16==0 or b<=1e-14:
            th,S=eigh_tridiagonal(np.array(al),np.array(be[:-1]),select="i",select_range=(max(m-2,0),m-1))
            r=abs(b*S[-1,-1]); checks.append((m,th[-1],th[0],r,S[:,-1].copy()))
            if r<=1e-10 or b<=1e-14: break
        vp,bp=v,b; v=wv/b; V.append(v)
    Vm=np.stack(V,1)
    def rv(c):
        x=Vm[:,:c[0]]@c[4]; return gm.fix_sign(x/np.linalg.norm(x))
    evf=rv(checks[-1]); prev=None; pr=None
    for c in checks[:-1]:
        if c[3]>1e-2: continue
        ev=rv(c); e=ev-evf
        gap=c[1]-c[2]
        est2=c[3]/gap
        dl=None
        if prev is not None:
            rho=c[3]/pr; dl=np.abs(ev-prev).max()*rho/(1-rho) if rho<1 else np.inf
        print("n=%5d m=%3d r=%.1e gap=%.1e e2=%.1e einf=%.1e | r/gap=%.1e ratio2=%.2f | einf/(e2*maxev)=%.2f | (r/gap)*maxev=%.1e ratio=%.2f | delta-est=%s ratio=%s | range=%.2e"%(
            n,c[0],c[3],gap,np.linalg.norm(e),np.abs(e).max(),est2,est2/np.linalg.norm(e),np.abs(e).max()/(np.linalg.norm(e)*np.abs(ev).max()),
            est2*np.abs(ev).max(), est2*np.abs(ev).max()/np.abs(e).max(), dl and "%.1e"%dl, dl and "%.2f"%(dl/np.abs(e).max()), ev.max()-ev.min()))
        prev=ev; pr=c[3]
    d_=d; mask,mcut,_=gm.sweep(evf,d_,w)
    return mask, mcut<T
level=[(A,np.arange(n))]
cnt=0
while level and cnt<40:
    nxt=[]
    for (w,lab) in level:
        nn=w.shape[0]
        if not gm._eligible(nn,n,0.01): continue
        nc,comp=connected_components(w,directed=False)
        if nc>1:
            for idx in gm.split_components(nc,comp): nxt.append((w[idx][:,idx],lab[idx]))
            continue
        mask,split=solve(w,lab); cnt+=1
        if split: nxt.append((w[mask][:,mask],lab[mask])); nxt.append((w[~mask][:,~mask],lab[~mask]))
    level=nxt
