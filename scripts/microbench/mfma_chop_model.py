"""Trajectory-level effect of the bf16 matrix pipe's in-group truncation, emulated: z = [h] @ W' with the packed bf16x3
slot layout of l2o_lstm_bx3.h; per (MFMA, lane group) the <= 8 products are chopped toward zero at 2^-24 of the group's
largest, summed exactly, then added to the accumulator chain in fp32 (RNE)."""
import sys, numpy as np, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import dill, oracle as O
from helpers import make_problem, rel_err
L2E=1.4426950408889634
d=dill.load(open('/root/repo/tests/golden/trained/dm_quadratic_d128/cw.l2l-0','rb'))
params={k:{v:np.asarray(a,np.float64) for v,a in m.items()} for k,m in d.items()}
def bf16(x):
    x=np.asarray(x,np.float32); u=x.view(np.uint32).astype(np.uint64)
    r=((u+0x7fff+((u>>16)&1))>>16)<<16
    return r.astype(np.uint32).view(np.float32).astype(np.float64)
def split3(x):
    x=np.asarray(x,np.float64); x1=bf16(x); x2=bf16(x-x1); x3=bf16(x-x1-x2); return [x1,x2,x3]
def slot_desc(j,r,h):
    k=j*4+r
    t={0:(0+h,0,0),1:(2+h,0,0),2:(4,0,h),3:(5,0,h),4:(0+h,0,1),5:(2+h,0,1),6:(0+h,0,2),7:(2+h,0,2),8:(0+h,1,0),9:(2+h,1,0),
       10:(4,h,0 if h else 2),11:(0+h,1,1),12:(2+h,1,1),13:(0+h,2,0),14:(2+h,2,0),15:(4,1+h,0 if h else 1)}
    if SWAP:
        t[3],t[12]=t[12],t[3]
    return t[k]
SWAP=False
def bias_level(kq,h): return h if kq==0 else (2 if (kq==1 and h==0) else -1)
def scale_rows(H):
    s=np.empty(4*H); s[:H]=-L2E; s[H:2*H]=2*L2E; s[2*H:3*H]=-L2E; s[3*H:]=-L2E; return s
def chop(p, emax):
    # toward zero at 2^(emax-24)
    q=np.ldexp(1.0, emax-24)
    return np.trunc(p/q)*q
def f32(x): return np.asarray(x,np.float64).astype(np.float32).astype(np.float64)
def z_chunk(h, Wp, bias, acc, mode):
    """h [N,20]; Wp [20,80] pre-scaled (float64); bias [80] pre-scaled or None; acc [N,80] fp32 accumulator in."""
    N=h.shape[0]
    if mode=='exact':
        return acc+h@Wp+(0 if bias is None else bias)
    hs=split3(h.astype(np.float32)); ws=split3(Wp.astype(np.float32) if False else Wp)  # weights split from float64
    bs=None if bias is None else split3(bias)
    if mode.startswith('unit_major'):
        # 5 MFMAs; group q of MFMA j holds the 6 (8) products of unit 4j+q ; bias = accumulator init
        a=f32(acc+(0 if bias is None else bias)) if bias is not None else acc
        prods=[(0,0),(0,1),(0,2),(1,0),(1,1),(2,0)]
        if '8' in mode: prods+= [(1,2),(2,1)]
        for j in range(5):
            for q in range(4):
                u=4*j+q
                P=np.stack([hs[xl][:,u:u+1]*ws[wl][u:u+1,:] for xl,wl in prods],2)   # [N,80,6]
                if 'nochop' not in mode:
                    m=np.abs(P).max(2); e=np.floor(np.log2(np.maximum(m,1e-300))).astype(int)
                    P=chop(P, e[:,:,None])
                a=f32(a+P.sum(2))
        return a
    global SWAP
    SWAP='swap' in mode
    a=acc.copy()
    if mode.endswith('biasinit') and bias is not None: a=f32(a+bias)
    for j in range(4):
        for q in range(4):
            cols=[]
            for r in range(4):
                for hh in range(2):
                    unit,xl,wl=slot_desc(j,r,hh)
                    if unit==5:
                        lv=bias_level(q,wl)   # slot index -> level
                        if lv>=0 and bs is not None and not mode.endswith('biasinit'):
                            cols.append(np.broadcast_to(bs[lv][None,:],(N,80)))
                        continue
                    u=4*unit+q
                    cols.append(hs[xl][:,u:u+1]*ws[wl][u:u+1,:])
            P=np.stack(cols,2)
            if 'nochop' not in mode:
                m=np.abs(P).max(2); e=np.floor(np.log2(np.maximum(m,1e-300))).astype(int)
                P=chop(P,e[:,:,None])
            a=f32(a+P.sum(2))
    return a
def sig_from(acc): return 1/(1+np.exp2(acc))   # acc = -log2e z
def run(mode,T,Bt,seed=14):
    prob,x0,arr=make_problem("quadratic",128,128,seed=seed)
    pr=O.Quadratic(prob.w[:Bt].astype(np.float64),prob.y[:Bt].astype(np.float64),batch_global=128)
    x=x0[:Bt].astype(np.float64); N=Bt*128; H=20
    s=scale_rows(H)
    W1=params['lstm_1']['w_gates']*s; b1=(params['lstm_1']['b_gates']+np.r_[np.zeros(2*H),np.ones(H),np.zeros(H)])*s
    W2=params['lstm_2']['w_gates']*s; b2=(params['lstm_2']['b_gates']+np.r_[np.zeros(2*H),np.ones(H),np.zeros(H)])*s
    h1=np.zeros((N,H));c1=h1.copy();h2=h1.copy();c2=h1.copy()
    fx=np.zeros(T+1)
    rnd=(lambda a:a) if mode=='exact' else f32
    def gates(acc,c):
        i,j,f,o=np.split(acc,4,1)
        cn=c/(1+np.exp2(f))+np.tanh(j/(2*L2E))/(1+np.exp2(i))
        hn=np.tanh(cn)/(1+np.exp2(o))
        return rnd(hn),rnd(cn)
    for t in range(T):
        fx[t]=pr.f(x); g=pr.grad(x).reshape(N,1)
        z=np.zeros((N,80))
        z=z_chunk(h1,W1[1:],b1,z,mode)                  # chunk L1H (+bias)
        z=rnd(z+g*W1[0:1])                              # input FMAs
        h1,c1=gates(z,c1)
        z=np.zeros((N,80))
        z=z_chunk(h2,W2[H:],b2,z,mode)                  # chunk L2B (+bias)
        z=z_chunk(h1,W2[:H],None,z,mode)                # chunk L2A
        h2,c2=gates(z,c2)
        dl=h2@params['linear']['w']+params['linear']['b']
        x=rnd(x+dl.reshape(x.shape))
    fx[T]=pr.f(x)
    return fx
if __name__=='__main__':
    T=int(sys.argv[1]) if len(sys.argv)>1 else 1000; Bt=int(__import__("os").environ.get("BT","2"))
    t0=time.time(); ex=run('exact',T,Bt); print("exact %.0fs"%(time.time()-t0), flush=True)
    for mode in sys.argv[2:] or ['packed_nochop','packed','packed_biasinit','unit_major','unit_major8']:
        t0=time.time(); f=run(mode,T,Bt)
        print("%-18s drift vs exact float64 at T=%d: %.3g (first 101: %.3g)  [%.0fs]"%(mode,T,rel_err(f,ex),rel_err(f[:101],ex[:101]),time.time()-t0),flush=True)
