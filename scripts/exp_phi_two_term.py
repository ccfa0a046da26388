"""numpy experiment (CPU): what bounds the Fourier features error against f64 -- the fp32 scaled state, the base sin/cos, or the fp32
angle-addition chain.  Output recorded in DESIGN.md section 4.1."""
import numpy as np
f32=np.float32
rng=np.random.default_rng(0)
lo=np.array([-1.2,-0.07]); hi=np.array([0.6,0.07])
lo32=lo.astype(f32); hi32=hi.astype(f32)
M=200000
s=(lo+(hi-lo)*rng.random((M,2))).astype(f32)
# f64 reference
st=(s.astype(np.float64)-lo)/(hi-lo)
c=np.array([(i,j) for i in range(6) for j in range(6)][1:])
phi64=np.cos(np.pi*(st@c.T))
def sincospi(x):  # emulate: exact-ish fp32 polynomial result: use float64 sin/cos rounded to f32 (<=0.5ulp; device is <=1.7ulp)
    return np.sin(np.pi*x.astype(np.float64)).astype(f32), np.cos(np.pi*x.astype(np.float64)).astype(f32)
def chain(s1,c1,n=5):
    C=[np.ones_like(c1),c1]; S=[np.zeros_like(s1),s1]
    for k in range(2,n+1):
        cn=(np.float32(-1)*S[k-1]*s1 + (C[k-1]*c1)).astype(f32)   # fma emul: compute in f64 then round once
        cn=(C[k-1].astype(np.float64)*c1 - S[k-1].astype(np.float64)*s1)
        # emulate fma(-s,s1, fl(c*c1))
        cn=( (C[k-1]*c1).astype(f32).astype(np.float64) - S[k-1].astype(np.float64)*s1.astype(np.float64)).astype(f32)
        sn=( (S[k-1]*c1).astype(f32).astype(np.float64) + C[k-1].astype(np.float64)*s1.astype(np.float64)).astype(f32)
        C.append(cn); S.append(sn)
    return np.stack(C,1),np.stack(S,1)
def project(st32, corr=None):
    out=[]
    tabs=[]
    for d in range(2):
        s1,c1=sincospi(st32[:,d])
        if corr is not None:  # first-order correction for the low part l: sin(pi(h+l)) ~ s + pi l c ; cos ~ c - pi l s
            l=corr[:,d]
            pl=(np.float32(np.pi)*l).astype(f32)
            s1n=(s1.astype(np.float64)+pl.astype(np.float64)*c1).astype(f32)
            c1n=(c1.astype(np.float64)-pl.astype(np.float64)*s1).astype(f32)
            s1,c1=s1n,c1n
        tabs.append(chain(s1,c1))
    (Cx,Sx),(Cv,Sv)=tabs
    cols=[]
    for (i,j) in c:
        re=((Cx[:,i]*Cv[:,j]).astype(f32).astype(np.float64) - Sx[:,i].astype(np.float64)*Sv[:,j]).astype(f32)
        cols.append(re)
    return np.stack(cols,1)
inv=(f32(1)/(hi32-lo32)).astype(f32)
# current: s~ = (s-lo)*inv in fp32
st_cur=((s-lo32)*inv).astype(f32)
print("current (one-term s~):", np.abs(project(st_cur)-phi64).max())
# two-term: hi = fl((s-lo)*inv); lo part = exact residual of true s~ (computed in f64) - hi
st_true=(s.astype(np.float64)-lo)/(hi-lo)
h=st_true.astype(f32); l=(st_true-h.astype(np.float64)).astype(f32)
print("two-term s~ + 1st-order correction:", np.abs(project(h,l)-phi64).max())
# perfect s~ in f64 fed to f32 sincos (i.e. only chain + poly error)
print("one-term but correctly rounded s~ (true s~ rounded to f32):", np.abs(project(h)-phi64).max())
# exact base sincos (f64) then chain in f32
def project_exact_base(st_true):
    tabs=[]
    for d in range(2):
        s1=np.sin(np.pi*st_true[:,d]).astype(f32); c1=np.cos(np.pi*st_true[:,d]).astype(f32)
        tabs.append(chain(s1,c1))
    (Cx,Sx),(Cv,Sv)=tabs
    cols=[]
    for (i,j) in c:
        cols.append(((Cx[:,i]*Cv[:,j]).astype(f32).astype(np.float64) - Sx[:,i].astype(np.float64)*Sv[:,j]).astype(f32))
    return np.stack(cols,1)
print("exact (f64-rounded) base sin/cos, fp32 chain + product:", np.abs(project_exact_base(st_true)-phi64).max())
