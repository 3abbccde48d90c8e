"""CPU prototype (not part of the product): residual history of the V-cycle-preconditioned CG of tools/mg_proto.py with
three, four and five damped-Jacobi sweeps per smoothing pass (dampings as shipped, and reciprocal Chebyshev roots on
[lo, 2] of D^-1 A).  Usage: python tools/mg_proto_sweeps.py 512 1.0   (size, prolongation scaling)"""
import sys, numpy as np
sys.path.insert(0, __file__.rsplit('/', 1)[0])
import mg_proto as M
def pcg_hist(op, Mf, b, maxit=8):
    x=np.zeros_like(b); r=b.copy(); nb=np.linalg.norm(b)
    z=Mf(r); p=z.copy(); rz=(r*z).sum(); h=[]
    for it in range(maxit):
        q=op.A(p); a=rz/(p*q).sum(); x+=a*p; r-=a*q
        h.append(np.linalg.norm(r)/nb)
        z=Mf(r); rz2=(r*z).sum(); p=z+(rz2/rz)*p; rz=rz2
    return h
n=int(sys.argv[1]); g=float(sys.argv[2])
op=M.model(n); rng=np.random.default_rng(0); b=rng.standard_normal((n,n)); b[0,0]=0
def cheb_om(k, lo, hi=2.0):
    # dampings = 1/roots of Chebyshev poly on [lo,hi] of D^-1 A spectrum
    j=np.arange(k); th=np.cos(np.pi*(2*j+1)/(2*k)); lam=0.5*(hi+lo)+0.5*(hi-lo)*th
    return tuple(1.0/lam)
V={ "3: 1.25/.5/1.0":dict(nu=3,om=(1.25,0.5,1.0)),
    "4: 1.25/.5/1.0/.5":dict(nu=4,om=(1.25,0.5,1.0,0.5)),
    "4: cheb[.5,2]":dict(nu=4,om=cheb_om(4,0.5)),
    "4: cheb[.4,2]":dict(nu=4,om=cheb_om(4,0.4)),
    "4: cheb[.3,2]":dict(nu=4,om=cheb_om(4,0.3)),
    "4: cheb[.6,2]":dict(nu=4,om=cheb_om(4,0.6)),
    "3: cheb[.5,2]":dict(nu=3,om=cheb_om(3,0.5)),
    "5: cheb[.4,2]":dict(nu=5,om=cheb_om(5,0.4)),
}
for name,kw in V.items():
    mg=M.MG(op,gamma=g,**kw)
    h=pcg_hist(op,lambda r:mg.cyc(0,r),b)
    print(f"{name:22s}", [round(o,3) for o in kw['om']], ' '.join(f"{v:.1e}" for v in h), flush=True)
