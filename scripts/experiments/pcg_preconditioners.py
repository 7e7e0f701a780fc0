"""Experiment (CPU, numpy; the oracle is test infrastructure, nothing here is product code): CG iterations per damped solve along the exact LM
trajectory of config 3 for three preconditioners of the reduced system S (eta = 0.1):
    bj      block-Jacobi, the 6x6 diagonal blocks of S                                   (what csrc/pcg_kernels.hip builds)
    cam     the camera-camera block of S exactly + 6x6 blocks for the markers
    schur2  the arrow matrix [[S_cc, S_cm], [S_mc, blockdiag(S_mm)]] solved exactly          (needs the camera columns of S: ~1/6 of a Schur complement)
Committed output: profiles/r03_pcg_preconditioners.txt.      python scripts/experiments/pcg_preconditioners.py
"""
import os, sys, time
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[os.path.join(ROOT,"automatic-ar_amd"),os.path.join(ROOT,"tests"),os.path.join(ROOT,"scripts","experiments")]
import aar, oracle_lib as ol
import pcg_reduced_system as E

def schur_pcg2(U, W, V, bs, bf, mu, eta, nc, mode, max_it=2000):
    ns, nf = U.shape[0], V.shape[0]
    Vi = np.linalg.inv(V + mu*np.eye(6))
    Wb = W.reshape(ns, nf, 6)
    rhs = bs - np.einsum("sfi,fij,fj->s", Wb, Vi, bf.reshape(nf,6))
    def apply(p):
        t = np.einsum("fij,fj->fi", Vi, np.einsum("sfi,s->fi", Wb, p))
        return U@p + mu*p - np.einsum("sfi,fi->s", Wb, t)
    # exact S for building preconditioners (experiment only)
    S = U + mu*np.eye(ns) - np.einsum("sfi,fij,tfj->st", Wb, Vi, Wb)
    if mode == "bj":
        Mi = np.zeros((ns,ns))
        for a in range(ns//6): sl=slice(6*a,6*a+6); Mi[sl,sl]=np.linalg.inv(S[sl,sl])
    elif mode == "cam":      # camera block exact, marker blocks 6x6
        Mi = np.zeros((ns,ns)); Mi[:nc,:nc]=np.linalg.inv(S[:nc,:nc])
        for a in range(nc//6, ns//6): sl=slice(6*a,6*a+6); Mi[sl,sl]=np.linalg.inv(S[sl,sl])
    elif mode == "schur2":   # block 2x2 (cams | markers) with marker part block-diagonal: M = [[Scc, Scm],[Smc, D_m]] solved exactly
        D = np.zeros((ns-nc, ns-nc))
        for a in range((ns-nc)//6): sl=slice(6*a,6*a+6); D[sl,sl]=S[nc+6*a:nc+6*a+6, nc+6*a:nc+6*a+6]
        M = S.copy(); M[nc:,nc:] = D
        Mi = np.linalg.inv(M)
    prec = lambda r: Mi@r
    x=np.zeros(ns); r=rhs.copy(); z=prec(r); p=z.copy(); rz=r@z; r0=np.linalg.norm(rhs); it=0
    while it<max_it and np.linalg.norm(r)>eta*r0:
        Ap=apply(p); al=rz/(p@Ap); x+=al*p; r-=al*Ap; z=prec(r); rzn=r@z; p=z+(rzn/rz)*p; rz=rzn; it+=1
    return it

for cfg in ():
    ds = aar.synth(cfg) if cfg<=3 else aar.synth(5, num_frames=150)
    o = ol.Oracle(ds); ns = 6*(ds.num_cams-1+ds.num_markers-1); nc = 6*(ds.num_cams-1)
    x=ds.x_full; z=o.extract_z(x)
    H,B = o.normal_equations(x, z=z, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32)
    mu0 = H.diagonal().max()
    for mu in (mu0, mu0*1e-3, mu0*1e-6, mu0*1e-8):
        parts = E.split(H,B,ns)
        res = {m: schur_pcg2(*parts, mu, 0.1, nc, m) for m in ("bj","cam","schur2")}
        print("config", cfg, "mu/mu0 %.0e" % (mu/mu0), res, flush=True)

CFGS = [int(a) for a in sys.argv[1:]] or [3]
print("---- along the exact LM trajectory ----")
for cfg in CFGS:
    ds = aar.synth(cfg) if cfg <= 3 else aar.synth(5, num_frames=400)
    o = ol.Oracle(ds); ns = 6*(ds.num_cams-1+ds.num_markers-1); nc = 6*(ds.num_cams-1)
    x=ds.x_full; z=o.extract_z(x)
    H,B = o.normal_equations(x, z=z, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32)
    err = float(np.sum(o.residuals(x, z=z, res_mode=ol.RES_F32)**2)); mu=H.diagonal().max(); v=2.0; prev=err
    tot = {"bj":0,"cam":0,"schur2":0}
    for step in range(15):
        parts = E.split(H,B,ns)
        res = {m: schur_pcg2(*parts, mu, 0.1, nc, m) for m in tot}
        for m in tot: tot[m]+=res[m]
        print("step", step, "mu %.3e" % mu, res, flush=True)
        d = np.linalg.solve(H + mu*np.eye(H.shape[0]), B)
        zt = z+d; et = float(np.sum(o.residuals(x, z=zt, res_mode=ol.RES_F32)**2))
        L = 0.5*d@(mu*d-B); gain=(et-prev)/L
        if gain>0 and et<prev:
            mu*=max(0.33,1-(2*gain-1)**3); z=zt; err=et
            H,B = o.normal_equations(x, z=z, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32)
        else: mu*=v; v*=5
        if abs(prev-err)/(8.0*o.N) <= 1e-4: break
        prev=err
    print("totals", tot)
