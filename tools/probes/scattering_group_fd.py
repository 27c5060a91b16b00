import numpy as np
from tests import scenarios as sc
from tests.emulation import run_emulated_vector

def mixture(x, nw=1, vector=False):
    s = sc.regression_1d(nw)
    alt = s["geometry"]["altitudes"]; ng = len(alt)
    ka = s["extinction"]; wa = np.full_like(ka, 0.93)
    nl = 16; l = np.arange(nl)
    ba = np.zeros(nl); ba[0], ba[1], ba[2] = 1.0, 0.08, 0.5
    bc = (2*l+1)*0.6**l
    e = 0.3*ka; wc = 0.95
    k = ka + x[:,None]*e
    sca = ka*wa + x[:,None]*e*wc
    w = sca/k
    beta = (ka*wa)[None]*ba[:,None,None] + (x[:,None]*e*wc)[None]*bc[:,None,None]
    beta = beta/sca[None]
    out = dict(s, extinction=k, ssa=w, leg=beta)
    meta = dict(e=e, wc=wc, bc=bc, sca=sca)
    if vector:
        out = sc.to_vector(out, rayleigh_fraction=None)
        # polarised constituent: add Rayleigh-like a2/b1 to the constituent only
        v = out["leg"]
        frac = (x[:,None]*e*wc/sca)   # [ng, nw]
        v[4*2+1] = 3.0*frac; v[4*2+3] = -np.sqrt(6.0)*0.5*frac; v[2::4][2] = 0.0
        out["leg"] = v
        meta["pol"] = (3.0, -np.sqrt(6.0)*0.5)
    return out, meta

ng = 25
x0 = np.ones(ng)
for vector in (False, True):
    s, M = mixture(x0, vector=vector)
    k, w, beta = s["extinction"], s["ssa"], s["leg"]
    S = 3 if vector else 1
    if vector:
        nl = 16
        dleg = np.zeros_like(beta)
        # d_legendre = constituent coefficients - mixture coefficients, per Greek coefficient
        bc4 = np.zeros((4*nl,)); bc4[0::4] = M["bc"]; bc4[4*2+1] = M["pol"][0]; bc4[4*2+3] = M["pol"][1]
        dleg = bc4[:,None,None] - beta
    else:
        dleg = M["bc"][:,None,None] - beta
    out = run_emulated_vector(s, derivatives=True, d_legendre=[dleg])
    d_ext = M["e"]; d_ssa = M["e"]*(M["wc"]-w)/k; scatf = M["e"]*M["wc"]/M["sca"]
    for q in (0, 3, 9, 17, 24):
        h = 1e-5
        xp, xm = x0.copy(), x0.copy(); xp[q]+=h; xm[q]-=h
        rp = run_emulated_vector(mixture(xp, vector=vector)[0])["radiance"]; rm = run_emulated_vector(mixture(xm, vector=vector)[0])["radiance"]
        fd = (rp-rm)/(2*h)
        if vector:
            an = d_ext[q,0]*out["d_extinction"][...,q] + d_ssa[q,0]*out["d_ssa"][...,q] + scatf[q,0]*out["d_scat"][:,:,:,0,q]
        else:
            an = d_ext[q,0]*out["d_extinction"][...,q] + d_ssa[q,0]*out["d_ssa"][...,q] + scatf[q,0]*out["d_scat"][:,:,0,q]
        print("S",S,"q",q,"max|fd-an|/max|an| =", np.abs(fd-an).max()/np.abs(an).max(), " (scat part share %.2f)"%(np.abs(scatf[q,0]*(out["d_scat"][:,:,:,0,q] if vector else out["d_scat"][:,:,0,q])).max()/np.abs(an).max()))
