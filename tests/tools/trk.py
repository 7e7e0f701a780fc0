"""Ad-hoc check of aar_track against the golden track fixtures (not collected by pytest; run from the repo root)."""
import sys
sys.path.insert(0,'automatic-ar_amd'); sys.path.insert(0,'tests')
import numpy as np, aar, oracle_lib as ol
from conftest import load_golden
for name in ("g_track_cfg2","g_track_cfg2_huber"):
    ds,g=load_golden(name); hub=bool(g["with_huber"][0])
    with aar.Problem(ds, with_huber=hub) as p:
        if hub: p.set_huber_delta(10.0)
        x,it,err=p.track(g["track_x0"])
    ns=6*(ds.num_cams-1)+6*(ds.num_markers-1)
    rel=np.abs(err-g["track_err"])/np.maximum(g["track_err"],1e-12)
    print(name,'max rel err diff',rel.max(),'argmax',rel.argmax(), err[rel.argmax()], g["track_err"][rel.argmax()], 'its',it[rel.argmax()],g["track_iterations"][rel.argmax()])
    print('  max dx', np.abs(x[ns:]-g["track_x"][ns:]).max(), 'iters eq frac', np.mean(it==g["track_iterations"]), 'max it diff', np.abs(it-g["track_iterations"]).max())
    print('  worst 5 rel', np.sort(rel)[-5:])
