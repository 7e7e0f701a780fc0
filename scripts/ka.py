import sys
sys.path.insert(0,'automatic-ar_amd'); sys.path.insert(0,'tests')
import numpy as np, aar
ds = aar.synth(2, noise_px=0.0)
with aar.Problem(ds, residual_mode=aar.RES_F64) as p:
    x, rep = p.lm_solve(ds.x_full, params=aar.lm_default_params(min_average_step_error_diff=1e-14))
    print(rep['iterations'], rep['final_err'], rep['stop_code'])
    worst=0
    for k,(a, b) in enumerate(zip(x.reshape(-1, 6), ds.x_truth.reshape(-1, 6))):
        dr=np.abs(aar.rodrigues_vec2mat(a[:3]) - aar.rodrigues_vec2mat(b[:3])).max(); dt=np.abs(a[3:] - b[3:]).max()
        if max(dr,dt)>2e-4: print(k, dr, dt, a, b)
    print(p.reproj_stats(x), p.reproj_stats(ds.x_truth))
