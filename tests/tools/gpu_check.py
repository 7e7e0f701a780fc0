"""Ad-hoc GPU bring-up check (not collected by pytest): HIP path vs oracle on configs 2 and 3.  Lives under tests/ because it
loads the CPU checker (oracle/), which only test infrastructure may do."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'automatic-ar_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, aar, oracle_lib as ol
print('devices', aar.device_count())
ds = aar.synth(2)
o = ol.Oracle(ds)
for mode in (aar.RES_F64, aar.RES_F32):
    with aar.Problem(ds, residual_mode=mode) as p:
        r, ss = p.eval_residuals(ds.x_full)
        ro = o.residuals(ds.x_full, res_mode=mode)
        print('mode', mode, 'residual max abs diff', np.abs(r-ro).max(), 'ss', ss, (ro**2).sum())
with aar.Problem(ds, residual_mode=aar.RES_F64) as p:
    H, B, ss = p.eval_normal_equations(ds.x_full)
    Ho, Bo = o.normal_equations(ds.x_full, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F64)
    print('JtJ max abs diff', np.abs(H-Ho).max(), 'rel', np.abs(H-Ho).max()/np.abs(Ho).max(), 'B rel', np.abs(B-Bo).max()/np.abs(Bo).max())
    mu = np.diag(Ho).max()
    for m in (mu, mu*1e-3, mu*1e-6):
        d = p.eval_damped_step(ds.x_full, m)
        do = o.damped_solve(ds.x_full, m, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F64)
        print('mu %.3g delta rel diff'%m, np.abs(d-do).max()/np.abs(do).max())
for cfg in (2, 3):
    ds = aar.synth(cfg); o = ol.Oracle(ds)
    with aar.Problem(ds) as p:
        t=time.time(); x, rep = p.lm_solve(ds.x_full); t=time.time()-t
        print('cfg', cfg, 'GPU LM iters', rep['iterations'], 'err', rep['final_err'], 'stop', rep['stop_code'], 'secs', rep['solve_seconds'], 'it/s', rep['iterations']/rep['solve_seconds'])
        print('   trace', [(round(t_['err'],3), t_['tries']) for t_ in rep['trace']][:20])
        print('   rmse', p.reproj_stats(x), o.reproj_stats(x))
        print('   stage', p.stage_times())
        for k in range(3):
            x, rep = p.lm_solve(ds.x_full); print('   repeat it/s', rep['iterations']/rep['solve_seconds'])
    xo, repo = o.lm_solve(ds.x_full, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32, threads=8)
    print('   oracle analytic LM iters', repo['iterations'], repo['final_err'], 'max|dx|', np.abs(x-xo).max(), 'rmse', o.reproj_stats(xo)['rmse'])
