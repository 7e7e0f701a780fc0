import sys, numpy as np
sys.path.insert(0,'automatic-ar_amd')
import aar
bad=0
for noise in (0.1, 0.3, 0.6, 1.0):
    for seed in (1, 2, 3):
        for hub in (False, True):
            ds = aar.synth(3, num_frames=120, noise_px=noise, seed=1000+seed)
            K = ds.cam_mats.reshape(-1,3,3)
            det = aar.Detections(ds.num_cams, int(ds.frame_ids.max())+1, ds.frame_ids[ds.obs_frame], ds.cam_ids[ds.obs_cam], ds.marker_ids[ds.obs_marker], ds.obs_uv)
            try:
                init = aar.initializer_run(det, K, [np.zeros(5)]*ds.num_cams, 0.05)
                with aar.Problem(init, with_huber=hub) as p:
                    r0 = p.reproj_stats(init.x_full)[0]
                    x, rep = p.lm_solve(init.x_full, trace_cap=16)
                    r1 = p.reproj_stats(x)[0]
                print(noise, seed, hub, 'rmse %.3f -> %.4f' % (r0, r1), 'iters', rep['iterations'], 'stop', rep['stop_code'], 'trial pts', rep['trial_points'])
            except Exception as e:
                bad+=1; print(noise, seed, hub, 'EXC', str(e)[:150])
print('failures', bad)
