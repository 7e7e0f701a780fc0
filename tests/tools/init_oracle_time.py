"""One-core timing of the oracle's Initializer restatement (oracle/init_oracle.cpp) on the scenes of scripts/init_bench.py, and a
check that the GPU path returns the same ids.  Not collected by pytest; test infrastructure (it loads oracle/).

    python tests/tools/init_oracle_time.py [--frames 60 200 500]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "automatic-ar_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import aar  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, nargs="+", default=[60, 200, 500])
    ap.add_argument("--cams", type=int, default=8)
    ap.add_argument("--markers", type=int, default=40)
    a = ap.parse_args()
    for F in a.frames:
        ds = aar.synth(3, num_cams=a.cams, num_markers=a.markers, num_frames=F)
        K = ds.cam_mats.reshape(-1, 3, 3)
        dists = [np.zeros(5)] * ds.num_cams
        det = aar.Detections(ds.num_cams, int(ds.frame_ids.max()) + 1, ds.frame_ids[ds.obs_frame], ds.cam_ids[ds.obs_cam],
                             ds.marker_ids[ds.obs_marker], ds.obs_uv)
        t0 = time.perf_counter()
        r = O.init_run(det.num_cams, det.num_frames, det.det_frame, det.det_cam, det.det_id, det.det_uv, 0.05, K, dists)
        line = dict(frames=F, detections=int(ds.num_obs), oracle_seconds_1_core=time.perf_counter() - t0)
        if aar.device_count() > 0:
            out = aar.initializer_run(det, K, dists, 0.05)
            line["same_ids"] = bool(np.array_equal(r["frame_ids"], out.frame_ids) and np.array_equal(r["marker_ids"], out.marker_ids))
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
