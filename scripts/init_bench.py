"""Initializer timing: wall time of aar_initializer_run (GPU) per scene size.

    python scripts/init_bench.py [--frames 60 200 500 2000] [--cams 8] [--markers 40]

The scene is the synthetic camera ring of BASELINE.json's configs (0.3 px corner noise); candidates per camera-pair /
marker-pair set grow linearly with the frame count, the vote quadratically.  Prints one JSON line per size.  The CPU
restatement is timed beside it by tests/tools/init_oracle_time.py (only test infrastructure loads oracle/).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "automatic-ar_amd"))
import aar  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, nargs="+", default=[60, 200, 500, 2000])
    ap.add_argument("--cams", type=int, default=8)
    ap.add_argument("--markers", type=int, default=40)
    a = ap.parse_args()
    for F in a.frames:
        ds = aar.synth(3, num_cams=a.cams, num_markers=a.markers, num_frames=F)
        K = ds.cam_mats.reshape(-1, 3, 3)
        dists = [np.zeros(5)] * ds.num_cams
        det = aar.Detections(ds.num_cams, int(ds.frame_ids.max()) + 1, ds.frame_ids[ds.obs_frame], ds.cam_ids[ds.obs_cam],
                             ds.marker_ids[ds.obs_marker], ds.obs_uv)
        aar.initializer_run(det, K, dists, 0.05)          # warm-up (module load, first allocations)
        t0 = time.perf_counter()
        out = aar.initializer_run(det, K, dists, 0.05)
        gpu_s = time.perf_counter() - t0
        line = dict(frames=F, detections=int(ds.num_obs), cams=out.num_cams, markers=out.num_markers, gpu_seconds=gpu_s)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
