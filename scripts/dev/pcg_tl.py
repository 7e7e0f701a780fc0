import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")]
import aar
os.environ["AAR_SOLVER"] = "pcg"
for cfg in (3, 5):
    ds = aar.synth(cfg)
    with aar.Problem(ds) as p:
        p.lm_solve(ds.x_full)
        print("config", cfg, p.pcg_iterations(), flush=True)
