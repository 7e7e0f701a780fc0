import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "automatic-ar_amd"), os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")]
import aar
from conftest import load_golden
from pose_metrics import pose_delta_max
for name in ("g1_cfg2", "g1_cfg2_intr", "g1_cfg3_cut"):
    ds, g = load_golden(name)
    intr = name.endswith("_intr")
    with aar.Problem(ds, solver="direct", intrinsics=intr) as p:
        x0 = p.x_with_intrinsics(ds.x_full) if intr else ds.x_full
        xd, _ = p.lm_solve(x0)
    for w32 in ("1", "0"):
        os.environ["AAR_PCG_W32"] = w32
        with aar.Problem(ds, solver="pcg", intrinsics=intr) as p:
            x, rep = p.lm_solve(x0)
        print(name, "W32", w32, rep["iterations"], pose_delta_max(ds, x, xd))
