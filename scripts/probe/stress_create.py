# Diagnostic: create / solve / destroy in a loop; device memory must come back
import sys, ctypes, numpy as np
sys.path.insert(0, "automatic-ar_amd")
import aar
hip = ctypes.CDLL("libamdhip64.so")
def free_mem():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value
ds2, ds3 = aar.synth(2), aar.synth(3)
with aar.Problem(ds3) as p: p.lm_solve(ds3.x_full)
m0 = free_mem()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
    ds = ds3 if i % 3 == 0 else ds2
    with aar.Problem(ds, with_huber=(i % 5 == 0), intrinsics=(i % 7 == 0)) as p:
        x0 = p.x_with_intrinsics(ds.x_full) if i % 7 == 0 else ds.x_full
        x, rep = p.lm_solve(x0)
        assert rep["iterations"] > 0
m1 = free_mem()
print("free before %.1f MB after %.1f MB delta %.2f MB" % (m0 / 1e6, m1 / 1e6, (m0 - m1) / 1e6))
