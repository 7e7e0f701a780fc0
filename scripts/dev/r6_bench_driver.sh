mkdir -p gpurun_out/r6c
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6c/bench_driver.json 2> gpurun_out/r6c/bench_driver.err; tail -3 gpurun_out/r6c/bench_driver.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r6c/bench_driver.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "cg/step", d["pcg_iterations_per_lm_step"], "pose", d["pose_delta_vs_direct"], "roofline", d["roofline"])
for w,o in d["other_workloads"].items():
    print(w, o["value"], o["solver_resolved"], o["cg_iterations_per_lm_step"], o["pose_delta_vs_direct"]); print("   amdahl", json.dumps(o["amdahl"])[:1200])
PY
