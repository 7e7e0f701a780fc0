"""Register / scratch / occupancy table of every kernel, from hipcc's -Rpass-analysis=kernel-resource-usage remarks (cross-compiled, no GPU needed).

    python scripts/vgpr_counts.py > profiles/rNN_vgpr_counts.txt
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "eval_kernels solve_kernels pcg_kernels spcg_kernels init_kernels undistort ba_capi".split()
PATS = (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
        ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("spill_v", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"))


def main():
    rows = []
    for f in SRC:
        p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics",
                            "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ROOT, "automatic-ar_amd", "csrc", f + ".hip"), "-o", "/dev/null"],
                           capture_output=True, text=True)
        cur = None
        for line in p.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", name).replace("aar::", "").replace("(anonymous namespace)::", ""))
                cur = {"file": f, "name": name}
                rows.append(cur)
                continue
            for key, pat in PATS:
                m = re.search(pat, line)
                if m and cur is not None:
                    cur[key] = int(m.group(1))
    print("# hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage   (scripts/vgpr_counts.py)")
    print("# occ = waves per SIMD the register file allows (VGPR + AGPR <= 512 / occ); scratch = bytes per lane (0 = nothing spilled)")
    print("%-14s %-78s %5s %5s %5s %8s %7s %4s %7s" % ("file", "kernel", "VGPR", "AGPR", "SGPR", "scratch", "spillV", "occ", "LDS"))
    seen = set()
    for r in rows:
        if (r["file"], r["name"]) in seen:
            continue
        seen.add((r["file"], r["name"]))
        print("%-14s %-78s %5d %5d %5d %8d %7d %4d %7d" % (r["file"], r["name"][:78], r.get("vgpr", -1), r.get("agpr", -1), r.get("sgpr", -1),
                                                          r.get("scratch", -1), r.get("spill_v", -1), r.get("occ", -1), r.get("lds", -1)))


if __name__ == "__main__":
    main()
