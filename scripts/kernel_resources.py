#!/usr/bin/env python3
"""Register / LDS / scratch table of every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage).

    python scripts/kernel_resources.py [filter-substring]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wno-unused-value", "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("NSK_CXXFLAGS", "").split() + [
       "-o", "/tmp/_nsk_ru.so", os.path.join(ROOT, "nekstab_amd", "csrc", "nsk.hip")]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: .*?Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("void ", "")}
        rows.append(cur)
        continue
    m = re.search(r"remark: .*?\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
print("%-64s %6s %6s %8s %6s %8s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "LDS"))
for r in rows:
    if flt in r["name"]:
        print("%-64s %6s %6s %8s %6s %8s" % (r["name"][:64], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("ScratchSize", "?"), r.get("Occupancy", "?"), r.get("LDS Size", "?")))
