"""Merges the per-configuration PMC summaries of scripts/profile_r06.sh / profile_r06_b.sh into <dir>/<tag>_pmc_traffic.json: HBM-side bytes
per launch (2 x FETCH_SIZE + WRITE_SIZE) keyed by kernel, stamped with the source hash of the library that ran (bench.py looks the dominant
kernel of each --case up in it).  Usage: pmc_traffic_merge_r06.py <dir> <tag>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, tag = sys.argv[1:3]
out = {}
p = os.path.join(d, tag + "_pmc_traffic.json")           # config 2: written by kernel_table.py, extended here
if os.path.exists(p):
    out.update(json.load(open(p)).get("kernels", {}))
for part in ("r04_cfg4_pmc_traffic_part.json", "r06_cfg5_pmc_traffic_part.json"):       # (kernel_table_cfg4.py keeps its round-4 file name)
    q = os.path.join(d, part)
    if os.path.exists(q):
        out.update(json.load(open(q)))
for cfg3 in (tag + "_cfg3_pmc_fetch_write_per_kernel.json",):
    q = os.path.join(d, cfg3)
    if os.path.exists(q):
        for k, v in json.load(open(q)).items():
            if "k_helm<12>" in k:
                out["k_helm<12>"] = {"bytes_per_launch": (2.0 * v["fetch_kb_p90"] + v["write_kb_p90"]) * 1024.0, "fetch_kb_p90": v["fetch_kb_p90"], "write_kb_p90": v["write_kb_p90"]}
stamp = os.path.join(ROOT, "nekstab_amd", "lib", "libnekstab_hip.so.srchash")
sys.path.insert(0, ROOT)
import bench
prev = json.load(open(p)) if os.path.exists(p) else {}
json.dump({"srchash": open(stamp).read().strip() if os.path.exists(stamp) else None, "family_hash": bench.family_hashes(), "kernels": out,
           "source": "scripts/profile_r06.sh + profile_r06_b.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only); config 2: p90 over the launches of the bench command; "
                     "configs 4 and 5: p50 over full-work launches of scripts/kernels3d_bench.py / prof_cfg5.py; config 3: k_helm<12> (p90 over a bench run)"},
          open(p, "w"), indent=1)
print(sorted(out))
