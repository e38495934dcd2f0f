"""Merges the per-configuration PMC summaries of scripts/profile_r04.sh into profiles-style rNN_pmc_traffic.json:
HBM-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) keyed by kernel, stamped with the source hash of the library that ran.
Usage: pmc_traffic_merge.py <dir> <tag>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, tag = sys.argv[1:3]
out = {}
p = os.path.join(d, tag + "_pmc_traffic.json")           # written by kernel_table.py (config 2)
if os.path.exists(p):
    out.update(json.load(open(p)).get("kernels", {}))
p = os.path.join(d, "r04_cfg4_pmc_traffic_part.json")
if os.path.exists(p):
    out.update(json.load(open(p)))
p = os.path.join(d, tag + "_cfg3_pmc_fetch_write_per_kernel.json")
if os.path.exists(p):
    for k, v in json.load(open(p)).items():
        if "k_helm<12>" in k:
            out["k_helm<12>"] = {"bytes_per_launch": (2.0 * v["fetch_kb_p90"] + v["write_kb_p90"]) * 1024.0, "fetch_kb_p90": v["fetch_kb_p90"], "write_kb_p90": v["write_kb_p90"]}
stamp = os.path.join(ROOT, "nekstab_amd", "lib", "libnekstab_hip.so.srchash")
json.dump({"srchash": open(stamp).read().strip() if os.path.exists(stamp) else None, "kernels": out,
           "source": "scripts/profile_r04.sh / profile_r05.sh + profile_r05_b.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only); config 2: p90 over the launches of the bench command; config 4: p50 over launches of scripts/kernels3d_bench.py at a known basis index; config 3: k_helm<12>"},
          open(os.path.join(d, tag + "_pmc_traffic.json"), "w"), indent=1)
print(sorted(out))
