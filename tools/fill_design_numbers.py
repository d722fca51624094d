#!/usr/bin/env python3
"""Regenerate DESIGN.md's section 4.4 from docs/design_4_4.template.md (its @@...@@ placeholders) and a
bench line (profiles/rNN_bench_cfg3_n1.json).
    python tools/fill_design_numbers.py profiles/r06_bench_cfg3_n1.json [profiles/r06_exact_pmc_traffic.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
line = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
k = {n: v["ms_per_step"] for n, v in d["kernels"].items()}
cb, ow, pc = d["cpu_baseline"], d.get("other_workloads", {}), d["pcie_inclusive"]


def ow_ms(prefix):
    for name, v in ow.items():
        if name.startswith(prefix):
            return "%.2f" % v["ms_per_step"] if "ms_per_step" in v else "n/a (%s)" % v.get("error", "?")[:40]
    return "n/a"


traffic = "not collected for this build"
if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):
    t = json.load(open(sys.argv[2]))
    g = [v for n, v in t["kernels"].items() if "gather" in n]
    if g and t.get("build_id") == d.get("library_build_id"):
        traffic = "%.3f GB = %.2f x algorithmic" % (g[0]["bytes"] / 1e9, g[0]["bytes"] / 1.6e9)
rep = {
    "STEP": "%.3f" % d["ms_per_step"], "VALUE": "%d" % round(d["value"]),
    "COUNT": "%.3f" % k.get("k_dsm_p3_count", 0), "SCATTER": "%.3f" % k.get("k_dsm_p3_scatter", 0),
    "PLACE": "%.3f" % k.get("k_dsm_p3_place", 0), "GATHER": "%.3f" % k.get("k_dsm_gather", 0),
    "ORTHO": "%.3f" % k.get("k_ortho_backward", 0), "FRAC": "%.3f" % d["roofline"]["frac"],
    "TRAFFIC": traffic, "WHOLE": "%.3f" % d["whole_step_hbm"]["frac"],
    "KD": "%.1f" % cb["kdtree_build_s"], "QUERY": "%.1f" % cb["query_s"], "MOSAIC": "%.1f" % cb["ortho_s"],
    "CPU": "%.2f" % cb["value"], "QUOTA": "%.2f" % cb.get("threads_quota_run", {}).get("Mcells_per_s", float("nan")),
    "FAST": "%.2f" % d.get("fast_mode", {}).get("ms_per_step", float("nan")),
    "ALL": "%.3f" % d.get("launch_skips", {}).get("all_launched", {}).get("ms_per_step", float("nan")),
    "ROUGH": "%.2f" % d.get("rough_terrain", {}).get("exact", {}).get("dsm_ms_per_call", float("nan")),
    "PCIE": "%.1f" % pc["ms"], "PCIEDSM": "%.1f" % pc["dsm_ms"],
    "CFG2": ow_ms("cfg2 (configs[1]"), "KNN": ow_ms("cfg2 --knn 4"), "COLOR": ow_ms("cfg3 --colored"),
    "CFG4": ow_ms("cfg4 at N = 1"), "CFG5": ow_ms("cfg5 at N = 1"),
}
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
sec = open(os.path.join(ROOT, "docs", "design_4_4.template.md")).read()
for key, val in rep.items():
    sec = sec.replace("@@%s@@" % key, val)
left = [w for w in sec.split("@@")[1::2]]
i, j = s.index("### 4.4 Measurements"), s.index("### 4.5 The other kernels")
s = s[:i] + sec + s[j:]
open(p, "w").write(s)
print("filled; build", d.get("library_build_id"), "placeholders left:", left)
