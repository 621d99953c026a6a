#!/bin/bash
# Copy what tools/r03_final_profiles.sh produced (gpurun_out/r03final) into profiles/ under the round's name,
# stamping the HEAD into the JSON / CSV files that do not carry it yet.
cd "$(dirname "$0")/.."
O=gpurun_out/r03final
H=$(sed 's/^HEAD //' $O/HEAD.txt)
python3 - "$H" <<'PY'
import json, sys, shutil, os
H = sys.argv[1].strip(); O = "gpurun_out/r03final"; P = "profiles"
def jcopy(src, dst):
    d = json.load(open(os.path.join(O, src)))
    if isinstance(d, dict): d = dict(d, head=H)
    json.dump(d, open(os.path.join(P, dst), "w"), indent=1)
def tcopy(src, dst, comment="# "):
    if not os.path.exists(os.path.join(O, src)): return
    body = open(os.path.join(O, src)).read()
    open(os.path.join(P, dst), "w").write((comment + "HEAD " + H + "\n" if not body.startswith(comment + "HEAD") else "") + body)
jcopy("bench.json", "r03_bench.json")
jcopy("valu.json", "r03_valu.json")
jcopy("caas_bfb.json", "r03_caas_bfb.json")
d = json.load(open(os.path.join(O, "prof_traffic", "r_traffic.json"))); d["head"] = H
json.dump(d, open(os.path.join(P, "r03_traffic.json"), "w"), indent=1)
for src, dst in (("kernel_stats.csv", "r03_kernel_stats.csv"), ("kernel_stats_by_grid.csv", "r03_kernel_stats_by_grid.csv"),
                 ("caas_bfb.csv", "r03_caas_bfb.csv")):
    tcopy(src, dst)
for src, dst in (("pmc_table.txt", "r03_pmc_table.txt"), ("shards_and_chain.txt", "r03_shards_and_chain.txt"),
                 ("top_stamps.txt", "r03_top_stamps.txt"), ("setup_time.txt", "r03_setup_time.txt"), ("mix.txt", "r03_mix.txt"),
                 ("traffic_summary.txt", "r03_traffic_summary.txt")):
    tcopy(src, dst)
PY
ls -la profiles/ | grep r03
