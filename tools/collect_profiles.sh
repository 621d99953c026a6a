#!/bin/bash
# Copy what tools/final_profiles.sh produced (gpurun_out/${RN}final) into profiles/ under the round's name,
# stamping the HEAD into the JSON / CSV files that do not carry it yet.
cd "$(dirname "$0")/.."
RN=${RN:-r05}
O=gpurun_out/${RN}final
H=$(sed 's/^HEAD //' $O/HEAD.txt)
python3 - "$H" "$RN" <<'PY'
import json, sys, shutil, os
H = sys.argv[1].strip(); RN = sys.argv[2]; O = "gpurun_out/" + RN + "final"; P = "profiles"
def jcopy(src, dst):
    d = json.load(open(os.path.join(O, src)))
    if isinstance(d, dict): d = dict(d, head=H)
    json.dump(d, open(os.path.join(P, dst), "w"), indent=1)
def tcopy(src, dst, comment="# "):
    if not os.path.exists(os.path.join(O, src)): return
    body = open(os.path.join(O, src)).read()
    open(os.path.join(P, dst), "w").write((comment + "HEAD " + H + "\n" if not body.startswith(comment + "HEAD") else "") + body)
jcopy("bench.json", RN + "_bench.json")
jcopy("valu.json", RN + "_valu.json")
jcopy("caas_bfb.json", RN + "_caas_bfb.json")
d = json.load(open(os.path.join(O, "prof_traffic", "r_traffic.json"))); d["head"] = H
json.dump(d, open(os.path.join(P, RN + "_traffic.json"), "w"), indent=1)
for sub, dst in (("prof_traffic_bfb", "_caas_bfb_traffic.json"), ("prof_traffic_mix", "_mix_traffic.json")):
    f = os.path.join(O, sub, "r_traffic.json")
    if os.path.exists(f):
        d = json.load(open(f)); d["head"] = H
        json.dump(d, open(os.path.join(P, RN + dst), "w"), indent=1)
for src, dst in (("kernel_stats.csv", RN + "_kernel_stats.csv"), ("kernel_stats_by_grid.csv", RN + "_kernel_stats_by_grid.csv"),
                 ("caas_bfb.csv", RN + "_caas_bfb.csv")):
    tcopy(src, dst)
for src, dst in (("pmc_table.txt", RN + "_pmc_table.txt"), ("shards_and_chain.txt", RN + "_shards_and_chain.txt"),
                 ("top_stamps.txt", RN + "_top_stamps.txt"), ("setup_time.txt", RN + "_setup_time.txt"), ("mix.txt", RN + "_mix.txt"),
                 ("traffic_summary.txt", RN + "_traffic_summary.txt"), ("other_configs.txt", RN + "_other_configs.txt"),
                 ("traffic_bfb_summary.txt", RN + "_caas_bfb_traffic_summary.txt")):
    tcopy(src, dst)
PY
ls -la profiles/ | grep $RN
