"""Fill the @@NAME@@ placeholders of DESIGN.md section 6 from gpurun_out/r03final (tools/r03_final_profiles.sh).
Run after tools/r03_collect.sh; idempotent only on a DESIGN.md that still has the placeholders (keep
tools/r03_DESIGN.template.md as the source: this writes DESIGN.md from it)."""
import csv, json, os, re, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
O = os.path.join(R, "gpurun_out", "r03final")
b = json.load(open(os.path.join(O, "bench.json")))
K = {(k["kernel"], k["launch"]): k for k in b["kernels"]}
n = 1500000
def ms(name, launch): return K[(name, launch)]["avg_ms"]
v = {}
v["VALUE"] = "%.2fe10" % (b["value"]/1e10); v["STEP"] = "%.3f" % b["ms_per_step"]
v["QLT"] = "%.3f" % b["qlt"]["ms_per_run"]; v["CAAS"] = "%.3f" % b["caas"]["ms_per_run"]
v["CAASFRAC"] = "%.0f" % (100*64.0*n*40/(b["caas"]["ms_per_run"]*1e-3)/8e12)
v["DOWN"] = "%.3f" % ms("qlt_down_kernel", 8192); v["UP"] = "%.3f" % ms("tree_up_kernel", 8192)
v["RED"] = "%.3f" % ms("caas_reduce_kernel", n); v["FIN"] = "%.3f" % ms("caas_finish_kernel", n)
v["DOWNF"] = "%.1f" % (100*K[("qlt_down_kernel", 8192)]["frac"]); v["UPF"] = "%.1f" % (100*K[("tree_up_kernel", 8192)]["frac"])
v["REDF"] = "%.1f" % (100*K[("caas_reduce_kernel", n)]["frac"]); v["FINF"] = "%.1f" % (100*K[("caas_finish_kernel", n)]["frac"])
v["D2"] = "%.3f" % ms("qlt_down_kernel", 1024); v["U2"] = "%.3f" % ms("tree_up_kernel", 1024)
v["D3"] = "%.3f" % ms("qlt_down_kernel", 128); v["U3"] = "%.3f" % ms("tree_up_kernel", 128)
v["TOP"] = "%.3f" % ms("qlt_top_kernel", 2048)
rows = list(csv.DictReader(open(os.path.join(O, "kernel_stats_by_grid.csv"))))
def rp(prefix, wgs):
    for r in rows:
        if r["Kernel"].startswith(prefix) and int(r["Workgroups_X"]) == wgs: return "%.3f" % (float(r["AverageNs"])/1e6)
    return "n/a"
v["DOWNR"] = rp("qlt_down_sp_kernel<1, false>", 8192); v["UPR"] = rp("tree_up_micro_kernel<16, false, true", 8192)
v["REDR"] = rp("caas_reduce_kernel", 2930); v["FINR"] = rp("caas_finish_kernel", 2930)
t = json.load(open(os.path.join(O, "prof_traffic", "r_traffic.json")))["kernels"]
for key, name in (("DOWNT", "qlt_down_kernel"), ("UPT", "tree_up_kernel"), ("REDT", "caas_reduce_kernel"), ("FINT", "caas_finish_kernel")):
    v[key] = "%.2f" % (t[name]["hbm_bytes"]/1e9)
sh = open(os.path.join(O, "shards_and_chain.txt")).read()
def step(pat): return float(re.search(pat + r".*?ms/step ([0-9.]+)", sh).group(1))
s15, s75, s37, s18 = [step("%d cells, one stream:" % n) for n in (1500000, 750000, 375000, 187500)]
t15, t75, t37, t18 = [step(r"%d cells, two streams \(default\):" % n) for n in (1500000, 750000, 375000, 187500)]
v.update(S1500="%.3f" % s15, S750="%.3f" % s75, S375="%.3f" % s37, S187="%.3f" % s18,
         R750="%.2f" % (s15/s75), R375="%.2f" % (s15/s37), R187="%.2f" % (s15/s18),
         T1500="%.3f" % t15, T750="%.3f" % t75, T375="%.3f" % t37, T187="%.3f" % t18,
         Q750="%.2f" % (t15/t75), Q375="%.2f" % (t15/t37), Q187="%.2f" % (t15/t18))
bs = json.load(open(os.path.join(O, "bench_serial.json")))
v["STEPSER"] = "%.3f" % bs["ms_per_step"]; v["VALUESER"] = "%.2fe10" % (bs["value"]/1e10)
def opt(pat):
    m = re.search(pat, sh)
    return m.group(1) if m else "n/a"
v["CHAIN"] = opt(r"rank 0 of 8, 1.5M cells\s*: ([0-9.]+) ms")
v["CHAINL"] = opt(r"level mode, stub messages\): ([0-9.]+) ms")
cb = json.load(open(os.path.join(O, "caas_bfb.json")))["runs"]
for r in cb:
    if r["config"].startswith("1 GPU, 1.5M cells, 4 cells"): v["BFB4"], v["BFB4F"] = "%.3f" % r["caas_run_ms"], "%.1f" % (100*r["frac_of_8TBs"])
    if r["config"].startswith("1 GPU, 1.5M cells, 16 cells"): v["BFB16"], v["BFB16F"] = "%.3f" % r["caas_run_ms"], "%.1f" % (100*r["frac_of_8TBs"])
p = b["pack_unpack"]
v.update(SET="%.2f" % p["qlt_set_Qm_ms"], GET="%.2f" % p["qlt_get_Qm_ms"], ZC="%.2f" % p["qlt_zero_copy_run_ms"], SRG="%.2f" % p["qlt_set_run_get_ms"])
mx = open(os.path.join(O, "mix.txt")).read()
m = re.search(r"qlt ([0-9.]+) \| qlt_down/8192=([0-9.]+)", mx)
v["MIX"], v["MIXD"] = m.group(1), m.group(2)
c = b["cpu_baseline"]
v["CPU1"] = "%.1fe7" % (c["single_thread"]["value"]/1e7); v["CPU16"] = "%.1fe8" % (c["all_threads"]["value"]/1e8)
src = open(os.path.join(R, "tools", "r03_DESIGN.template.md")).read()
missing = set(re.findall(r"@@([A-Z0-9]+)@@", src)) - set(v)
assert not missing, missing
out = re.sub(r"@@([A-Z0-9]+)@@", lambda m: v[m.group(1)], src)
open(os.path.join(R, "DESIGN.md"), "w").write(out)
print("DESIGN.md written;", len(v), "values")
