import json,sys
d=json.load(open(sys.argv[1]))
print("value %.4g  ms/step %.3f  qlt %.3f ms  caas %.3f ms  check %s" % (d["value"], d["ms_per_step"], d["qlt"]["ms_per_run"], d["caas"]["ms_per_run"], d.get("check_ok")))
for k in d["kernels"][:8]: print("  %-26s launch %-8d avg_ms %.4f  %s" % (k["kernel"], k["launch"], k["avg_ms"], ("%.0f GB/s (%.1f%%)" % (k["achieved_GBs"], 100*k["frac"])) if "frac" in k else ""))
