import sys, json
for l in sys.stdin:
    if not l.startswith("{"): continue
    j = json.loads(l)
    if "shape" in j: print(j["shape"], j["M"], j["N"], j["K"], "fp32", j["us_fp32_mfma"], "split", j["us_split"], "x", j["speedup"], "TFeq", j["tflops_equiv_split"], "err", "%.1e %.1e" % (j["max_err_fp32_mfma"], j["max_err_split"]), "stats %.1e" % j["stats_rel_diff"])
    else: print(j)
