"""Print the numbers of a final measurement set (profiles/rNN/final/) in the order DESIGN section 5 / README quote them.
    python tools/summarize_final.py profiles/r03/final"""
import csv, glob, json, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "profiles/r03/final"
def L(name):
    p = os.path.join(d, name)
    return json.load(open(p)) if os.path.exists(p) else None
def line(name):
    j = L(name)
    if not j: return "%s: missing" % name
    r = j["roofline"]
    return "%-34s %7.1f M/s  %7.3f ms  %5.0f GB/s  frac %.3f" % (name[6:-5], j["value"] / 1e6, r["kernel_ms_avg"], r["achieved"], r["frac"])
j = L("bench_default.json")
if j:
    print("default line: %.1f M %.3f ms frac %.3f | cfg4 %.1f M (%.3f) cfg5 %.1f M (%.3f) | cpu C %.3f M B2 %.3f M B1 %.0f" % (
        j["value"] / 1e6, j["roofline"]["kernel_ms_avg"], j["roofline"]["frac"],
        j["other_configs"]["cfg4"]["value"] / 1e6, j["other_configs"]["cfg4"]["roofline"]["frac"],
        j["other_configs"]["cfg5"]["value"] / 1e6, j["other_configs"]["cfg5"]["roofline"]["frac"],
        j["cpu_baseline"]["value"] / 1e6, j["cpu_baseline"]["others"]["B2_vectorised_numpy"]["value"] / 1e6,
        j["cpu_baseline"]["others"]["B1_reference_shaped"]["value"]))
for f in sorted(glob.glob(os.path.join(d, "bench_cfg*.json"))):
    print(line(os.path.basename(f)))
for m in ("linear", "cubic"):
    j = L("bench_symbols_%s.json" % m)
    if j:
        f, s = j["device_frame_fused"], j["device_frame_separate_calls"]
        print("symbols %-6s fused %.4f ms %.0f GB/s frac %.3f | separate %.3f ms frac %.3f | batch %.0f (first %.0f) frame %.0f single %.0f cpu %.0f symbols/s" % (
            m, f["ms"], f["GBps"], f["frac_of_8TBps"], s["ms"], s["frac_of_8TBps"], j["end_to_end_batch"]["symbols_per_s"],
            j["end_to_end_batch"].get("first_call_symbols_per_s", 0), j["end_to_end_frame"]["symbols_per_s"],
            j["end_to_end_single"]["symbols_per_s"], j["cpu_reference_shaped"]["symbols_per_s"]))
for f in sorted(glob.glob(os.path.join(d, "kernel_stats_*.csv"))):
    rows = [r for r in csv.DictReader(open(f)) if "ivs::" in r["Name"]]
    print(os.path.basename(f)[13:-4] + ": " + "; ".join("%s %s x %.1f us" % (r["Name"].split("(")[0].replace("void ivs::", "")[:52], r["Calls"], float(r["AverageNs"]) / 1e3) for r in rows[:6]))
for t, alg in (("cubic", 17024000000), ("cfg5", 17583354200), ("nan10", 17024000000)):
    j = L("pmc_%s.json" % t)
    if j:
        fe = sum(v.get("FETCH_SIZE", 0) for v in j.values()); wr = sum(v.get("WRITE_SIZE", 0) for v in j.values())
        print("traffic %s: HBM bytes / algorithmic = %.4f" % (t, (2 * fe + wr) * 1024 / alg))
j = L("pmc_symbols.json")
if j and "ivs::frame_fused_kernel" in "".join(j):
    for k, v in j.items():
        if "frame_fused" in k: print("frame pass WRITE_SIZE %.0f KB FETCH_SIZE %.0f KB" % (v.get("WRITE_SIZE", 0), v.get("FETCH_SIZE", 0)))
