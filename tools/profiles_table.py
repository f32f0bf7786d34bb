"""Print the markdown rows of profiles/README.md's per-run table from profiles/<tag>_kernel_stats_<name>.csv and <tag>_bench_<name>.json."""
import csv, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
names = ["empty8x8_1M", "empty8x8_1M_k20", "doorkey8x8_1M", "lavacrossing_512k", "empty16x16_full_256k", "empty8x8_4M", "lavacrossing_4M", "lavacrossing_1M",
         "lavacrossing_1M_newlevel", "lavacrossing_512k_newlevel", "doorkey8x8_1M_newlevel", "dynobs8x8_1M", "dynobs16x16_1M", "multiroom_n6_256k_newlevel", "keycorridor_s3r3_256k_newlevel", "fourrooms_full_128k", "fourrooms_full_512k", "multiroom_n6_full_128k", "multiroom_n6_256k",
         "fourrooms_1M", "empty16x16_512k", "keycorridor_s6r3_512k", "obstructedmaze_2dlhb_256k", "empty8x8_1M_partial_onehot"]
names = [n for n in names if os.path.exists(os.path.join(P, "%s_kernel_stats_%s.csv" % (tag, n)))]
for n in names:
    rows = list(csv.DictReader(open(os.path.join(P, "%s_kernel_stats_%s.csv" % (tag, n)))))
    b = json.loads(open(os.path.join(P, "%s_bench_%s.json" % (tag, n))).read().strip().split("\n")[-1])
    r = b["roofline"]
    ks = [x for x in rows if any(k in x["Name"] for k in ("k_step", "k_dynobs", "k_levelgen", "k_onehot")) and int(x["Calls"]) > 8]
    def short(x):
        s = x["Name"]
        s = s[s.index("k_"):]
        return s.split("(")[0].replace(" ", "")
    kern = "; ".join("`%s` %.2f µs × %s (min %.1f, max %.1f)" % (short(x), float(x["AverageNs"]) / 1e3, x["Calls"], float(x["MinNs"]) / 1e3, float(x["MaxNs"]) / 1e3) for x in ks)
    step = [x for x in ks if "k_step" in x["Name"]][0]
    avg = float(step["AverageNs"]) / 1e3
    envs = b["config"]["envs_per_gpu"]
    B = r["algorithmic_bytes_per_env_step"]
    tb = B * envs / avg / 1e6
    pairs = ("%.2f" % r["event_pair_us"]) if r.get("event_pair_us") else "—"
    print("| `%s` | %s | %.2f µs/step; span %.2f, pairs %s | %s B × %s ÷ %.2f µs = **%.2f TB/s = %.3f** |" % (
        n, kern, b["ms_per_step"] * 1e3, r["span_us_per_step"], pairs, "{:,}".format(B), "{:,}".format(envs), avg, tb, tb / 8.0))
