"""Summarise the counter CSVs written by tools/pmc_one_gemm.sh: python tools/pmc_summary.py gpurun_out/pmc_<name>"""
import collections, csv, glob, os, sys
d = sys.argv[1]
KEYS = ("gemm", "flash", "gn_", "layernorm", "ff_geglu", "norm_linear", "ln_qkv", "cross_attn", "conv3x3_")
for sub in sorted(os.listdir(d)):
    f = os.path.join(d, sub, "out_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if any(s in r["Kernel_Name"] for s in KEYS):
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print(f"{sub:5s} {c:32s} n={len(v)} mean={sum(v) / len(v):.4g}   [{k}]")
    kt = os.path.join(d, sub, "out_kernel_trace.csv")
    if sub == "sq1" and os.path.exists(kt):
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)) if any(s in r["Kernel_Name"] for s in KEYS)]
        print("durations us:", " ".join(f"{x:.1f}" for x in durs))
