"""Condense a rocprofv3 --kernel-trace directory into one row per kernel with its per-launch times (what profiles/*_kernel_stats.csv
of the index and l = 3 legs hold).   python scripts/trace_table.py <trace dir> <out.csv> ["# comment line" ...]"""
import csv, glob, re, sys
src, dst, comments = sys.argv[1], sys.argv[2], sys.argv[3:]
per = {}
for f in glob.glob(f"{src}/*/*_kernel_trace.csv"):
    for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
        n = r["Kernel_Name"]
        n = re.sub(r"^void ", "", n)
        if "rocprim" in n or "hipcub" in n:
            m = re.search(r"(segmented_radix_sort|radix_sort|onesweep|scan|histogram|transform|partition|merge|reduce)", n)
            n = "rocprim " + (m.group(1) if m else "other")
        else:
            n = re.sub(r"\(.*", "", n)
        per.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open(dst, "w") as o:
    for c in comments:
        o.write(c if c.startswith("#") else "# " + c)
        o.write("\n")
    o.write("Name,Calls,AverageMs,TotalMs,PerLaunchMs\n")
    for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        shown = " ".join(f"{x:.3f}" for x in v[:10])
        o.write(f'"{n}",{len(v)},{sum(v) / len(v):.4f},{sum(v):.3f},"{shown}"\n')
print(open(dst).read()[:3000])
