"""Per-launch durations of one kernel family from a rocprofv3 --kernel-trace CSV, grouped by grid size.
usage: gemm_layers.py [trace_dir=/tmp/unet] [name_substring=qconv_gemm]"""
import collections, csv, glob, sys
root = sys.argv[1] if len(sys.argv) > 1 else "/tmp/unet"
pat = sys.argv[2] if len(sys.argv) > 2 else "qconv_gemm"
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:60], r["Grid_Size_X"], r["Grid_Size_Y"])].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    print(k, "calls", len(v), "avg_us %.2f" % (sum(v) / len(v)))
