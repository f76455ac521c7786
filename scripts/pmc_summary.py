"""Condenses rocprofv3 --pmc counter_collection CSVs into one small table: per kernel and counter, the number
of dispatches and the mean value per dispatch.  usage: pmc_summary.py out.csv in1.csv [in2.csv ...]"""
import collections, csv, re, sys
out, ins = sys.argv[1], sys.argv[2:]
agg = collections.OrderedDict()
for f in ins:
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        agg.setdefault((name, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
with open(out, "w") as fp:
    w = csv.writer(fp)
    w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
    for (k, c), v in agg.items():
        w.writerow([k, c, len(v), "%.6g" % (sum(v) / len(v))])
