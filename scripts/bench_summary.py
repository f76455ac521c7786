"""Prints the figures of a bench.py line that one looks at first.  Usage: python scripts/bench_summary.py <file with the JSON line>"""
import json
import sys

for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)

    def show(name, x):
        roof = x.get("roofline", {})
        print("%-12s %s = %.1f  ms_per_step %.3f  frac %.3f  frac_walked %s  kernel_ms %s" % (
            name, x.get("metric"), x.get("value", float("nan")), x.get("ms_per_step", float("nan")), roof.get("frac", float("nan")),
            roof.get("frac_walked"), roof.get("kernel_ms", roof.get("kernel_ms_per_launch"))))

    show("line", d)
    for k, v in d.items():
        if isinstance(v, dict) and "roofline" in v:
            show(k, v)
    print("cpu_baseline", d.get("cpu_baseline"))
