"""Condenses a round's PMC summaries (scripts/pmc_summary.py output) into the two small JSON files bench.py quotes:
   scripts/pmc_to_json.py <tag> gpurun_out/prof_<tag>  ->  profiles/<round>_clutter_500_pmc.json, profiles/<round>_pathtrace_pmc.json
(<round> = the tag's first three characters, e.g. r03).
HBM bytes: FETCH_SIZE and WRITE_SIZE are KiB.  MI355X_MICROARCH.md ("HBM"): on gfx950 FETCH_SIZE tallies the 128-byte requests of
16 B/lane reads at 64 B, i.e. at HALF their bytes; WRITE_SIZE is exact.  These kernels mix 8 and 16 B/lane loads, so their reads lie
between the raw figure and twice it: `fetch_bytes_raw` is the counter, `fetch_bytes_upper` = 2 x raw, and the `hbm_bytes_*` totals use
the upper figure (the conservative one for a kernel that is priced on how little it moves).
VALU busy = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); lane utilisation = SQ_THREAD_CYCLES_VALU /
(64 x SQ_ACTIVE_INST_VALU); waves_parked_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES."""
import csv, json, os, sys
tag, d = sys.argv[1], sys.argv[2]
rnd = tag[:3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    t = {}
    for r in csv.DictReader(open(path)):
        t.setdefault(r["kernel"], {})[r["counter"]] = float(r["mean_per_dispatch"])
    return t


def metrics(c):
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    return {"valu_busy": c["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cycles),
            "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
            "waves_parked_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
            "valu_insts_per_dispatch": c["SQ_INSTS_VALU"],
            "fetch_bytes_raw": c["FETCH_SIZE"] * 1024.0, "fetch_bytes_upper": 2.0 * c["FETCH_SIZE"] * 1024.0,
            "write_bytes": c["WRITE_SIZE"] * 1024.0,
            "gpu_cycles_per_dispatch": cycles}


def find(t, prefix):
    ks = [k for k in t if k.startswith(prefix)]
    if not ks:
        raise SystemExit("no kernel %s* in %s" % (prefix, sorted(t)))
    return metrics(t[ks[0]]), ks[0]


src = "profiles/%s_pmc_%%s.csv, rocprofv3 --pmc, separate passes (scripts/gpu_profile_r03.sh); reads: FETCH_SIZE x 2 (gfx950 tallies 16 B/lane reads at half their bytes: upper bound for this mix of 8 and 16 B/lane loads)" % tag
t = load(os.path.join(d, "pmc_clutter_500.csv"))
(roots, _), (frame, _) = find(t, "fw_roots_kernel"), find(t, "fw_kernel")
out = {"workload": "clutter_500_4k, one frame at a time (scripts/gpu_prof_case.py clutter_500 3840 2160 3 u8)",
       "hbm_bytes_per_frame": roots["fetch_bytes_upper"] + roots["write_bytes"] + frame["fetch_bytes_upper"] + frame["write_bytes"],
       "hbm_bytes_per_frame_fetch_raw": roots["fetch_bytes_raw"] + roots["write_bytes"] + frame["fetch_bytes_raw"] + frame["write_bytes"],
       "valu_busy": (roots["valu_insts_per_dispatch"] + frame["valu_insts_per_dispatch"]) * 4.0 /
                    (1024.0 * (roots["gpu_cycles_per_dispatch"] + frame["gpu_cycles_per_dispatch"])),
       "lane_utilisation": frame["lane_utilisation"], "fw_roots_kernel": roots, "fw_kernel": frame, "source": src % "clutter_500"}
json.dump(out, open(os.path.join(ROOT, "profiles", rnd + "_clutter_500_pmc.json"), "w"), indent=1)
t = load(os.path.join(d, "pmc_pathtrace.csv"))
k, name = find(t, "pq_kernel") if any(x.startswith("pq_kernel") for x in t) else find(t, "pt_kernel")
out = {"workload": "pathtrace.json 1920x1080, lightSamples 16, one launch = 1024 strata (scripts/gpu_prof_case.py pathtrace 1920 1080 2 spp1024)",
       "kernel": name, "strata_per_launch": 1024,
       "hbm_bytes_per_launch": k["fetch_bytes_upper"] + k["write_bytes"], "hbm_bytes_per_launch_fetch_raw": k["fetch_bytes_raw"] + k["write_bytes"],
       "valu_busy": k["valu_busy"], "lane_utilisation": k["lane_utilisation"], name.split("<")[0]: k, "source": src % "pathtrace"}
json.dump(out, open(os.path.join(ROOT, "profiles", rnd + "_pathtrace_pmc.json"), "w"), indent=1)
print(open(os.path.join(ROOT, "profiles", rnd + "_pathtrace_pmc.json")).read())
print(open(os.path.join(ROOT, "profiles", rnd + "_clutter_500_pmc.json")).read())
