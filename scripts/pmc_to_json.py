"""Condenses a round's PMC summaries (scripts/pmc_summary.py output) into the two small JSON files bench.py quotes:
   scripts/pmc_to_json.py <tag> gpurun_out/prof_<tag>  ->  profiles/r02_clutter_500_pmc.json, profiles/r02_pathtrace_pmc.json
HBM bytes = (FETCH_SIZE + WRITE_SIZE) KiB x 1024, raw (the gfx950 x2 correction of FETCH_SIZE is calibrated for 16 B/lane
streaming reads only; these kernels read 8 B per lane, so reads lie between 1x and 2x the raw value).  VALU busy =
SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); lane utilisation = SQ_THREAD_CYCLES_VALU /
(64 x SQ_ACTIVE_INST_VALU)."""
import csv, json, os, sys
tag, d = sys.argv[1], sys.argv[2]
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
            "fetch_bytes_raw": c["FETCH_SIZE"] * 1024.0, "write_bytes": c["WRITE_SIZE"] * 1024.0,
            "gpu_cycles_per_dispatch": cycles}


t = load(os.path.join(d, "pmc_clutter_500.csv"))
roots, frame = metrics(t["fw_roots_kernel<true, 3>"]), metrics(t["fw_kernel<0, true, 3>"])
out = {"workload": "clutter_500_4k, one frame at a time (scripts/gpu_prof_case.py clutter_500 3840 2160 3 u8)",
       "hbm_bytes_per_frame": roots["fetch_bytes_raw"] + roots["write_bytes"] + frame["fetch_bytes_raw"] + frame["write_bytes"],
       "valu_busy": (roots["valu_insts_per_dispatch"] + frame["valu_insts_per_dispatch"]) * 4.0 /
                    (1024.0 * (roots["gpu_cycles_per_dispatch"] + frame["gpu_cycles_per_dispatch"])),
       "lane_utilisation": frame["lane_utilisation"], "fw_roots_kernel": roots, "fw_kernel": frame,
       "source": "profiles/%s_pmc_clutter_500.csv, rocprofv3 --pmc, separate passes (scripts/gpu_profile_r02.sh); FETCH_SIZE raw" % tag}
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_clutter_500_pmc.json"), "w"), indent=1)
t = load(os.path.join(d, "pmc_pathtrace.csv"))
k = metrics(t["pt_kernel<true, 3>"])
out = {"workload": "pathtrace.json 1920x1080, lightSamples 16, one launch = 4 strata (scripts/gpu_prof_case.py pathtrace 1920 1080 2)",
       "hbm_bytes_per_launch": k["fetch_bytes_raw"] + k["write_bytes"], "valu_busy": k["valu_busy"],
       "lane_utilisation": k["lane_utilisation"], "pt_kernel": k,
       "source": "profiles/%s_pmc_pathtrace.csv, rocprofv3 --pmc, separate passes (scripts/gpu_profile_r02.sh); FETCH_SIZE raw" % tag}
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_pathtrace_pmc.json"), "w"), indent=1)
print(open(os.path.join(ROOT, "profiles", "r02_pathtrace_pmc.json")).read())
print(open(os.path.join(ROOT, "profiles", "r02_clutter_500_pmc.json")).read())
