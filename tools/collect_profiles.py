#!/usr/bin/env python3
"""Turn the raw outputs of tools/gpu_round.sh (gpurun_out/<tag>/) into the committed evidence under profiles/:
  <round>_kernel_stats.csv, <round>_kernel_stats_long.csv, <round>_kernel_stats_row_blocks.csv   rocprofv3 --kernel-trace --stats summaries (kernel names shortened)
  <round>_pmc_hbm.csv        FETCH_SIZE / WRITE_SIZE rows of the dominant kernels (separate --pmc passes)
  <round>_pmc_sq.csv         SQ counter passes of the two sweep kernels
  <round>_hbm_traffic.json   HBM bytes per pair of the dominant kernel per (series, path), FETCH_SIZE doubled (gfx950 correction,
                             MI355X_MICROARCH.md), VALU instructions per pair / VALU busy; stamped with the commit and the hash of the
                             kernel sources the counters were taken from -- bench.py refuses it for any other kernel sources
  <round>_*.json(l)          the bench lines of the same run
Usage: python tools/collect_profiles.py gpurun_out/<tag> <round>"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def short(name):
    m = re.search(r"::(\w+_kernel)(<[^>(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def pmc_rows(d):
    rows = []
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                rows.append({"kernel": short(r["Kernel_Name"]), "grid": int(r["Grid_Size"]), "counter": r["Counter_Name"], "value": float(r["Counter_Value"]),
                             "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
    return rows


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    for sub, dst in (("stats", "_kernel_stats.csv"), ("stats_long", "_kernel_stats_long.csv"), ("stats_rb", "_kernel_stats_row_blocks.csv")):
        for path in glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True):
            with open(path) as fh, open(os.path.join(prof, rnd + dst), "w") as out:
                wr = csv.writer(out)
                for k, row in enumerate(csv.reader(fh)):
                    if k > 0:
                        row[0] = short(row[0])
                    wr.writerow(row)
    def read1(name, default):
        try:
            with open(os.path.join(src, name)) as fh:
                return fh.read().strip() or default
        except OSError:
            return default
    commit = read1("commit.txt", "unknown")
    khash = read1("kernel_source_hash.txt", None)  # recorded on the GPU box by gpu_round.sh: the sources the counters were taken from
    if commit == "unknown":
        try:
            commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip() + " (at collection)"
        except Exception:
            pass
    traffic = {"commit": commit, "kernel_source_hash": khash or bench.kernel_source_hash(),
               "note": "FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md; separate --pmc passes of bench.py --steps 1 --warmup 0"}
    hbm_rows = []
    # (series, path, pmc dir stem, kernel regex, pairs per launch, SQ dir stem, SQ pairs)
    cases = (("affine", "fast_path", "pmc_fast", r"^fp_sweep_kernel", 100000, "pmc_sq_fast", 100000),
             ("affine", "general_path", "pmc_general", r"^fill_affine_kernel", 100000, None, 0),
             ("long", "const_long", "pmc_long", r"^cl_sweep_(wg_)?kernel", 1024, "pmc_sq_long", 1024))
    sq_lines = []
    for series, path, stem, kre, pairs, sqstem, sqpairs in cases:
        tot = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            rows = [r for r in pmc_rows(os.path.join(src, "%s_%s" % (stem, c))) if re.search(kre, r["kernel"]) and r["counter"] == c]
            hbm_rows += [dict(r, series=series, path=path) for r in rows]
            if rows:
                big = max(r["grid"] for r in rows)  # the sweep / full fill is the largest launch of its kind
                sel = [r for r in rows if r["grid"] == big]
                # the counter comes back per XCD / dimension instance: one launch = the sum over the rows of one dispatch
                tot[c] = sum(r["value"] for r in sel) / len(sel)  # one row per launch (the sizing call and the timed step): their mean
                tot["kernel"] = sel[0]["kernel"]
        if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
            hbm = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
            entry = {"kernel": tot["kernel"], "FETCH_SIZE_KB": tot["FETCH_SIZE"], "WRITE_SIZE_KB": tot["WRITE_SIZE"], "pairs_per_launch": pairs,
                     "hbm_bytes_per_launch": hbm, "hbm_bytes_per_pair": hbm / pairs}
            if sqstem:
                sq, kms = {}, {}
                for d in glob.glob(os.path.join(src, sqstem + "*")):
                    rows = [r for r in pmc_rows(d) if re.search(kre, r["kernel"])]
                    big = max([r["grid"] for r in rows] or [0])
                    for r in rows:
                        if r["grid"] == big:
                            sq.setdefault(r["counter"], []).append(r["value"])
                            kms.setdefault(r["counter"], []).append(r["ms"])
                if sq:
                    val = {k: sum(v) / len(v) for k, v in sq.items()}  # one row per launch and pass: the mean
                    for k in sorted(val):
                        sq_lines.append("%s,%s,%s,%f\n" % (series, tot["kernel"], k, val[k]))
                    if "SQ_INSTS_VALU" in val:
                        entry["valu_insts_per_pair"] = val["SQ_INSTS_VALU"] / sqpairs
                    if "SQ_ACTIVE_INST_VALU" in val and "GRBM_GUI_ACTIVE" in val:
                        # quad-cycles of VALU activity summed over the 1024 SIMDs vs the kernel's cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
                        entry["valu_busy"] = val["SQ_ACTIVE_INST_VALU"] * 4.0 / (val["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
                    if "SQ_WAVE_CYCLES" in val and "GRBM_GUI_ACTIVE" in val:
                        entry["waves_per_simd"] = val["SQ_WAVE_CYCLES"] * 4.0 / (val["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
                    if "SQ_WAIT_INST_LDS" in val and "SQ_WAVE_CYCLES" in val:
                        entry["wait_lds_frac_of_wave_cycles"] = val["SQ_WAIT_INST_LDS"] / val["SQ_WAVE_CYCLES"]
                    if "SQ_WAIT_ANY" in val and "SQ_WAVE_CYCLES" in val:
                        entry["wait_any_frac_of_wave_cycles"] = val["SQ_WAIT_ANY"] / val["SQ_WAVE_CYCLES"]
                # the 2-round launch of the same kernel (32 768 pairs), for the ramp / tail share of a small launch
                sq2 = {}
                for d in glob.glob(os.path.join(src, sqstem.replace("pmc_sq_", "pmc_sq32k_") + "*")):
                    rows = [r for r in pmc_rows(d) if re.search(kre, r["kernel"])]
                    big = max([r["grid"] for r in rows] or [0])
                    for r in rows:
                        if r["grid"] == big:
                            sq2.setdefault(r["counter"], []).append(r["value"])
                if sq2:
                    v2 = {k: sum(v) / len(v) for k, v in sq2.items()}
                    for k in sorted(v2):
                        sq_lines.append("%s@32768,%s,%s,%f\n" % (series, tot["kernel"], k, v2[k]))
                    if "SQ_ACTIVE_INST_VALU" in v2 and "GRBM_GUI_ACTIVE" in v2:
                        entry["valu_busy_at_32768_pairs"] = v2["SQ_ACTIVE_INST_VALU"] * 4.0 / (v2["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
                        entry["waves_per_simd_at_32768_pairs"] = v2.get("SQ_WAVE_CYCLES", 0) * 4.0 / (v2["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            traffic.setdefault(series, {})[path] = entry
    with open(os.path.join(prof, rnd + "_pmc_hbm.csv"), "w") as out:
        wr = csv.writer(out)
        wr.writerow(["series", "path", "kernel", "grid", "counter", "value_KB"])
        for r in hbm_rows:
            wr.writerow([r["series"], r["path"], r["kernel"], r["grid"], r["counter"], r["value"]])
    if sq_lines:
        with open(os.path.join(prof, rnd + "_pmc_sq.csv"), "w") as out:
            out.write("# rocprofv3 --pmc <SQ counters> passes (bench.py --steps 1 --warmup 0; affine: the 100 000-pair launch, affine@32768: --pairs 32768, long: --pairs 1024), mean over the\n"
                      "# launches of the kernel; SQ_*_CYCLES / WAIT / ACTIVE are in quad-cycles\nseries,kernel,counter,value\n")
            out.writelines(sq_lines)
    with open(os.path.join(prof, rnd + "_hbm_traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)
        fh.write("\n")
    for nm, dst in (("bench.json", "_bench.json"), ("bench_long.json", "_bench_long.json"), ("all_series.jsonl", "_all_series.jsonl"), ("host_entry.jsonl", "_host_entry.jsonl"),
                    ("shapes_affine.jsonl", "_shapes_affine.jsonl"), ("shapes_const.jsonl", "_shapes_const.jsonl"), ("shapes_local.jsonl", "_shapes_local.jsonl"), ("gsw_reads.jsonl", "_gsw_reads.jsonl"), ("cabi_n1_n2.jsonl", "_cabi_n1_n2.jsonl"),
                    ("pytest_gpu.log", "_pytest_gpu.log"), ("bench_2ranks_shared_gpu.json", "_bench_2ranks_shared_gpu.json"), ("lds_occupancy.txt", "_lds_occupancy.txt"), ("stress.log", "_stress.log"), ("switch_matrix.log", "_switch_matrix.log"), ("concurrent_pairs.json", "_concurrent_pairs.json"), ("pmc_c5_wg_ab.txt", "_pmc_c5_wg_ab.txt"), ("wg_occupancy.txt", "_wg_occupancy.txt"), ("pair_latency.jsonl", "_pair_latency.jsonl"), ("n1_cmd.json", "_n1_cmd.json"), ("gsw_threads.jsonl", "_gsw_threads.jsonl"),
                    ("lat_crossover.jsonl", "_lat_crossover.jsonl"), ("gsw_genome.jsonl", "_gsw_genome.jsonl"), ("few_long_pairs.jsonl", "_few_long_pairs.jsonl"),
                    ("long_pairs_farm_ab.jsonl", "_long_pairs_farm_ab.jsonl"), ("switch_matrix_farm.log", "_switch_matrix_farm.log"),
                    ("long_pairs.jsonl", "_long_pairs.jsonl"), ("long_pairs_first_call.jsonl", "_long_pairs_first_call.jsonl"), ("pmc_long_pair.txt", "_pmc_long_pair.txt"), ("pmc_long_pair.json", "_pmc_long_pair.json"),
                    ("kernel_stats_long_pair_affine_1M.csv", "_kernel_stats_long_pair.csv"), ("alloc_probe.txt", "_alloc_probe_round_end.txt"), ("stress_routes.log", "_stress_routes.log")):
        if nm == "gsw_genome.jsonl" and os.path.exists(os.path.join(src, nm)) and sum(1 for _ in open(os.path.join(src, nm))) < 2:
            continue  # (a run with GNX_SKIP_GENOME=1 holds the 1e8-base line only: the committed file keeps its 1e9 / 3e9 lines)
        p = os.path.join(src, nm)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copyfile(p, os.path.join(prof, rnd + dst))
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
