#!/usr/bin/env python3
"""Turn the raw outputs of tools/gpu_round.sh (gpurun_out/<tag>/) into the committed evidence under profiles/:
  <round>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (kernel names shortened)
  <round>_pmc_hbm_{fastpath,general}.csv   FETCH_SIZE / WRITE_SIZE rows of the dominant kernel (separate --pmc passes)
  <round>_hbm_traffic.json   HBM bytes per launch / per pair of the dominant kernel, FETCH_SIZE doubled (gfx950 correction,
                             MI355X_MICROARCH.md) -- what bench.py reports as roofline.traffic
  <round>_bench.json         the bench line of the same run
Usage: python tools/collect_profiles.py gpurun_out/<tag> <round>"""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"::(\w+_kernel)(<[^>(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def pmc_rows(d, kernel_re):
    rows = []
    path = os.path.join(d, "pmc_counter_collection.csv")
    if not os.path.exists(path):
        return rows
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if re.search(kernel_re, r["Kernel_Name"]):
                rows.append({"kernel": short(r["Kernel_Name"]), "grid": int(r["Grid_Size"]), "vgpr": int(r["VGPR_Count"]), "lds": int(r["LDS_Block_Size"]),
                             "counter": r["Counter_Name"], "value": float(r["Counter_Value"])})
    return rows


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    with open(os.path.join(src, "stats", "stats_kernel_stats.csv")) as fh, open(os.path.join(prof, rnd + "_kernel_stats.csv"), "w") as out:
        rd = csv.reader(fh)
        wr = csv.writer(out)
        for k, row in enumerate(rd):
            if k > 0:
                row[0] = short(row[0])
            wr.writerow(row)
    traffic = {}
    for key, sub, kre, pairs_field in (("fast_path", "", r"fp_sweep_kernel", 8), ("general_path", "_gen", r"fill_affine_kernel", 4)):
        allrows = []
        tot = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            rows = pmc_rows(os.path.join(src, "pmc%s_%s" % (sub, c)), kre)
            if key == "fast_path":
                rows = [r for r in rows]
            allrows += rows
            if rows:
                big = max(rows, key=lambda r: r["grid"])  # the forward sweep / full fill is the largest launch
                tot[c] = sum(r["value"] for r in rows if r["grid"] == big["grid"]) / max(1, sum(1 for r in rows if r["grid"] == big["grid"]))
                tot["pairs"] = big["grid"] // 64 * pairs_field
                tot["kernel"] = big["kernel"]
        if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
            with open(os.path.join(prof, "%s_pmc_hbm_%s.csv" % (rnd, "fastpath" if key == "fast_path" else "general")), "w") as out:
                wr = csv.writer(out)
                wr.writerow(["kernel", "grid", "vgpr", "lds", "counter", "value_KB"])
                for r in allrows:
                    wr.writerow([r["kernel"], r["grid"], r["vgpr"], r["lds"], r["counter"], r["value"]])
            hbm = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
            traffic[key] = {"kernel": tot["kernel"], "pairs_per_launch_upper": tot["pairs"], "FETCH_SIZE_KB": tot["FETCH_SIZE"], "WRITE_SIZE_KB": tot["WRITE_SIZE"],
                            "hbm_bytes_per_launch": hbm, "note": "FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md; separate --pmc passes, bench.py --steps 1 --warmup 0"}
    # SQ counter passes (32768 pairs): one line per counter for the sweep kernel; VALU instructions per pair feed bench.py's
    # second (VALU issue) roofline
    sq = {}
    kms = {}
    for gi in (1, 2, 3):
        path = os.path.join(src, "pmc_sq%d" % gi, "pmc_counter_collection.csv")
        if not os.path.exists(path):
            continue
        with open(path) as fh:
            for r in csv.DictReader(fh):
                if re.search(r"fp_sweep_kernel", r["Kernel_Name"]):
                    sq.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                    kms.setdefault(r["Counter_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    if sq:
        with open(os.path.join(prof, rnd + "_pmc_sq.csv"), "w") as out:
            out.write("# rocprofv3 --pmc <SQ counters> passes on bench.py --pairs 32768 --steps 1 --warmup 0: fp_sweep_kernel (4096 waves); "
                      "SQ_*_CYCLES / WAIT / ACTIVE are in quad-cycles summed over the SEs\n")
            out.write("counter,value,kernel_ms\n")
            for k in sorted(sq):
                out.write("%s,%f,%.3f\n" % (k, sum(sq[k]) / len(sq[k]), sum(kms[k]) / len(kms[k])))
        if "SQ_INSTS_VALU" in sq and "fast_path" in traffic:
            traffic["fast_path"]["valu_insts_per_pair"] = sum(sq["SQ_INSTS_VALU"]) / len(sq["SQ_INSTS_VALU"]) / 32768.0
            traffic["fast_path"]["lds_insts_per_pair"] = sum(sq.get("SQ_INSTS_LDS", [0])) / max(1, len(sq.get("SQ_INSTS_LDS", [0]))) / 32768.0
            traffic["fast_path"]["lds_bank_conflict_cycles"] = sum(sq.get("SQ_LDS_BANK_CONFLICT", [0])) / max(1, len(sq.get("SQ_LDS_BANK_CONFLICT", [0])))
            if "SQ_ACTIVE_INST_VALU" in sq and "GRBM_GUI_ACTIVE" in sq:
                # quad-cycles of VALU activity summed over the 1024 SIMDs vs the kernel's cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
                act = sum(sq["SQ_ACTIVE_INST_VALU"]) / len(sq["SQ_ACTIVE_INST_VALU"]) * 4.0
                cyc = sum(sq["GRBM_GUI_ACTIVE"]) / len(sq["GRBM_GUI_ACTIVE"]) / 8.0
                traffic["fast_path"]["valu_busy"] = act / (cyc * 1024.0)
                traffic["fast_path"]["clock_ghz"] = cyc / (sum(kms["GRBM_GUI_ACTIVE"]) / len(kms["GRBM_GUI_ACTIVE"]) * 1e6)
    bench = None
    for nm in ("bench.json", "stats_bench.json"):
        try:
            with open(os.path.join(src, nm)) as fh:
                for line in fh:
                    if line.startswith("{"):
                        bench = json.loads(line)
            if bench:
                break
        except OSError:
            pass
    pairs = bench["config"]["pairs_per_gpu"] if bench else 100000
    for v in traffic.values():
        v["pairs_per_launch"] = pairs
        v["hbm_bytes_per_pair"] = v["hbm_bytes_per_launch"] / pairs
        v.pop("pairs_per_launch_upper", None)
    old = {}
    tpath = os.path.join(prof, rnd + "_hbm_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as fh:
            old = json.load(fh)
    old.update(traffic)
    with open(tpath, "w") as fh:
        json.dump(old, fh, indent=1)
        fh.write("\n")
    if os.path.exists(os.path.join(src, "bench.json")):
        shutil.copyfile(os.path.join(src, "bench.json"), os.path.join(prof, rnd + "_bench.json"))
    for nm, dst in (("all_series.jsonl", "_all_series.jsonl"), ("host_entry.jsonl", "_host_entry.jsonl"), ("bench_general_path.json", "_bench_general_path.json")):
        if os.path.exists(os.path.join(src, nm)) and os.path.getsize(os.path.join(src, nm)) > 0:
            shutil.copyfile(os.path.join(src, nm), os.path.join(prof, rnd + dst))
    if os.path.exists(os.path.join(src, "all_series.jsonl")):  # per-topic views of the same lines
        with open(os.path.join(src, "all_series.jsonl")) as fh:
            lines = [ln for ln in fh if ln.startswith("{")]
        longs = [ln for ln in lines if '"series": "one AffineGap pair' in ln or '"series": "C5 miniature' in ln]
        n1 = [ln for ln in lines if '"series": "AffineGapChunk' in ln or '"series": "multipleAffineGap' in ln]
        n2 = [ln for ln in lines if 'DynamicAln' in ln]
        for part, dst in ((longs, "_long_bench.jsonl"), (n1, "_n1_bench.jsonl"), (n2, "_n2_bench.jsonl")):
            if part:
                with open(os.path.join(prof, rnd + dst), "w") as out:
                    out.writelines(part)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
