#!/bin/bash
# Counters of the sweep of ONE long pair: rocprofv3 kernel stats, one SQ group, FETCH_SIZE and WRITE_SIZE in separate --pmc passes (as
# MI355X_MICROARCH.md prescribes) of `python tools/long_pairs.py gpu <case>` -> <out>/pmc_long_pair.{txt,json}; copy to profiles/r6_pmc_long_pair.{txt,json}.
# Usage (GPU box, repo root): bash tools/pmc_long_pair.sh gpurun_out/<tag> [case ...]
out=$PWD/$1; shift; cases=${@:-affine_1M}
repo=$PWD; mkdir -p $out; export TMPDIR=/tmp
python -c 'import bench; print(bench.kernel_source_hash())' > $out/kernel_source_hash.txt
cd /tmp
for c in $cases; do
  rm -rf /tmp/lp_$c; mkdir -p /tmp/lp_$c
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/lp_$c/stats -o stats --output-format csv -- python $repo/tools/long_pairs.py gpu $c > /tmp/lp_$c/row.jsonl 2> $out/pmc_long_pair.err
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES -d /tmp/lp_$c/sq -o pmc --output-format csv -- python $repo/tools/long_pairs.py gpu $c > /dev/null 2>> $out/pmc_long_pair.err
  for k in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $k -d /tmp/lp_$c/$k -o pmc --output-format csv -- python $repo/tools/long_pairs.py gpu $c > /dev/null 2>> $out/pmc_long_pair.err
  done
done
cd $repo
python - $out $cases <<'PY'
import csv, glob, json, os, re, sys
out, cases = sys.argv[1], sys.argv[2:]
res = {"kernel_source_hash": open(os.path.join(out, "kernel_source_hash.txt")).read().strip(), "commit": os.environ.get("GNX_COMMIT", "unknown"),
       "how": "tools/pmc_long_pair.sh: rocprofv3 --kernel-trace --stats, --pmc <SQ group>, --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes) of tools/long_pairs.py gpu <case>; FETCH_SIZE doubled (gfx950: 64 B units counted as 32, MI355X_MICROARCH.md); per launch of the sweep kernel (mean over the case's calls)"}
txt = []
def short(n):
    m = re.search(r"::(\w+_kernel)(<[^>(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:60]
for c in cases:
    d = "/tmp/lp_" + c
    def rows(sub):
        r = []
        for path in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as fh:
                r += list(csv.DictReader(fh))
        return r
    sq = [r for r in rows("sq") if "sweep_kernel" in r["Kernel_Name"]]
    if not sq:
        continue
    kname = short(sq[0]["Kernel_Name"])
    v = {}
    for r in sq:
        v.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    v = {k: sum(x) / len(x) for k, x in v.items()}
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
    e = {"kernel": kname, "launches": len(sq) // max(len(v), 1), "valu_busy": round(v["SQ_ACTIVE_INST_VALU"] * 4 / cyc, 4), "waves_per_simd": round(v["SQ_WAVE_CYCLES"] * 4 / cyc, 3),
         "wait_any_frac_of_wave_cycles": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4), "valu_insts": v["SQ_INSTS_VALU"], "waves": v["SQ_WAVES"]}
    for k in ("FETCH_SIZE", "WRITE_SIZE"):
        x = [float(r["Counter_Value"]) for r in rows(k) if "sweep_kernel" in r["Kernel_Name"] and r["Counter_Name"] == k]
        e[k + "_KB"] = sum(x) / max(len(x), 1)
    e["hbm_bytes_per_launch"] = int((2 * e["FETCH_SIZE_KB"] + e["WRITE_SIZE_KB"]) * 1024)
    for path in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                if "sweep_kernel" in r["Name"]:
                    e["kernel_ms_avg"] = float(r["AverageNs"]) / 1e6; e["kernel_calls"] = int(r["Calls"])
        with open(path) as fh, open(os.path.join(out, "kernel_stats_long_pair_%s.csv" % c), "w") as o:
            wr = csv.writer(o)
            for k, row in enumerate(csv.reader(fh)):
                if k > 0:
                    row[0] = short(row[0])
                wr.writerow(row)
    try:
        e["row"] = json.loads(open(os.path.join(d, "row.jsonl")).read().strip().split("\n")[-1])
        e["row"] = {k: e["row"][k] for k in ("case", "n", "m", "cells", "sweep_ms", "walk_ms", "call_s", "rows_per_lane", "snapshot_steps", "ok") if k in e["row"]}
    except Exception:
        pass
    res[c] = e
    txt.append(json.dumps({"case": c, **e}))
json.dump(res, open(os.path.join(out, "pmc_long_pair.json"), "w"), indent=1)
open(os.path.join(out, "pmc_long_pair.txt"), "w").write("# " + res["how"] + "\n# valu_busy = SQ_ACTIVE_INST_VALU * 4 / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); waves_per_simd = SQ_WAVE_CYCLES * 4 / (same)\n" + "\n".join(txt) + "\n")
print("\n".join(txt))
PY
