#!/bin/bash
# A/B of SQ counters for one kernel between ENVIRONMENTS (same build).  Usage (GPU box, repo root):
#   bash tools/pmc_env_ab.sh <out dir> "<bench.py args>" <kernel regex> "TAG1:VAR=VAL VAR2=VAL" "TAG2:..." ...
out=$1; args=$2; kre=$3; shift 3
repo=$PWD; out=$repo/$out; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
G1="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES"
G2="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}
  g=0
  for grp in "$G1" "$G2"; do g=$((g+1))
    rm -rf /tmp/pmc_$tag$g
    env $envs timeout 600 rocprofv3 --pmc $grp -d /tmp/pmc_$tag$g -o pmc --output-format csv -- python $repo/bench.py $args > /dev/null 2>> $out/pmc.err
  done
  python - "$tag" "$kre" >> $out/pmc_ab.txt <<'PY'
import csv, glob, re, sys
tag, kre = sys.argv[1], sys.argv[2]
val, ms = {}, []
for g in (1, 2):
    rows = []
    for path in glob.glob("/tmp/pmc_%s%d/**/*counter_collection.csv" % (tag, g), recursive=True):
        with open(path) as fh:
            rows += [r for r in csv.DictReader(fh) if re.search(kre, r["Kernel_Name"])]
    if not rows:
        continue
    big = max(int(r["Grid_Size"]) for r in rows)
    for r in rows:
        if int(r["Grid_Size"]) == big:
            val.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            ms.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
v = {k: sum(x) / len(x) for k, x in val.items()}
print("== %s: kernel %.2f ms (under the profiler)" % (tag, sum(ms) / max(len(ms), 1)))
for k in sorted(v):
    print("   %-22s %.4g" % (k, v[k]))
if "GRBM_GUI_ACTIVE" in v:
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
    print("   valu_busy %.3f  waves_per_simd %.2f  wait_any/wave_cycles %.3f  wait_inst_any/wave_cycles %.3f  active_valu/wave_cycles %.3f  lds_conflict/lds_idx_active %.3f" % (
        v.get("SQ_ACTIVE_INST_VALU", 0) * 4 / cyc, v.get("SQ_WAVE_CYCLES", 0) * 4 / cyc, v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1),
        v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1), v.get("SQ_ACTIVE_INST_VALU", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1),
        v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
done
cat $out/pmc_ab.txt
