#!/bin/bash
# usage: tools/osc_pmc2.sh <tag> [B]   (env GOLF_OSCF_GEOM / GOLF_OSC_UNFUSED select the variant)
R=$GRAFT_REPO_ROOT; tag=$1; B=${2:-256}
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  bash $R/tools/prof_pmc.sh $R/gpurun_out/pmc_${tag}_$i $set -- python $R/tools/osc_case.py $B > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
done
cd $R && python - "$tag" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[-60:]
            if "golf::" in k:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            res[k][c] = sum(v) / len(v)
for k, d in res.items():
    print(tag, k)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} {v:14.0f}")
PY
