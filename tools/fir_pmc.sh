#!/bin/bash
# dev: PMC passes over the noise-filter kernels (tools/time_noise_fir.py)
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_SMEM"; do
  i=$((i+1))
  bash $R/tools/prof_pmc.sh $R/gpurun_out/fir_pmc_$i $set -- python $R/tools/time_noise_fir.py > $R/gpurun_out/fir_pmc_$i.log 2>&1
done
cd $R && python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/fir_pmc_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[-40:]
            if "golf::" in k:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            res[k][c] = sum(v) / len(v)
for k, d in res.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} {v:14.0f}")
PY
