#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- bash tools/refresh_profiles.sh): every bench line, rocprofv3 kernel traces and the
# separate HBM-counter passes that profiles/ is built from (tools/collect_profiles.py turns the output into profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}
RND=${ROUND_TAG:-r06}
O=$R/gpurun_out/$RND
rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
b() { python bench.py "$@" 2>/dev/null | tail -1; }
b > $O/bench_golf_ss_synth.json
b --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_golf_ss_synth_driver_cmd.json     # the driver's command line
b --streams 1 --no-graphs --no-cpu-baseline > $O/bench_single_stream.json
# batch sweep of the whole synthesis step, 4 batches in flight like the headline (round 5: rounds 2 - 4 ran these on ONE stream,
# which is what made B = 2048 look slower than B = 256: the serial filter leaves most of the chip to the other batches)
b --batch 256 --no-cpu-baseline --steps 48 --warmup 8 --settle 2 --refresh-inputs 0 --recipe-stream 0 > $O/bench_b256.json
b --batch 1024 --no-cpu-baseline --steps 24 --warmup 8 --settle 2 --refresh-inputs 0 --recipe-stream 0 > $O/bench_b1024.json
b --batch 2048 --steps 16 --warmup 4 --repeats 3 --prereplay 2 --settle 1 --no-cpu-baseline --refresh-inputs 0 --recipe-stream 0 > $O/bench_b2048.json
b --batch 8192 --steps 12 --warmup 4 --repeats 3 --prereplay 2 --settle 1 --no-cpu-baseline --refresh-inputs 0 --recipe-stream 0 > $O/bench_b8192.json
b --batch 16384 --steps 8 --warmup 4 --repeats 3 --prereplay 1 --settle 1 --no-cpu-baseline --refresh-inputs 0 --recipe-stream 0 > $O/bench_b16384.json
b --batch 2048 --streams 1 --steps 10 --warmup 2 --repeats 3 --prereplay 2 --settle 1 --no-cpu-baseline > $O/bench_b2048_1stream.json
b --batch 16384 --streams 1 --steps 6 --warmup 2 --repeats 3 --prereplay 1 --settle 1 --no-cpu-baseline > $O/bench_b16384_1stream.json
b --batch 8192 --workload lpc-ss-fast --streams 1 --steps 10 --warmup 2 --repeats 3 --prereplay 2 --no-cpu-baseline > $O/bench_b8192_lpc_only.json
b --batch 16384 --workload lpc-ss-fast --streams 1 --steps 6 --warmup 2 --repeats 3 --prereplay 1 --no-cpu-baseline > $O/bench_b16384_lpc_only.json
b --batch 2048 --workload golf-ss-train --streams 1 --steps 6 --warmup 2 --repeats 3 --prereplay 1 --no-cpu-baseline > $O/bench_b2048_train.json
b --workload golf-ss-train --no-cpu-baseline --steps 100 > $O/bench_train.json
b --workload golf-ff-synth --no-cpu-baseline > $O/bench_ff.json
b --workload golf-ff-train --no-cpu-baseline --steps 100 > $O/bench_ff_train.json
b --workload golf-ss-decoder --no-cpu-baseline > $O/bench_decoder.json
b --workload golf-ss-decoder-train --no-cpu-baseline --steps 100 > $O/bench_decoder_train.json
b --workload ddsp-decoder --no-cpu-baseline > $O/bench_ddsp_decoder.json
b --workload golf-ss-decoder-logits --no-cpu-baseline > $O/bench_decoder_logits.json
b --fp64-transitions --no-cpu-baseline > $O/bench_fp64_transitions.json
b --workload golf-ss-train-step --batch 64 --steps 20 --warmup 5 > $O/bench_train_step.json
python tools/train_step_profile.py 64 > $O/train_step_profile.json 2>/dev/null
python tools/recipe_latency.py 64 2>/dev/null | grep seed > $O/recipe_latency.txt   # one batch alone, per recipe seed, with its conditioning words
b --batch 16384 --workload osc-only --streams 1 --steps 6 --warmup 2 --repeats 3 --prereplay 1 --no-cpu-baseline > $O/bench_b16384_osc_only.json
prof() { out=$1; shift; (cd /tmp && rocprofv3 --kernel-trace -d $O/$out -- python $R/bench.py --no-cpu-baseline --steps 50 --warmup 10 "$@" > $O/$out.log 2>&1); }
prof trace_synth
prof trace_synth_single --streams 1 --no-graphs
prof trace_train --workload golf-ss-train
prof trace_decoder --workload golf-ss-decoder --streams 1 --no-graphs
prof trace_decoder_train --workload golf-ss-decoder-train
prof trace_ff_train --workload golf-ff-train
for c in FETCH_SIZE WRITE_SIZE; do
  bash tools/prof_pmc.sh $O/pmc_${c}_synth $c -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --workload golf-ss-synth --streams 1 --no-graphs > $O/pmc_${c}_synth.log 2>&1
  # ... and of the throughput chain the headline runs (GOLF_SS_THROUGHPUT), eager on one stream like the pass above
  bash tools/prof_pmc.sh $O/pmc_${c}_synthtp $c -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --workload golf-ss-synth --streams 1 --no-graphs --lpc-chain throughput > $O/pmc_${c}_synthtp.log 2>&1
  bash tools/prof_pmc.sh $O/pmc_${c}_decoder $c -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --workload golf-ss-decoder --streams 1 --no-graphs > $O/pmc_${c}_decoder.log 2>&1
  bash tools/prof_pmc.sh $O/pmc_${c}_train $c -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --workload golf-ss-decoder-train > $O/pmc_${c}_train.log 2>&1
done
# SQ counters (separate passes, kernel-trace only) on the eager single-stream decoder and training steps
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  bash tools/prof_pmc.sh $O/sq_${i}_decoder $set -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --workload golf-ss-decoder --streams 1 --no-graphs > $O/sq_${i}_decoder.log 2>&1
  bash tools/prof_pmc.sh $O/sq_${i}_train $set -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --workload golf-ss-decoder-train --streams 1 --no-graphs > $O/sq_${i}_train.log 2>&1
done
# the headline run itself (4 batches in flight, hipGraph replay): issued VALU wave-instructions per kernel launch, for
# roofline.valu_issue_frac (VERDICT r3 #1b); same command as the driver's, counters only (no timing is read from this pass)
bash tools/prof_pmc.sh $O/sq4_synth SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES -- python $R/bench.py --no-cpu-baseline --recipe-stream 0 --steps 20 --warmup 5 --headline-only > $O/sq4_synth.log 2>&1
b --lpc-chain latency --no-cpu-baseline --recipe-stream 0 > $O/bench_golf_ss_synth_latency_chain.json
b --workload golf-ss-synth-have-maps --no-cpu-baseline --recipe-stream 0 > $O/bench_synth_have_maps.json
b --workload osc-only --no-cpu-baseline > $O/bench_osc_only.json
# the oscillator's own counters (LDS bank conflicts, instruction counts) at a saturating batch
bash tools/osc_pmc2.sh osc 2048 > $O/osc_pmc_b2048.txt 2>&1; rm -rf $R/gpurun_out/pmc_osc_[0-9] $R/gpurun_out/pmc_osc_*.log
b --workload lpc-ss-fast --no-cpu-baseline > $O/bench_lpc_only.json
# keep the merge-back small: summarise the rocpd databases here, drop them and the per-dispatch traces of the counter passes
for d in $O/trace_*; do
  [ -d "$d" ] || continue
  db=$(find $d -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db $O/$(basename $d | sed 's/trace_//')_kernel_stats.csv
  rm -rf $d
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
# the counter passes write one row per dispatch and counter (100 MB over a refresh: gpurun merges back at most 64 MiB): keep the
# per-kernel means (tools/collect_profiles.py averages them anyway) and the dispatch counts
python - $O <<'PY'
import collections, csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(f, "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Dispatches"])
        for (k, c), v in acc.items():
            w.writerow([k, c, sum(v) / len(v), len(v)])
PY
find $O -name "*.db" -delete
ls -R $O | head -60
du -sh $O
