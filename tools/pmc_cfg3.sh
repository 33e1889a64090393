# Counters of the long-filter convolution (cfg3 = BASELINE configs[2]: STFTMaskedNoiseReverb, 512 x 2 x 240000, 60001 taps):
#   bash tools/pmc_cfg3.sh        -> gpurun_out/pmc_cfg3/summary.txt
# One rocprofv3 pass per counter set (separate --pmc passes with --kernel-trace only, as the microarchitecture guide asks).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_cfg3
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_BUSY_sum TCC_TAG_STALL_sum" \
           "TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_COALESCED_READ_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py --config cfg3 --steps 2 --warmup 1 --no-cpu-baseline ${CFG3_EXTRA} > $OUT/p$i.log 2>&1
done
python3 - > $OUT/summary.txt <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", re.sub(r"^void\s+", "", r["Kernel_Name"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", re.sub(r"^void\s+", "", r["Kernel_Name"]))
        if re.search("xspec|macinv|istft|stft_ir|ir_energy|hspec", k):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    d = dur.get(k, [])
    big = [x for x in d if x >= 0.5 * max(d)] if d else []
    print("==", k, f"launches={len(d)} ms(mean of the large ones)={sum(big)/max(len(big),1):.3f}")
    for c in sorted(agg[k]):
        v = agg[k][c]
        top = max(v)
        v = [x for x in v if x >= 0.5 * top] or v      # drop the small warm-up launches
        print(f"  {c:32s} n={len(v):2d} mean={sum(v)/len(v):.5g}")
PY
cat $OUT/summary.txt
