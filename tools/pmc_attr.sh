#!/bin/bash
# Which unit is busy while the vector ALU idles: PMC passes over one batched Farneback call (tools/ab_iter.py), counters picked from
# what `rocprofv3 -L` lists on this box (names that do not exist are dropped instead of failing the pass), small sets per pass.
# usage: pmc_attr.sh [--size WxH] [--batch N] [--opts "k=v,..."] [--tag T] [--sets "1 2 6"  (line numbers of the set list; default all)]      summary -> gpurun_out/pmc_attr/<tag>.txt
SIZE=1920x1080; BATCH=8; OPTS=""; TAG=default; SETS=""
while [ -n "$1" ]; do
  case "$1" in --size) SIZE=$2; shift 2;; --batch) BATCH=$2; shift 2;; --opts) OPTS=$2; shift 2;; --tag) TAG=$2; shift 2;; --sets) SETS=$2; shift 2;; *) break;; esac
done
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/pmc_attr_$TAG; rm -rf $RAW; mkdir -p $RAW
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_attr; mkdir -p $OUT
[ -s $OUT/avail.txt ] || rocprofv3 -L > $OUT/avail.txt 2>&1
python - "$OUT/avail.txt" > $RAW/sets.txt <<'PY'
import re, sys
avail = set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", open(sys.argv[1]).read()))
want = [
 ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"],
 ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_MISC"],
 ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "SQ_INST_CYCLES_VMEM", "SQ_INST_LEVEL_VMEM", "SQ_INSTS_SALU", "SQ_INSTS_LDS"],
 ["SQ_INST_LEVEL_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INST_CYCLES_SALU", "SQ_THREAD_CYCLES_VALU", "SQ_IFETCH", "SQ_IFETCH_LEVEL"],
 ["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_DCACHE_REQ", "SQC_DCACHE_MISSES"],
 ["TA_TA_BUSY_sum", "TA_BUSY_avr", "TA_BUSY_max", "TA_BUSY_min"],
 ["TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_ADDR_STALLED_BY_TD_CYCLES_sum"],
 ["TA_BUFFER_WAVEFRONTS_sum", "TA_BUFFER_READ_WAVEFRONTS_sum", "TA_BUFFER_WRITE_WAVEFRONTS_sum"],
 ["TA_BUFFER_TOTAL_CYCLES_sum", "TA_BUFFER_COALESCED_READ_CYCLES_sum", "TA_BUFFER_COALESCED_WRITE_CYCLES_sum"],
 ["TD_TD_BUSY_sum", "TD_TC_STALL_sum", "TD_LOAD_WAVEFRONT_sum", "TD_COALESCABLE_WAVEFRONT_sum", "TD_SPI_STALL_sum"],
 ["TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum", "TCP_TD_TCP_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum"],
 ["TCP_PENDING_STALL_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TA_TCP_STATE_READ_sum"],
 ["TCP_TOTAL_ACCESSES_sum", "TCP_TOTAL_READ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"],
 ["TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_VOLATILE_sum"],
 ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum"],
 ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"],
 ["TCC_TAG_STALL_sum", "TCC_EA0_RDREQ_LEVEL_sum", "TCC_BUSY_sum", "TCC_EA0_RD_UNCACHED_32B_sum"],
]
for s in want:
    have = [c for c in s if c in avail]
    if have:
        print(" ".join(have))
PY
cat $RAW/sets.txt > $OUT/${TAG}_sets.txt
i=0
while read -r set; do
  i=$((i+1))
  if [ -n "$SETS" ] && ! echo " $SETS " | grep -q " $i "; then continue; fi
  timeout 240 rocprofv3 --pmc $set --output-format csv -d $RAW/p$i -o p -- python $GRAFT_REPO_ROOT/tools/ab_iter.py --size $SIZE --batch $BATCH --calls 1 "$OPTS" > $RAW/p$i.log 2>&1 || { echo "pass $i ($set) failed/timeout"; tail -3 $RAW/p$i.log; }
done < $RAW/sets.txt
python - "$RAW" "$OUT/${TAG}.txt" <<'PY'
import csv, glob, collections, re, sys
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(sys.argv[1] + "/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ofxcv_fb::", "").replace("ofxcv_fb::", ""))
        if "iterate" not in name:
            continue
        key = (name[:48], r["Grid_Size"], r["Counter_Name"])
        agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
with open(sys.argv[2], "w") as fo:
    for k, (v, n) in sorted(agg.items()):
        fo.write("%-48s grid %-9s %-40s per-launch %16.1f (n=%d)\n" % (k[0], k[1], k[2], v / max(n, 1), n))
PY
grep "iterate_col_kernel<1, 1" $OUT/${TAG}.txt | head -120
