#!/bin/bash
# lat_trace.sh <n> -- rocprofv3 kernel trace of one-call-at-a-time pairing batches of n items on the lane-cooperative path: kernel durations and the gaps between the launches of one call
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
N=${1:-16384}
OUT=$GRAFT_REPO_ROOT/gpurun_out/lat_trace_$N
mkdir -p $OUT
export TMPDIR=/tmp LAT_SIZES=$N
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tr -- python3 $GRAFT_REPO_ROOT/tools/exp/lat_ab.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "k_cvm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-7:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = None
tot = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{r['Kernel_Name'][:40]:40s} grid {r.get('Grid_Size','?'):>8s} lds {r.get('LDS_Block_Size','?'):>7s}  start +{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:9.1f} us  gap {gap:7.1f} us")
    prev_end = e
    tot += e - s
print("sum of kernel durations", tot / 1e3, "us; first start to last end", (int(last[-1]["End_Timestamp"]) - t0) / 1e3, "us")
PY
