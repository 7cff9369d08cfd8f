#!/bin/bash
# which hardware queue every launch of a 1/8-shard pass ran on (rocprofv3 kernel trace), to see who serialises behind whom
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6q; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/shard_timeline.py 8 > $OUT/tl.txt 2>&1
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $OUT/queues.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-260:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"]
    short = n[:n.index("(")] if "(" in n else n
    short = short.replace("swk::", "").replace("void ", "")[:64]
    print("%-64s queue %-3s stream %-3s grid %7s wg %4s  %9.1f .. %9.1f us (%8.1f)" % (short, r.get("Queue_Id"), r.get("Stream_Id", "?"), r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")),
          (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf $OUT/trace
tail -5 $OUT/tl.txt; wc -l $OUT/queues.txt
