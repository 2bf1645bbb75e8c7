#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/emit_p1 -- python3 $R/profiles/vgd_stream_probe.py > /dev/null 2> $OUT/emit_p1.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/emit_p2 -- python3 $R/profiles/vgd_stream_probe.py > /dev/null 2> $OUT/emit_p2.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/emit_p3 -- python3 $R/profiles/vgd_stream_probe.py > /dev/null 2> $OUT/emit_p3.err
cd $R
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for d in ("emit_p1", "emit_p2", "emit_p3"):
    for fn in glob.glob(f"{out}/{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0]
            if "voxel" in k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in acc:
    print(k, {c: round(v / len(cnt[k][c]), 1) for c, v in acc[k].items()}, "dispatches", max(len(v) for v in cnt[k].values()))
PY
rm -rf $OUT/emit_p1 $OUT/emit_p2 $OUT/emit_p3
