#!/bin/bash
# Counters of the l = 3 emission (k_deep3_slices) and count on the config-5 graph, one 2^26-path range: two rocprofv3 --pmc passes over bench_deep.py
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/deep_pmc_$i
  timeout -k 10 250 rocprofv3 --pmc $set --output-format csv -d gpurun_out/deep_pmc_$i -- python3 scripts/bench_deep.py --vertices 4000000 --edges 64000000 --powerlaw --max-degree 3000 --embedding 8 --max-paths 67108864 --chunk 67108864 > gpurun_out/deep_pmc_$i.log 2>&1
  echo "set $i rc=$?"
  python3 - "$i" <<'PY'
import csv, glob, sys
i = sys.argv[1]
for f in glob.glob(f"gpurun_out/deep_pmc_{i}/*/*_counter_collection.csv"):
    per = {}
    for r in csv.DictReader(open(f)):
        for kn in ("k_deep3_slices<8, true", "k_deep3_slices<8, false", "k_deep3_count_rows"):
            if kn in r["Kernel_Name"]:
                per.setdefault((kn, r["Counter_Name"]), {}).setdefault(int(r["Dispatch_Id"]), 0.0)
                per[(kn, r["Counter_Name"])][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (kn, c), v in sorted(per.items()):
        vals = list(v.values())
        print(f"{kn:26s} {c:36s} mean {sum(vals)/len(vals):.5g} over {len(vals)} launches")
PY
done
