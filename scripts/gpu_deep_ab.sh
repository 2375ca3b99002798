#!/bin/bash
# l = 3 emission, slices against one workgroup per unit (GNNPE_DEEP_EMIT=units|slices), same process settings otherwise:
# the config-5 graph at two chunk sizes and G(100K, 1M) e = 8 in full.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
C5="--vertices 4000000 --edges 64000000 --powerlaw --max-degree 3000 --embedding 8"
for mode in slices unit; do
  if [ $mode = unit ]; then export GNNPE_DEEP_EMIT=units; else export GNNPE_DEEP_EMIT=slices; fi
  for chunk in 16777216 67108864; do
    echo "== $mode config5 chunk $chunk"
    timeout -k 10 200 python3 scripts/bench_deep.py $C5 --max-paths 67108864 --chunk $chunk | tail -1 | cut -c1-700 || exit 1
  done
  echo "== $mode G(100K,1M) e=8"
  timeout -k 10 200 python3 scripts/bench_deep.py --embedding 8 | tail -1 | cut -c1-700 || exit 1
done
