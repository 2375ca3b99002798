#!/bin/bash
# Kernel trace of `gnnpe_main -m offline --index` on the reference's sample graph (max degree 168: one hub row):
# evidence that a mixed-degree graph runs the ranked emit kernel (hub pairs in-line) and which index path it takes.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=${1:-r02}
W=$(mktemp -d)
mkdir -p $W/gnn-pe/partitions/partition-0 gpurun_out
python3 - "$W" <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
import gnnpe_amd
from gnnpe_amd import synth
g = "tests/golden/test_graph/data_graph.graph"
deg = np.array([int(l.split()[3]) for l in open(g) if l.startswith("v")])
synth.write_membership(sys.argv[1] + "/gnn-pe/membership.txt", np.argsort(deg, kind="stable").astype(np.uint32), np.zeros(len(deg), np.uint32))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_testgraph_trace -- ./gnn-pe_amd/gnnpe_main -f $W/ -d tests/golden/test_graph/data_graph.graph -p 1 --index > gpurun_out/${R}_testgraph.log 2>&1
echo "rc=$? header=$(head -1 $W/gnn-pe/all_paths.txt)"
rm -rf $W
