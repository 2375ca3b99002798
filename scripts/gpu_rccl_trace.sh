#!/bin/bash
# Kernel trace of `gnnpe_main --gpus 1 --transport rccl --index`: a 1-rank communicator whose halo / vde / tuple
# exchanges go through librccl (self ncclSend/ncclRecv).  The trace must show an RCCL device kernel next to ours.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=${1:-r03}
W=$(mktemp -d)
mkdir -p gpurun_out
python3 - "$W" <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
import gnnpe_amd
from gnnpe_amd import synth
g = synth.gnm_graph(100_000, 1_000_000)
sn = synth.degree_order(g["offsets"])
synth.write_graph_file(sys.argv[1] + "/g.graph", g)
synth.make_dataset_dir(sys.argv[1], 4)
synth.write_membership(sys.argv[1] + "/gnn-pe/membership.txt", sn, synth.block_membership(g["n"], 4))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_rccl_trace -- ./gnn-pe_amd/gnnpe_main -f $W/ -d $W/g.graph -p 4 --gpus 1 --transport rccl --index --timing > gpurun_out/${R}_rccl.log 2>&1
echo "rc=$? header=$(head -1 $W/gnn-pe/all_paths.txt)"
tail -2 gpurun_out/${R}_rccl.log
rm -rf $W
