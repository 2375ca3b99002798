#!/bin/bash
# Backtrace of `gnnpe_main --gpus 2 --same-device` under rocgdb (batch mode): where does an uncaught exception come from?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
W=$(mktemp -d)
python3 - "$W" <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
import gnnpe_amd
from gnnpe_amd import synth
g = synth.gnm_graph(3000, 21000, n_labels=9, seed=17)
sn = synth.degree_order(g["offsets"])
synth.write_graph_file(sys.argv[1] + "/g.graph", g)
synth.make_dataset_dir(sys.argv[1], 4)
synth.write_membership(sys.argv[1] + "/gnn-pe/membership.txt", sn, synth.block_membership(g["n"], 4))
PY
rocgdb -batch -ex "catch throw" -ex run -ex "bt 25" -ex "info threads" --args ./gnn-pe_amd/gnnpe_main -f $W/ -d $W/g.graph -p 4 --gpus 2 --same-device --chunk 50000 --index --timing 2>&1 | tail -60
rm -rf $W
