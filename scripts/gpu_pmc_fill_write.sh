#!/bin/bash
# The write-side counter pass of the fill kernel alone (the round profile's set 2 needs more than its 240 s under --pmc)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=${1:-r03}
rm -rf gpurun_out/${R}_pmc_2
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d gpurun_out/${R}_pmc_2 -- python3 bench.py --steps 1 --warmup 0 --placements 1 --no-cpu-baseline --no-index --no-config5 > gpurun_out/${R}_pmc_2.log 2>&1
echo "pmc write set rc=$?"
