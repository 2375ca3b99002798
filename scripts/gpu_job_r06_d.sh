cd $GRAFT_REPO_ROOT
bash scripts/gpu_profile_r06.sh r06
python3 scripts/emit_pmc_json.py gpurun_out/r06_emit_pmc.txt gpurun_out/r06_pmc_fill.json
cp profiles/r06_kernel_stats.csv gpurun_out/r06_kernel_stats.csv
