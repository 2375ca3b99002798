"""profiles/<round>_pmc_fill.json from the output of scripts/emit_pmc.sh (per-launch means of the fabric counters of both emit
kernels): traffic per launch = read requests by size + write requests by size (64-byte ones counted, the rest are 32-byte).
usage: python scripts/emit_pmc_json.py gpurun_out/r04_emit_pmc.txt profiles/r04_pmc_fill.json"""
import json, sys
src, dst = sys.argv[1], sys.argv[2]
per = {}
for ln in open(src):
    f = ln.split()
    if len(f) >= 4 and f[1] in ("k_fill_ranked", "k_fill_tiles", "k_fill_tickets", "k_fill_tile_jobs") and f[2] == "mean":
        per.setdefault(f[1], {})[f[0]] = float(f[3])
out = {"command": "scripts/emit_pmc3.sh (round 5; round 4: emit_pmc.sh) = rocprofv3 --pmc <set> -- python3 scripts/emit_ab3.py --quick (one pass per counter "
                  "set; BASELINE config 3, 2.0e8 paths, the emit kernels into the same buffers in one process; k_fill_ranked with its start "
                  "vertices from ticket counters, five workgroups per CU; round 6, r06_pmc_fill.json: the one-shot launch, one wave per start vertex in launch order)",
       "algorithmic_bytes": 200031576 * 92}
for k, c in per.items():
    rd = c.get("TCC_EA0_RDREQ_128B_sum", 0) * 128 + c.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 + c.get("TCC_EA0_RDREQ_32B_sum", 0) * 32
    w64 = c.get("TCC_EA0_WRREQ_64B_sum", 0)
    wr = w64 * 64 + (c.get("TCC_EA0_WRREQ_sum", 0) - w64) * 32
    out[k] = {"counters": c, "derived": {"read_bytes": rd, "write_bytes": wr, "traffic_bytes": rd + wr,
                                         "traffic_over_algorithmic": (rd + wr) / out["algorithmic_bytes"]}}
json.dump(out, open(dst, "w"), indent=1)
print({k: v["derived"] for k, v in out.items() if isinstance(v, dict)})
