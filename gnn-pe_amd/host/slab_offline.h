// slab_offline.h -- multi-GPU `-m offline` of gnnpe_main (host/slab_offline.cpp).
#pragma once

#include <vector>

#include "cli_common.h"
#include "graph_loader.h"

namespace slab {

// Vertex-partitioned offline step over o.gpus devices: slab rows per device, halo all-to-all-v (RCCL or device copies),
// concurrent emit / render / pwrite, index.dat per partition.  Writes the reference's files; returns 0.
int run_offline_slabs(const cli::Options &o, const gnnpe_host::StaticGraph &g, const std::vector<uint32_t> &sorted_nodes,
                      const std::vector<uint32_t> &membership, const std::vector<double> &label_table,
                      Clock::time_point t_start, Clock::time_point t_loaded);

}  // namespace slab
