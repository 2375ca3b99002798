"""CPU parity oracle for the GNN-PE offline path -- TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package; the product (``gnn-pe_amd/``) never does.  See ``gnnpe_oracle.h``.
"""
from .binding import (Oracle, bitmap_to_sets, build_oracle, ref_dump_path, ref_main_path, ref_main_pge_path,  # noqa: F401
                      ref_online_path)
