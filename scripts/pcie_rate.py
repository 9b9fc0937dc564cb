"""PCIe-inclusive rate of the Q6 pipeline: pages handed over as PA_MEM_HOST buffers (what a JNI shim passes) instead
of device-resident columns.  Reported in DESIGN.md; never the bench `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from presto_amd import _lib, abi, tpch
from presto_amd.operators import FusedAggregationOperator, download_page
_lib.init(0)
rows = 1 << 25
dev = tpch.DeviceColumns(tpch.Q6_COLUMNS, 10.0, rows)
host = download_page(dev.page(0, rows))          # pageable numpy buffers
pages = [host.get_region(i, 1 << 22) for i in range(0, rows, 1 << 22)]
for it in range(3):
    op = FusedAggregationOperator(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), [], tpch.Q6_AGGREGATES)
    t0 = time.perf_counter()
    for p in pages:
        op.addInput(p)
    op.finish()
    out = op.getOutput().to_rows()
    dt = time.perf_counter() - t0
    op.close()
    print("host pages (pageable, 4Mi-row pages): %.3g rows/s = %.1f GB/s over PCIe, result %r" % (rows / dt, rows * 28 / dt / 1e9, out))
