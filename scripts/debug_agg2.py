import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("WITH_TORCH"):
    import torch; torch.cuda.set_device(0); _x = torch.empty(1, device="cuda")
from presto_amd import _lib, abi, tpch
from presto_amd.operators import HashAggregationOperator
_lib.init(0)
rows = 1 << 26
NCH = int(os.environ.get("NCH", "3"))
cols = [abi.L_QUANTITY, abi.L_EXTENDEDPRICE, abi.L_SHIPDATE][:NCH]
types = [abi.DOUBLE, abi.DOUBLE, abi.DATE][:NCH]
dev = tpch.DeviceColumns(cols, 100.0, rows)
pages = list(dev.pages(1 << 24))
for key, name in ((0, "quantity DOUBLE (50 groups) nch=%d" % NCH),):
    for rep in range(2):
        op = HashAggregationOperator(types, [key], [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)], output_mem=abi.MEM_DEVICE)
        t0 = time.perf_counter()
        for p in pages: op.addInput(p)
        op.finish(); out = op.getOutput(); dt = time.perf_counter() - t0
        print(name, "%.1f ms" % (dt * 1e3), "kernel", op.kernelTime(), "groups", out.position_count, flush=True)
        op.close()
