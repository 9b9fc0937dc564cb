import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from presto_amd import _lib, abi
from presto_amd.operators import HashAggregationOperator
from presto_amd.page import Block, DeviceBuffer, Page
torch.cuda.set_device(0); _lib.init(0)
rows = 1 << 26
g = torch.Generator(device="cuda").manual_seed(1)
vals = torch.rand(rows, dtype=torch.float64, device="cuda", generator=g)
if os.environ.get("VALS") == "ones":
    vals = torch.ones(rows, dtype=torch.float64, device="cuda")
elif os.environ.get("VALS") == "big":
    vals = vals * 1e5 + 900.0
for groups in [int(x) for x in sys.argv[1].split(",")]:
    keys = torch.randint(0, groups, (rows,), dtype=torch.int64, device="cuda", generator=g)
    mode = os.environ.get("KEYMODE", "int")
    ktype = abi.BIGINT
    if mode == "double":
        keys = keys.double(); ktype = abi.DOUBLE
    elif mode == "scaled":
        keys = keys * 1000003 + 77
    torch.cuda.synchronize()
    sub = [Page([Block(ktype, abi.FLAT, 1 << 24, values=DeviceBuffer(keys.data_ptr() + 8 * i, 8 << 24, keys)),
                 Block(abi.DOUBLE, abi.FLAT, 1 << 24, values=DeviceBuffer(vals.data_ptr() + 8 * i, 8 << 24, vals))], 1 << 24, abi.MEM_DEVICE)
           for i in range(0, rows, 1 << 24)]
    for rep in range(2):
        t0 = time.perf_counter()
        op = HashAggregationOperator([ktype, abi.DOUBLE], [0], [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)], expected_groups=groups, output_mem=abi.MEM_DEVICE)
        ts = [time.perf_counter() - t0]
        for p in sub:
            a = time.perf_counter(); op.addInput(p); ts.append(time.perf_counter() - a)
        a = time.perf_counter(); op.finish(); out = op.getOutput(); ts.append(time.perf_counter() - a)
        kt = op.kernelTime()
        a = time.perf_counter(); op.close(); ts.append(time.perf_counter() - a)
        print(groups, "rep", rep, " ".join("%.2f" % (x * 1e3) for x in ts), "ms | kernel", kt, "groups", out.position_count, flush=True)
