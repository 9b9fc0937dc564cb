import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.set_device(0)
from presto_amd import _lib, abi, tpch
from presto_amd.operators import HashAggregationOperator
from presto_amd.page import Block, DeviceBuffer, Page
_lib.init(0)
rows = 1 << 26
mode = sys.argv[1]
if os.environ.get("PREWARM"):
    # allocate the operator-side buffers (tables, spill lists) BEFORE torch allocates anything
    small = tpch.DeviceColumns([abi.L_QUANTITY, abi.L_EXTENDEDPRICE], 100.0, 1 << 24)
    op = HashAggregationOperator([abi.DOUBLE, abi.DOUBLE], [0], [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)], output_mem=abi.MEM_DEVICE)
    t0 = time.perf_counter()
    op.addInput(small.page()); op.finish(); out = op.getOutput()
    print("prewarm %.1f ms" % ((time.perf_counter() - t0) * 1e3), op.kernelTime(), flush=True)
    op.close()
g = torch.Generator(device="cuda").manual_seed(1)
if mode == "tpch_in_torch":
    alloc = lambda nbytes: torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dev = tpch.DeviceColumns([abi.L_QUANTITY, abi.L_EXTENDEDPRICE], 100.0, rows, allocator=alloc)
    pages = list(dev.pages(1 << 24))
elif mode == "torch_in_pa":
    keys = torch.randint(0, 64, (rows,), dtype=torch.int64, device="cuda", generator=g).double()
    vals = torch.rand(rows, dtype=torch.float64, device="cuda", generator=g)
    torch.cuda.synchronize()
    from presto_amd._lib import DeviceAllocation, check, lib
    ka, va = DeviceAllocation(8 * rows), DeviceAllocation(8 * rows)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(ctypes.c_void_p(ka.ptr), ctypes.c_void_p(keys.data_ptr()), ctypes.c_size_t(8 * rows), 3)
    hip.hipMemcpy(ctypes.c_void_p(va.ptr), ctypes.c_void_p(vals.data_ptr()), ctypes.c_size_t(8 * rows), 3)
    hip.hipDeviceSynchronize()
    pages = [Page([Block(abi.DOUBLE, abi.FLAT, 1 << 24, values=DeviceBuffer(ka.ptr + 8 * i, 8 << 24, ka)),
                   Block(abi.DOUBLE, abi.FLAT, 1 << 24, values=DeviceBuffer(va.ptr + 8 * i, 8 << 24, va))], 1 << 24, abi.MEM_DEVICE) for i in range(0, rows, 1 << 24)]
for rep in range(2):
    op = HashAggregationOperator([abi.DOUBLE, abi.DOUBLE], [0], [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)], output_mem=abi.MEM_DEVICE)
    t0 = time.perf_counter()
    for p in pages: op.addInput(p)
    op.finish(); out = op.getOutput(); dt = time.perf_counter() - t0
    print(mode, "%.1f ms" % (dt * 1e3), "kernel", op.kernelTime(), "groups", out.position_count, flush=True)
    op.close()
