"""HashBuilder over 8 M rows with about five rows per key (BenchmarkHashBuildAndJoinOperators' duplicate build rows): for a kernel trace."""
import sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from presto_amd import abi
from presto_amd.operators import HashBuilderOperator, LookupSourceFactory
from presto_amd.page import Block, DeviceBuffer, Page

g = torch.Generator(device="cuda").manual_seed(3)
nb = 8_000_000
keys = torch.randint(0, nb // 5, (nb,), dtype=torch.int64, device="cuda", generator=g)
pay = torch.arange(nb, dtype=torch.int32, device="cuda")
page = Page([Block(abi.BIGINT, abi.FLAT, nb, values=DeviceBuffer(keys.data_ptr(), 8 * nb, keys)),
             Block(abi.INTEGER, abi.FLAT, nb, values=DeviceBuffer(pay.data_ptr(), 4 * nb, pay))], nb, abi.MEM_DEVICE)
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bridge = LookupSourceFactory()
    b = HashBuilderOperator(bridge, [abi.BIGINT, abi.INTEGER], [0], [1], expected_positions=nb)
    b.addInput(page); b.finish()
    torch.cuda.synchronize()
    print("build %.3f ms" % ((time.perf_counter() - t0) * 1e3))
    b.close()
