import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from presto_amd import _lib, abi
from presto_amd.operators import HashBuilderOperator, LookupJoinOperator, LookupSourceFactory
from presto_amd.page import Block, DeviceBuffer, Page
torch.cuda.set_device(0); _lib.init(0)
def dev_block(type_, t):
    return Block(type_, abi.FLAT, t.numel(), values=DeviceBuffer(t.data_ptr(), t.numel() * t.element_size(), t))
g = torch.Generator(device="cuda").manual_seed(2)
for nb in (1 << 20, 15_000_000):
  for npr in (1 << 22, 1 << 26):
      #
      bkeys = torch.randperm(nb, device="cuda", generator=g).to(torch.int64) * 4
      pkeys = torch.randint(0, nb * 8, (npr,), dtype=torch.int64, device="cuda", generator=g)
      expect = int(((pkeys % 4 == 0) & (pkeys < nb * 4)).sum().item())
      bridge = LookupSourceFactory()
      t0 = time.perf_counter()
      bpay = torch.arange(nb, dtype=torch.int32, device='cuda'); b = HashBuilderOperator(bridge, [abi.BIGINT, abi.INTEGER], [0], [1], expected_positions=nb)
      b.addInput(Page([dev_block(abi.BIGINT, bkeys), dev_block(abi.INTEGER, bpay)], nb, abi.MEM_DEVICE)); t1 = time.perf_counter()
      b.finish(); torch.cuda.synchronize(); t2 = time.perf_counter()
      j = LookupJoinOperator(bridge, [abi.BIGINT], [0], [0], output_mem=abi.MEM_DEVICE)
      j.addInput(Page([dev_block(abi.BIGINT, pkeys)], npr, abi.MEM_DEVICE)); out = j.getOutput()
      got = out.position_count if out is not None else 0
      key, links = bridge.tables()
      print(nb, "addInput %.1f ms finish %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), "matches", got, "expected", expect, "occupied slots", int((key >= 0).sum()), "hash_size", len(key))
