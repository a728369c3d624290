"""Does a small kernel on a second stream run WHILE the 65 536-gate blind rotation occupies every CU?
(Measured r01: a 16-gate elementwise batch completes in 0.05 ms during the big batch, i.e. yes for kernels
that fit beside the 154 KiB-LDS / 2x216-VGPR workgroup; a 16-NAND batch, whose low-latency kernel needs
139 KiB of LDS, waits for a round boundary: 34 ms instead of 6.6.)  DESIGN.md section 9."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from iyokan_amd import client, hip
from iyokan_amd.params import OPS, params_128bit
p = params_128bit(); keys = client.keygen(p, seed=1); hip.initialize(keys, device_ids=(0,))
G = 65536
rng = np.random.default_rng(0)
bits = rng.integers(0, 2, size=2*G).astype(np.uint8)
arena_t = torch.zeros((3*G + 64, p.n+1), dtype=torch.int32, device="cuda")
arena_t[:2*G].copy_(torch.from_numpy(client.encrypt_bits(keys, bits, seed=2).view(np.int32)))
torch.cuda.synchronize()
arena = hip.Arena.from_torch(arena_t)
A = hip.Stream(0); B = hip.Stream(0)
idx = np.arange(G, dtype=np.int32)
big = (np.full(G, OPS["NAND"], np.int32), idx, idx+G, np.full(G, -1, np.int32), idx+2*G)
n = 16
small = (np.full(n, OPS["NOT"], np.int32), np.arange(n, dtype=np.int32), np.full(n, -1, np.int32), np.full(n, -1, np.int32), 3*G + np.arange(n, dtype=np.int32))
smallks = (np.full(n, OPS["NAND"], np.int32), np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32)+n, np.full(n, -1, np.int32), 3*G + 32 + np.arange(n, dtype=np.int32))
A.gate_batch(arena, *big); A.sync()
for name, job in (("elementwise NOT x16", small), ("NAND x16 (lat2 BR + KS)", smallks)):
    t0 = time.perf_counter(); B.gate_batch(arena, *job); B.sync(); alone = time.perf_counter() - t0
    A.gate_batch(arena, *big); time.sleep(0.1)
    t0 = time.perf_counter(); B.gate_batch(arena, *job); B.sync(); during = time.perf_counter() - t0
    t1 = time.perf_counter(); A.sync(); rest = time.perf_counter() - t1
    print(f"{name}: alone {alone*1e3:.2f} ms, while the big batch runs {during*1e3:.2f} ms (big batch had {rest*1e3:.0f} ms left)")
