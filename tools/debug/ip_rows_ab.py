#!/usr/bin/env python3
"""A/B of the small-N InnerProduct kernel (cls_pred / bbox_pred of the 7s nets: R x 4096 -> 2 / 8): 4 against 8 rows of x per workgroup
(mscnn_debug_inner_product_rows), interleaved, HIP events around 200 back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mscnn_amd import hipapi as hip

for M, N, K in [(676, 8, 4096), (676, 2, 4096), (300, 8, 4096), (1352, 8, 4096), (676, 20, 4096)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.relu(torch.randn((M, K), device="cuda", generator=g)); w = torch.randn((N, K), device="cuda", generator=g) * 0.02
    b = torch.randn(N, device="cuda", generator=g)
    res = {2: 0.0, 4: 0.0, 8: 0.0}
    for rnd in range(4):
        for rows in (2, 4, 8):
            hip.debug_inner_product_rows(rows)
            for _ in range(10):
                hip.inner_product(x, w, b)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(200):
                hip.inner_product(x, w, b)
            e1.record(); torch.cuda.synchronize()
            res[rows] += e0.elapsed_time(e1) / 200 / 4 * 1e3
    hip.debug_inner_product_rows(0)
    print(f"M={M} N={N} K={K}:  2 rows/wg {res[2]:6.1f} us   4 rows/wg {res[4]:6.1f} us   8 rows/wg {res[8]:6.1f} us", flush=True)
