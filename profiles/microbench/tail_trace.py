#!/usr/bin/env python3
"""Diagnostic (round 6): the tail of a large batch's solve, dispatch by dispatch.  Two modes:
  run:     rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tt -- python3 profiles/microbench/tail_trace.py run [B]
  report:  python3 profiles/microbench/tail_trace.py report gpurun_out/tt      (the LAST solve of the trace, per hardware queue)"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    import torch
    from quadrotorilqr_amd import capi, problems as pb
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    dev = torch.device("cuda", 0)
    cfg = pb.config2(B=B, N=100, seed=4)
    init = torch.from_numpy(cfg["init"]).to(dev)
    bufs = (torch.empty_like(init), torch.empty(B, dtype=torch.float64, device=dev), [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(4)])
    s = capi.from_config(cfg, device=0)
    for _ in range(3):
        s.solve_batch_device(init, bufs[0], bufs[1], *bufs[2])
        torch.cuda.synchronize()
    s.close()
else:
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    begins = [i for i, r in enumerate(rows) if "k_begin" in r["Kernel_Name"]]
    rows = rows[begins[-1]:]
    t0 = int(rows[0]["Start_Timestamp"])
    short = lambda n: n.split("qilqr::")[-1].split("(")[0][:44]
    queues = sorted({r["Queue_Id"] for r in rows})
    print("solve: %.1f us, %d dispatches, queues %s" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e3, len(rows), queues))
    for q in queues:
        mine = [r for r in rows if r["Queue_Id"] == q]
        print(f"-- queue {q}: {len(mine)} dispatches")
        prev_end = None
        for r in mine:
            st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gap = (st - prev_end) / 1e3 if prev_end else 0.0
            print(f"   {(st - t0) / 1e3:10.1f} us  +{(en - st) / 1e3:8.1f}  gap {gap:7.1f}  grid {r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', '?'):>8}  {short(r['Kernel_Name'])}")
            prev_end = en
