#!/usr/bin/env python3
"""Per-segment cycle stamps of one stream-K launch (evp_gemm_set_debug_buffer): where a workgroup's time goes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eventpretrain_amd import ops
from eventpretrain_amd._lib import call
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (6272, 768, 3072)))
T = torch.bfloat16
x = torch.randn(M, K, device="cuda").to(T); w = (torch.randn(N, K, device="cuda") * 0.05).to(T)
o = torch.empty(M, N, dtype=T, device="cuda")
buf = torch.zeros(1024 * 16, dtype=torch.int64, device="cuda")
for _ in range(3):
    ops.gemm(x, w, o, M=M, N=N, K=K, tile=12)
call("evp_gemm_set_debug_buffer", buf.data_ptr())
ops.gemm(x, w, o, M=M, N=N, K=K, tile=12)
torch.cuda.synchronize()
call("evp_gemm_set_debug_buffer", 0)
b = buf.cpu().numpy().reshape(1024, 16)
b = b[b[:, 0] != 0]
t0 = b[:, 0].min()
MHZ = float(os.environ.get("TICK_MHZ", 2250.0))   # readcyclecounter ticks per us (calibrated against the HIP-event time of the launch)
print("workgroups that reported: %d" % len(b))
rows = []
for g in range(len(b)):
    prev = b[g, 0]
    for s_ in range(7):
        end, info = b[g, 1 + 2 * s_], b[g, 2 + 2 * s_]
        if end == 0:
            break
        rows.append((g, s_, int(info >> 16), int(info & 0xFFFF), (prev - t0) / MHZ, (end - prev) / MHZ))
        prev = end
rows = np.array(rows, dtype=np.float64)
print("kernel span %.1f us; start skew max %.1f us" % ((b[:, 1:14:2].max() - t0) / MHZ, (b[:, 0].max() - t0) / MHZ))
for mode, name in ((1, "park"), (2, "whole"), (3, "collect")):
    r = rows[rows[:, 2] == mode]
    if len(r):
        per = r[:, 5] / np.maximum(r[:, 3], 1)
        print("%-8s n=%4d  K tiles avg %5.1f  duration avg %6.2f us (min %5.2f max %6.2f)  us per K tile %5.2f" %
              (name, len(r), r[:, 3].mean(), r[:, 5].mean(), r[:, 5].min(), r[:, 5].max(), (r[:, 5].sum() / r[:, 3].sum())))
# linear fit duration = a + b * ktiles per mode
for mode, name in ((1, "park"), (3, "collect"), (2, "whole")):
    r = rows[rows[:, 2] == mode]
    if len(r) > 10 and r[:, 3].std() > 0:
        A = np.stack([np.ones(len(r)), r[:, 3]], 1)
        coef = np.linalg.lstsq(A, r[:, 5], rcond=None)[0]
        print("%-8s fit: %.2f us fixed + %.3f us per K tile" % (name, coef[0], coef[1]))

for g in (0, 1, 2, 3, 30, 31, 32, 60, 61, 62, 63):
    sel = rows[rows[:, 0] == g]
    print("row %3d: " % g + "  ".join("[%s k=%2d start %5.1f dur %5.1f]" % ({1: "park", 2: "whole", 3: "coll"}[int(r[2])], int(r[3]), r[4], r[5]) for r in sel))

c = b[b[:, 13] != 0]
# end stamp of the collect segment = last recorded segment end
ends = np.array([r[1:14:2][r[1:14:2] != 0].max() for r in c])
starts = np.array([np.sort(r[1:14:2][r[1:14:2] != 0])[-2] if (r[1:14:2] != 0).sum() > 1 else r[0] for r in c])
print("collect segments: main loop %.1f us, wait for the flag %.1f us, add the parts %.1f us, epilogue %.1f us (averages)" %
      (((c[:, 13] - starts) / MHZ).mean(), ((c[:, 14] - c[:, 13]) / MHZ).mean(), ((c[:, 15] - c[:, 14]) / MHZ).mean(), ((ends - c[:, 15]) / MHZ).mean()))
