"""Writes tests/golden/kat.json: the reference's known-answer test.

Inputs are the literals of /root/reference/tests/test_spmv.cpp:56-70 (5x5
symmetric matrix, Gaussian x).  Expected outputs `y` and `norm_y` are the
values the UNMODIFIED reference kernel (spmv/csr_kernels.cpp:20-52, general
and symmetric branch, and the OpenMP path with 1 and 2 threads) produced in
this container during the survey stage -- recorded in SURVEY.md Appendix B;
`halo_chain` is the 3-rank L2GMap exchange recorded in SURVEY.md section 8c
(4 rows per rank, chain ghosts, x_local = 100*rank + local index).  The
reference cannot be rebuilt without writing stand-ins for its CMake-generated
headers (see oracle/Makefile), so these recorded outputs plus the test's own
1-ulp norm criterion (test_spmv.cpp:20-23,159-160) are what pins the oracle.

Run from the repo root:  python tests/golden/make_kat.py
The script re-derives x and y with the oracle and refuses to write the file
if they differ from the recorded reference outputs.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

rowptr = [0, 3, 6, 9, 13, 15]
colind = [0, 1, 3, 0, 1, 3, 2, 3, 4, 0, 1, 2, 3, 2, 4]
values = [1.0, -2.0, -3.0, -2.0, 5.0, 4.0, 6.0, 4.0, -4.0, -3.0, 4.0, 4.0, 8.0,
          -4.0, 8.0]
# SURVEY.md Appendix B (reference kernel output, bit for bit)
x_ref = [7.1877817390609889e-28, 1.6918979226151304e-10, 0.082084998623898869,
         0.082084998623898869, 1.6918979226151183e-10]
y_ref = [-0.24625499621007618, 0.32833999534154445, 0.82084998556222943,
         0.98501998416354564, -0.32833999314207712]
norm_ref = 1.3857542692681533

x = oracle.gaussian_x(5)
assert list(x) == x_ref, "oracle x differs from the reference's"
y = oracle.csr_spmv(rowptr, colind, np.array(values), x)
assert list(y) == y_ref, "oracle y differs from the reference's"
assert float(np.sqrt(np.sum(y * y))) == norm_ref

out = dict(
    source="reference tests/test_spmv.cpp:56-80; outputs: SURVEY.md App. B / 8c",
    rowptr=rowptr, colind=colind, values=values, x=x_ref, y=y_ref,
    norm_y=norm_ref,
    halo_chain=dict(ranks=3, rows_per_rank=4,
                    ghosts=[[4], [3, 8], [7]],
                    ghost_tails=[[100.0], [3.0, 200.0], [103.0]]))
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote kat.json")
