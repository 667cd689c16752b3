"""Runs tests/test_gpu_parity.py::test_head_small_tail_is_bit_identical_to_unfused_schedule in a loop inside one process,
NaN-filling freed device memory between rounds (an uninitialised read would show), and counts failures: the tool that
reproduced the first-call race of FusedHeads.  python tools/stress_heads.py [rounds]"""
import sys, traceback
sys.path.insert(0, ".")
import torch
from tests import test_gpu_parity as T
fails = 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    for res, batch in ((9, 3), (4, 2), (16, 1)):
        try:
            T.test_head_small_tail_is_bit_identical_to_unfused_schedule(res, batch)
        except AssertionError as e:
            fails += 1
            print("FAIL iter", i, (res, batch), str(e)[:300])
    # churn the allocator between rounds
    junk = [torch.randn(1 << (10 + (i + j) % 12), device="cuda") * float("nan") for j in range(8)]
    del junk
print("done, fails", fails)
