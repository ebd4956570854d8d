"""One-off: the row-block composition check of tests/test_gpu_parity.py at large ragged sizes (rendezvous, split-major
order and bf16 column chunks active):  python tools/big_compose.py   (GPU box; 7 cases passed at 9c0ffee+)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_parity as T
cases = [(33001, 768, 2, torch.float32), (33001, 768, 3, torch.float32), (24011, 1024, 2, torch.float32),
         (41003, 768, 2, torch.bfloat16), (41003, 768, 5, torch.bfloat16), (70001, 300, 2, torch.float32), (90017, 512, 3, torch.bfloat16)]
bad = 0
for c in cases:
    try:
        T.test_mmd_row_blocks_compose(*c); print("ok", c, flush=True)
    except AssertionError as e:
        bad += 1; print("FAIL", c, repr(e)[:300], flush=True)
sys.exit(1 if bad else 0)
