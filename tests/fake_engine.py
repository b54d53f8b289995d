"""Stand-in for the HIP engine in the CPU launcher test of bench.py (tests/test_sharded.py): same call surface as the
package (`generate_pairs_dev`, `pairing_batch_dev`, `last_status`, `layout`), trivial arithmetic on CPU tensors.  Test
infrastructure only -- bench.py labels any line produced with it "[TEST ENGINE, not a measurement]"."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Engine:
    def __init__(self):
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        self.layout = importlib.import_module("plonky2-bn254-pairing_amd").layout

    def generate_pairs_dev(self, seed, g1, g2, n, device=0, stream=None):
        g1.copy_(torch.arange(8 * n, dtype=torch.int64) + seed)
        g2.copy_(torch.arange(16 * n, dtype=torch.int64) * 3 + seed)

    def pairing_batch_dev(self, g1, g2, out, n, device=0, stream=None):
        o = out.view(48, n)
        o[:8] = g1.view(8, n)
        o[8:24] = g2.view(16, n)
        o[24:] = 7

    def last_status(self, device=0, stream=None):
        return None
