#!/usr/bin/env python3
"""The shipped L1 routines, timed in isolation."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import kgen3 as K3
from kgen import Emitter
V = {}
for name in ("mul", "mul3", "sqr", "mulfq", "fqmul", "fqsqr", "norm", "redn", "mulxi"):
    e = Emitter()
    getattr(K3.L1v3(e), "r_" + name)()
    V["L1 " + name] = [l.replace("%=", "0") for l in e.finalize() if not l.strip().startswith("s_setpc")]
json.dump(V, open(sys.argv[1], "w"))
