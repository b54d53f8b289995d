"""The committed kernel header is exactly what the generator emits (deterministic label names, no hand edits):
tools/gen_kernels.py is re-run in memory and compared byte for byte with plonky2-bn254-pairing_amd/csrc/pairing_asm_gen.h."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_committed_header_is_the_generator_output():
    import gen_kernels
    text, stats = gen_kernels.render(verbose=False)
    with open(gen_kernels.OUT) as f:
        committed = f.read()
    assert text == committed, "pairing_asm_gen.h is stale: run python tools/gen_kernels.py"
    # determinism of the generator itself (label names, section placement): a second build of two kernels in this process
    import kgen4_prog as K4P
    for kw in (dict(generate=True), dict(do_miller=True, do_fexp=False, track=True)):
        assert K4P.KernelBuilder(**kw).build() == K4P.KernelBuilder(**kw).build(), "the generator is not deterministic"


def test_committed_latency_header_is_the_generator_output():
    """csrc/cvm_asm_gen.h (the lane-cooperative kernel and its round program) likewise"""
    import gen_kernels
    text, stats = gen_kernels.render_cvm()
    with open(gen_kernels.OUT_CVM) as f:
        committed = f.read()
    assert text == committed, "cvm_asm_gen.h is stale: run python tools/gen_kernels.py"
