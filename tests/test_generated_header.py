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


def _inflate_programs(text):
    """The header with every deflated + base64-encoded round program replaced by the SHA-256 of its INFLATED bytes (and without the
    deflated lengths): what must be reproducible is the program, not the deflate stream -- zlib builds (zlib-ng, distribution patches,
    other versions) emit different streams for the same input."""
    import base64
    import hashlib
    import re
    import zlib

    def digest(m):
        raw = zlib.decompress(base64.b64decode("".join(re.findall(r'"([^"]*)"', m.group(2)))))
        return f"static const char BN254_CVM_{m.group(1)}_B64[] = <{len(raw)} bytes, sha256 {hashlib.sha256(raw).hexdigest()}>;"

    text = re.sub(r'static const char BN254_CVM_(\w+)_B64\[\] =\n((?:    "[^"\n]*"\n)+)    ;', digest, text)
    return re.sub(r"#define BN254_CVM_\w+_Z_BYTES \d+\n", "", text)


def test_committed_latency_header_is_the_generator_output():
    """csrc/cvm_asm_gen.h (the lane-cooperative kernel and its round programs) likewise: the kernel text byte for byte, the programs
    by the digest of their inflated blobs."""
    import gen_kernels
    text, stats = gen_kernels.render_cvm()
    with open(gen_kernels.OUT_CVM) as f:
        committed = f.read()
    a, b = _inflate_programs(text), _inflate_programs(committed)
    assert a.count("sha256") == 35 and b.count("sha256") == 35
    assert a == b, "cvm_asm_gen.h is stale: run python tools/gen_kernels.py"
