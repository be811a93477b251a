"""CPU: the committed generated sources are what their generators produce (constants of both curves, the
asm multiplier bodies) -- a stale header would silently change field arithmetic on the device."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "snark-verifier_amd", "csrc")


def _gen(script, *args):
    return subprocess.run([sys.executable, os.path.join(CSRC, script)] + list(args), check=True, capture_output=True,
                          text=True).stdout


def test_curve_constants_are_current():
    assert _gen("gen_consts.py") == open(os.path.join(CSRC, "bn254_consts.h")).read()
    assert _gen("gen_consts.py", "pallas") == open(os.path.join(CSRC, "pallas_consts.h")).read()


def test_multiplier_bodies_are_current():
    for kind in ("mul", "mul2", "sqr"):
        assert _gen("gen_fq29_mul_asm.py", kind) == open(os.path.join(CSRC, "fq29_%s_asm.inc" % kind)).read(), kind
    assert _gen("gen_fq_mul_asm.py") == open(os.path.join(CSRC, "fq_mul_asm.inc")).read()


def test_generic_and_bn254_names_agree():
    """bn254_consts.h carries the field constants twice (generic SNARKV_* for the shared layers, BN254_* for the
    pairing): they must be the same numbers."""
    import re

    txt = open(os.path.join(CSRC, "bn254_consts.h")).read()

    def val(name):
        m = re.search(r"#define %s (\{[^}]*\}|\S+)" % name, txt)
        assert m, name
        return m.group(1)

    assert val("SNARKV_FQ_P_LIMBS") == val("BN254_P_LIMBS")
    assert val("SNARKV_FQ_ONE_MONT") == val("BN254_ONE_MONT")
    assert val("SNARKV_FQ_R2_MONT") == val("BN254_R2_MONT")
    assert val("SNARKV_FQ_P_INV32") == val("BN254_P_INV32")
    assert val("SNARKV_FR_R_LIMBS") == val("BN254_R_LIMBS")
    assert val("SNARKV_G1_B_MONT") == val("BN254_THREE_MONT")
