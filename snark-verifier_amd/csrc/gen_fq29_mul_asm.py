#!/usr/bin/env python3
"""Generates fq29_{mul,mul2,sqr}_asm.inc: the DEVICE bodies of fq29_mul / fq29_mul2 / fq29_sqr (fq29.h) with every column of the product
scanning loop written as ONE chain of `v_mad_i64_i32` on the 64-bit column accumulator -- operand
products a_i * b_j and reduction products m_i * p_j alike (p_j from SGPRs).  Left to the compiler, the two
kinds of products become two chains joined by a 64-bit add per column, with moves around them; on this
machine every VOP3 instruction costs the same issue slot, so the column is spelled out.
An asm statement takes at most 30 operands: the widest columns are split in two.  Products with a limb
of the modulus that is zero are left out (the plain-C form got that from constant folding).

    python gen_fq29_mul_asm.py mul > fq29_mul_asm.inc ; ... mul2 > fq29_mul2_asm.inc ; ... sqr > fq29_sqr_asm.inc
"""
MAXOPS = 30


def emit_chain(prods):
    """prods: list of (x_expr, y_expr, y_is_sgpr) accumulated into acc"""
    out = []
    while prods:
        take = prods[: (MAXOPS - 1) // 2]
        prods = prods[len(take):]
        lines, ops = [], []
        for n, (x, y, sg) in enumerate(take):
            lines.append("v_mad_i64_i32 %%0, vcc, %%%d, %%%d, %%0" % (1 + 2 * n, 2 + 2 * n))
            ops.append('"v"(%s)' % x)
            ops.append(('"s"(%s)' if sg else '"v"(%s)') % y)
        out.append('    asm("%s"\n        : "+v"(acc)\n        : %s\n        : "vcc");' % ("\\n\\t".join(lines), ", ".join(ops)))
    return "\n".join(out)


import sys


def body(kind, zero_limbs=()):
    """kind: 'mul' (a*b), 'mul2' (a*b + c*d), 'sqr' (a*a with the doubled limbs a2[]);
    zero_limbs: 29-bit limbs of the modulus that are 0 (their reduction products are left out)"""
    print("  int32_t m[9];")
    print("  Fq29 r;")
    if kind == "sqr":
        print("  int32_t a2[9];")
        print("#pragma unroll")
        print("  for (int i = 0; i < 9; ++i) a2[i] = a.v[i] * 2;")
    print("  int64_t acc = 0;")
    for k in range(17):
        lo, hi = max(0, k - 8), min(k, 8)
        if kind == "sqr":
            prods = [("a2[%d]" % i, "a.v[%d]" % (k - i), False) for i in range(lo, hi + 1) if 2 * i < k]
            if k % 2 == 0:
                prods.append(("a.v[%d]" % (k // 2), "a.v[%d]" % (k // 2), False))
        else:
            prods = [("a.v[%d]" % i, "b.v[%d]" % (k - i), False) for i in range(lo, hi + 1)]
            if kind == "mul2":
                prods += [("c.v[%d]" % i, "d.v[%d]" % (k - i), False) for i in range(lo, hi + 1)]
        prods += [("m[%d]" % i, "fq29_p(%d)" % (k - i), True) for i in range(lo, hi + 1)
                  if i < k and 1 <= k - i <= 8 and (k - i) not in zero_limbs]
        print("  {  // column %d" % k)
        print(emit_chain(prods))
        if k < 9:
            print("    m[%d] = (int32_t)(((uint32_t)acc * (uint32_t)SNARKV_FQ29_NINV) & (uint32_t)kMask29);" % k)
            print(emit_chain([("m[%d]" % k, "fq29_p(0)", True)]))
        else:
            print("    r.v[%d] = (int32_t)acc & kMask29;" % (k - 9))
        print("    acc >>= 29;")
        print("  }")
    print("  r.v[8] = (int32_t)acc;")
    print("  return r;")


class Prod:
    """One Montgomery product of a PAIR: kind 'mul' / 'mul2' / 'sqr', operand names, a suffix for its locals."""

    def __init__(self, kind, ops, sfx, out):
        self.kind, self.ops, self.sfx, self.out = kind, ops, sfx, out

    def decls(self):
        s = self.sfx
        print("  int32_t m%s[9];" % s)
        if self.kind == "sqr":
            print("  int32_t dbl%s[9];" % s)
            print("#pragma unroll")
            print("  for (int i = 0; i < 9; ++i) dbl%s[i] = %s.v[i] * 2;" % (s, self.ops[0]))
        print("  int64_t acc%s = 0;" % s)

    def prods(self, k, zero_limbs):
        lo, hi = max(0, k - 8), min(k, 8)
        o, s = self.ops, self.sfx
        if self.kind == "sqr":
            pr = [("dbl%s[%d]" % (s, i), "%s.v[%d]" % (o[0], k - i), False) for i in range(lo, hi + 1) if 2 * i < k]
            if k % 2 == 0:
                pr.append(("%s.v[%d]" % (o[0], k // 2), "%s.v[%d]" % (o[0], k // 2), False))
        else:
            pr = [("%s.v[%d]" % (o[0], i), "%s.v[%d]" % (o[1], k - i), False) for i in range(lo, hi + 1)]
            if self.kind == "mul2":
                pr += [("%s.v[%d]" % (o[2], i), "%s.v[%d]" % (o[3], k - i), False) for i in range(lo, hi + 1)]
        pr += [("m%s[%d]" % (s, i), "fq29_p(%d)" % (k - i), True) for i in range(lo, hi + 1)
               if i < k and 1 <= k - i <= 8 and (k - i) not in zero_limbs]
        return pr


def emit_chain_named(prods, acc):
    return emit_chain(prods).replace('"+v"(acc)', '"+v"(%s)' % acc)


def pair_body(pa, pb, zero_limbs=()):
    """Two INDEPENDENT products column by column: A's chain, B's chain, A's tail, B's tail.  The compiler pads an asm
    statement whose VGPR result is read by the very next instruction with `s_nop` (it must assume the asm wrote with
    dst_sel); with a second, independent product in between that next instruction is never the dependent one."""
    pa.decls()
    pb.decls()
    for k in range(17):
        print("  {  // column %d" % k)
        for pr in (pa, pb):
            print(emit_chain_named(pr.prods(k, zero_limbs), "acc" + pr.sfx))
        if k < 9:
            for pr in (pa, pb):
                print("    m%s[%d] = (int32_t)(((uint32_t)acc%s * (uint32_t)SNARKV_FQ29_NINV) & (uint32_t)kMask29);" % (pr.sfx, k, pr.sfx))
            for pr in (pa, pb):
                print(emit_chain_named([("m%s[%d]" % (pr.sfx, k), "fq29_p(0)", True)], "acc" + pr.sfx))
        else:
            for pr in (pa, pb):
                print("    %s.v[%d] = (int32_t)acc%s & kMask29;" % (pr.out, k - 9, pr.sfx))
        for pr in (pa, pb):
            print("    acc%s >>= 29;" % pr.sfx)
        print("  }")
    for pr in (pa, pb):
        print("  %s.v[8] = (int32_t)acc%s;" % (pr.out, pr.sfx))


PAIRS = {
    # name: (kind A, operands A, kind B, operands B); results r1, r2
    "mul_mul": (("mul", ("a", "b")), ("mul", ("c", "d"))),
    "sqr_sqr": (("sqr", ("a",)), ("sqr", ("c",))),
    "mul2_mul": (("mul2", ("a", "b", "c", "d")), ("mul", ("e", "f"))),
}

kind = sys.argv[1] if len(sys.argv) > 1 else "mul"
if kind in PAIRS:
    (ka, oa), (kb, ob) = PAIRS[kind]
    print("// GENERATED by gen_fq29_mul_asm.py %s -- do not edit." % kind)
    print("#if defined(SNARKV_CURVE_PALLAS)")
    pair_body(Prod(ka, oa, "_a", "r1"), Prod(kb, ob, "_b", "r2"), zero_limbs=(5, 6, 7))
    print("#else")
    pair_body(Prod(ka, oa, "_a", "r1"), Prod(kb, ob, "_b", "r2"))
    print("#endif")
    sys.exit(0)
print("// GENERATED by gen_fq29_mul_asm.py %s -- do not edit." % kind)
# pallas: p = 2^254 + (a 125-bit number): limbs 5, 6, 7 of its 9 x 29-bit form are zero (pallas_consts.h)
print("#if defined(SNARKV_CURVE_PALLAS)")
print("  static_assert(fq29_p(5) == 0 && fq29_p(6) == 0 && fq29_p(7) == 0, \"zero limbs of the pallas modulus\");")
body(kind, zero_limbs=(5, 6, 7))
print("#else")
body(kind)
print("#endif")
