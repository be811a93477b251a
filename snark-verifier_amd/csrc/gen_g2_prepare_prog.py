#!/usr/bin/env python3
"""Generates g2_prepare_prog.inc: the level-by-level program of the wavefront-parallel G2 line-table kernel
(k_g2_prepare_w, g2_prepare_w.h) -- `G2Prepared::from` of the reference's decider (pcs/kzg/decider.rs:74), the formulas of
pairing.h g2_double_step / g2_add_step.

One LEVEL = up to seven Fq2 products A * B that do not depend on each other; A and B are small integer combinations of
LDS slots (at most three terms, coefficients in -3 .. 3), so the additions / subtractions / doublings of the formulas cost
no level of their own.  A doubling step is 4 levels, an addition step 5; the line coefficients of a step are reduced to
canonical form and stored by spare lanes during the next step's first level.  64 doublings + 38 additions = 446 levels
(+ one set-up level and one flush level) instead of ~3 000 dependent Fq products on one lane.

    python gen_g2_prepare_prog.py > g2_prepare_prog.inc
"""
ATE = 29793968203157093288  # 6x + 2
NBITS = 65

# ---- slots (Fq2 values in LDS)
names = []


def slot(n):
    names.append(n)
    return len(names) - 1


ONE, B3, G12, G13, G22, G23 = (slot(n) for n in ("ONE", "B3", "G12", "G13", "G22", "G23"))
QX, QY, Q1X, Q1Y, Q2X, Q2Y = (slot(n) for n in ("QX", "QY", "Q1X", "Q1Y", "Q2X", "Q2Y"))
TB = [[slot("T%d_%s" % (b, c)) for c in ("X", "Z", "YA", "YB")] for b in range(2)]  # T = (X, YA - YB, Z), two buffers
TMP = [[slot("t%d_%d" % (p, i)) for i in range(12)] for p in range(2)]                # temporaries, two sets by step parity
NSLOTS = len(names)

levels = []  # each: list of tasks (dst, A, B, flags, out)


def term(*pairs):
    t = list(pairs)[:3]
    while len(t) < 3:
        t.append((0, 0))
    return t


def task(dst, A, B, conj_a=False, out=-1):
    return {"dst": dst, "A": A, "B": B, "conj": conj_a, "out": out}


pending = []  # output tasks of the previous step: (line index, coefficient 0..2, combination)
line_idx = 0


def flush_into(level):
    global pending
    for (li, which, comb) in pending:
        level.append(task(-1, comb, term((ONE, 1)), out=3 * li + which))
    pending = []


def double_step(c, n, p):
    """T_n = 2 T_c; temporaries of parity p; line (cy, cx, cw) = (2 Y Z, -3 X^2, Y^2 - 3b' Z^2)"""
    global line_idx
    X, Z, YA, YB = TB[c]
    Xn, Zn, YAn, YBn = TB[n]
    xx, yy, zz, yz, bb, aa, b3zz, xbb, aaz, bbb = TMP[p][:10]
    Y = term((YA, 1), (YB, -1))
    l1 = [task(xx, term((X, 1)), term((X, 1))), task(yy, Y, Y), task(zz, term((Z, 1)), term((Z, 1))), task(yz, Y, term((Z, 1)))]
    flush_into(l1)
    l2 = [task(bb, term((yz, 2)), term((yz, 2))), task(aa, term((xx, 3)), term((xx, 3))), task(b3zz, term((B3, 1)), term((zz, 1)))]
    l3 = [task(xbb, term((X, 1)), term((bb, 1))), task(aaz, term((aa, 1)), term((Z, 1))), task(bbb, term((bb, 1)), term((yz, 2)))]
    l4 = [task(YAn, term((xx, 3)), term((xbb, 3), (aaz, -1))), task(YBn, Y, term((bbb, 1))),
          task(Xn, term((aaz, 1), (xbb, -2)), term((yz, 2))), task(Zn, term((bbb, 1)), term((Z, 1)))]
    levels.extend([l1, l2, l3, l4])
    pending.extend([(line_idx, 0, term((yz, 2))), (line_idx, 1, term((xx, -3))), (line_idx, 2, term((yy, 1), (b3zz, -1)))])
    line_idx += 1


def add_step(c, n, p, qx, qy):
    """T_n = T_c + Q; line (cy, cx, cw) = (F, -E, E x2 - F y2) with E = y2 Z - Y, F = x2 Z - X"""
    global line_idx
    X, Z, YA, YB = TB[c]
    Xn, Zn, YAn, YBn = TB[n]
    y2z, x2z, ym, ff, ee, ex2, fy2, d, eez, xff, x2d = TMP[p][:11]
    Y = term((YA, 1), (YB, -1))
    E = term((y2z, 1), (ym, -1))
    F = term((x2z, 1), (X, -1))
    l1 = [task(y2z, term((qy, 1)), term((Z, 1))), task(x2z, term((qx, 1)), term((Z, 1))), task(ym, Y, term((ONE, 1)))]
    flush_into(l1)
    l2 = [task(ff, F, F), task(ee, E, E), task(ex2, E, term((qx, 1))), task(fy2, F, term((qy, 1)))]
    l3 = [task(d, term((ff, 1)), term((Z, 1))), task(eez, term((ee, 1)), term((Z, 1))), task(xff, term((X, 1)), term((ff, 1)))]
    l4 = [task(x2d, term((qx, 1)), term((d, 1))), task(Zn, F, term((d, 1)))]
    l5 = [task(YAn, E, term((x2d, 2), (eez, -1), (xff, 1))), task(YBn, term((qy, 1)), term((Zn, 1))),
          task(Xn, term((eez, 1), (xff, -1), (x2d, -1)), F)]
    levels.extend([l1, l2, l3, l4, l5])
    pending.extend([(line_idx, 0, F), (line_idx, 1, [(s, -k) for s, k in E]), (line_idx, 2, term((ex2, 1), (fy2, -1)))])
    line_idx += 1


# set-up level: the Frobenius images  q1 = (conj(x) g12, conj(y) g13),  q2 = (x g22, -y g23)   [-pi^2(Q)]
levels.append([task(Q1X, term((QX, 1)), term((G12, 1)), conj_a=True), task(Q1Y, term((QY, 1)), term((G13, 1)), conj_a=True),
               task(Q2X, term((QX, 1)), term((G22, 1))), task(Q2Y, term((QY, -1)), term((G23, 1)))])
cur, step = 0, 0
for i in range(NBITS - 2, -1, -1):
    double_step(cur, cur ^ 1, step & 1)
    cur ^= 1
    step += 1
    if (ATE >> i) & 1:
        add_step(cur, cur ^ 1, step & 1, QX, QY)
        cur ^= 1
        step += 1
for qx, qy in ((Q1X, Q1Y), (Q2X, Q2Y)):
    add_step(cur, cur ^ 1, step & 1, qx, qy)
    cur ^= 1
    step += 1
last = []
flush_into(last)
levels.append(last)
assert line_idx == 64 + 36 + 2
MAXT = max(len(lv) for lv in levels)
assert MAXT <= 7

print("// GENERATED by gen_g2_prepare_prog.py -- do not edit.")
print("// %d levels of at most %d Fq2 products; %d slots.  A task: dst slot (-1: a line coefficient, `out` = 3 * line + which)," % (len(levels), MAXT, NSLOTS))
print("// A and B as three (slot, coefficient) terms each, conj = use the Fq2-conjugate of A.")
print("constexpr int kG2wLevels = %d, kG2wTasks = %d, kG2wSlots = %d;" % (len(levels), MAXT, NSLOTS))
print("constexpr int kG2wSlotONE = %d, kG2wSlotB3 = %d, kG2wSlotG12 = %d, kG2wSlotG13 = %d, kG2wSlotG22 = %d, kG2wSlotG23 = %d;" % (ONE, B3, G12, G13, G22, G23))
print("constexpr int kG2wSlotQX = %d, kG2wSlotQY = %d, kG2wSlotTX = %d, kG2wSlotTZ = %d, kG2wSlotTYA = %d, kG2wSlotTYB = %d;" % (QX, QY, TB[0][0], TB[0][1], TB[0][2], TB[0][3]))
print("struct G2wTask { int16_t dst, out; int8_t as[3], ac[3], bs[3], bc[3]; int8_t conj, used; };")
print("static const G2wTask kG2wProg[kG2wLevels][kG2wTasks] = {")
for lv in levels:
    row = []
    for t in lv + [None] * (MAXT - len(lv)):
        if t is None:
            row.append("{-1, -1, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, 0, 0}")
        else:
            A, B = list(t["A"]), list(t["B"])
            while len(A) < 3:
                A.append((0, 0))
            while len(B) < 3:
                B.append((0, 0))
            row.append("{%d, %d, {%d, %d, %d}, {%d, %d, %d}, {%d, %d, %d}, {%d, %d, %d}, %d, 1}" % (
                t["dst"], t["out"], A[0][0], A[1][0], A[2][0], A[0][1], A[1][1], A[2][1],
                B[0][0], B[1][0], B[2][0], B[0][1], B[1][1], B[2][1], 1 if t["conj"] else 0))
    print("  {" + ", ".join(row) + "},")
print("};")
