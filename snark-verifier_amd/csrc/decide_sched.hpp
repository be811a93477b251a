// Host side of the program-driven decide kernel (decide_w.h): builds the operation DAG of one KZG decide
//   e(lhs, g2) * e(rhs, -s_g2)  ->  Gt value        (reference snark-verifier/src/pcs/kzg/decider.rs:70-82)
// -- the 2-pair Miller loop with shared squarings over the prepared line tables (pairing.h `multi_miller_loop`) and the
// exact final exponentiation (pairing.h `final_exponentiation`, same addition chain) -- list-schedules it on the two duos
// of a workgroup by critical path, and allocates LDS registers by liveness.  Pure host C++ (no HIP): the product library
// runs it once per process, tests/hosttest runs it to emulate the kernel.
#pragma once
#include <algorithm>
#include <stdexcept>
#include <vector>
#include "decide_w.h"

namespace snarkv {

struct WtProgram {
  std::vector<WtOp> ops;  // 2 per round: [2 r + duo]
  int rounds = 0;
  int result = 0;         // value index of the register that holds the Gt value after the last round
  int regs_used = 0;
  int critical_path = 0;  // rounds a machine with unlimited duos would need
};

namespace wt_detail {

struct Opnd {
  enum Kind { VAL, LINEA, LINEB, CONST } kind;
  int id;     // VAL: node index; LINEA / LINEB: line index; CONST: value index
  bool conj;  // VAL only: the conjugate of the node's value (a view, not an operation)
};

struct Node {
  uint8_t kind, flags;
  Opnd a, b;
  std::vector<int> deps;   // nodes whose values this one reads
  std::vector<int> after;  // scheduling-only edges: not before these nodes have been issued
  int round = -1, duo = -1, reg = -1, last_use = -1, prio = 0;
};

struct Builder {
  std::vector<Node> n;

  static Opnd val(int id, bool conj = false) { return Opnd{Opnd::VAL, id, conj}; }
  static Opnd conj(Opnd x) {
    if (x.kind != Opnd::VAL) throw std::logic_error("conj of a non-value");
    x.conj = !x.conj;
    return x;
  }
  int add(uint8_t kind, uint8_t flags, Opnd a, Opnd b) {
    Node x;
    x.kind = kind, x.flags = flags, x.a = a, x.b = b;
    if (a.kind == Opnd::VAL) x.deps.push_back(a.id);
    if (b.kind == Opnd::VAL) x.deps.push_back(b.id);
    n.push_back(x);
    return (int)n.size() - 1;
  }
  // a * b.  conj is a ring automorphism: conj(a) conj(b) = conj(a b), so two conjugated operands cost no flag.
  Opnd mul(Opnd a, Opnd b) {
    uint8_t fl = 0;
    bool out_conj = false;
    if (a.kind == Opnd::LINEA) fl |= WT_A_LINE;
    if (b.kind == Opnd::LINEB) fl |= WT_B_LINE;
    if (a.kind == Opnd::LINEB || b.kind == Opnd::LINEA) throw std::logic_error("line in the wrong role");
    const bool ca = a.kind == Opnd::VAL && a.conj, cb = b.kind == Opnd::VAL && b.conj;
    if (ca && cb) out_conj = true;
    else if (ca) fl |= WT_A_CONJ;
    else if (cb) fl |= WT_B_CONJ;
    a.conj = b.conj = false;
    return val(add(WT_MUL, fl, a, b), out_conj);
  }
  // a^(p^k), k = 1, 2, 3: pointwise by the gamma constants; commutes with conj
  Opnd frob(Opnd a, int k) {
    const bool c = a.conj;
    a.conj = false;
    return val(add(WT_PW, (k & 1) ? WT_A_UCONJ : 0, a, Opnd{Opnd::CONST, kWtGamma0 + 12 * (k - 1), false}), c);
  }
  // a view made real (the final result must be a stored value)
  Opnd materialise(Opnd a) {
    if (!a.conj) return a;
    a.conj = false;
    return val(add(WT_PW, WT_A_CONJ_PW, a, Opnd{Opnd::CONST, kWtOnes, false}));
  }
  static constexpr uint8_t WT_A_CONJ_PW = WT_A_CONJ;  // in a PW operation A's coefficient index is k itself: odd k negated

  // a^-1 through the norms Fq12 -> Fq6 -> Fq2 -> Fq (decider.hip coop_inv of round 1, same algebra):
  //   N = a conj(a) (in Fq6), adj = N^(p^2) N^(p^4), d = N adj (in Fq2), a^-1 = conj(a) adj / d
  Opnd inv(Opnd a) {
    Opnd N = mul(a, conj(a));
    Opnd n2 = frob(N, 2), n4 = frob(n2, 2);
    Opnd adj = mul(n2, n4);
    Opnd d = mul(N, adj);
    const int s = add(WT_FQ2INV, 0, d, d);  // (b is not read)
    Opnd ca = mul(conj(a), adj);
    const bool c = ca.conj;
    ca.conj = false;
    const int r = add(WT_PW, WT_B_BCAST, ca, Opnd{Opnd::CONST, kWtScalar, false});
    n[r].deps.push_back(s);  // reads the scalar register the inversion wrote
    return val(r, c);
  }
  // a^x, x = BN254_X_U64, right to left: S <- S^2 beside R <- R * S (pairing.h fq12_exp_by_x gives the same value)
  Opnd exp_by_x(Opnd a) {
    Opnd S = a, R = a;
    bool have = false;
    for (int i = 0; i <= 62; ++i) {
      if ((BN254_X_U64 >> i) & 1ull) {
        R = have ? mul(R, S) : S;
        have = true;
      }
      if (i < 62) S = mul(S, S);
    }
    return R;
  }
};

}  // namespace wt_detail

// window: how many Miller steps the line products may run ahead of the chain (bounds the live registers);
// combine_every: of the steps with an addition line, every `combine_every`-th keeps its two line products separate (the
// chain multiplies twice) so that the line duo does not fall behind the chain duo
inline WtProgram wt_build_program(int window = 5, int combine_skip_every = 6) {
  using namespace wt_detail;
  Builder B;
  auto lineA = [](int t) { return Opnd{Opnd::LINEA, kWtLineA0 + 6 * t, false}; };
  auto lineB = [](int t) { return Opnd{Opnd::LINEB, kWtLineB0 + 12 * t, false}; };
  // ---- Miller loop: f <- f^2 * prod(lines of the step)
  Opnd f{};
  bool have_f = false;
  int idx = 0, set_seen = 0;
  std::vector<int> chain_marks;  // the chain's node after each step (for the look-ahead window)
  auto pace = [&](Opnd x, size_t step) {
    if (x.kind == Opnd::VAL && step >= (size_t)window) B.n[x.id].after.push_back(chain_marks[step - window]);
  };
  size_t step = 0;
  for (int b = kAteBits - 2; b >= 0; --b, ++step) {
    Opnd pd = B.mul(lineA(idx), lineB(idx));
    ++idx;
    pace(pd, step);
    if (have_f) f = B.mul(f, f);
    if (ate_bit(b)) {
      Opnd pa = B.mul(lineA(idx), lineB(idx));
      ++idx;
      pace(pa, step);
      const bool separate = (++set_seen % combine_skip_every) == 0;
      if (separate || !have_f) {
        f = have_f ? B.mul(f, pd) : pd;
        f = B.mul(f, pa);
      } else {
        Opnd m = B.mul(pd, pa);
        pace(m, step);
        f = B.mul(f, m);
      }
    } else {
      f = have_f ? B.mul(f, pd) : pd;
    }
    have_f = true;
    chain_marks.push_back(f.id);
  }
  {
    Opnd p1 = B.mul(lineA(idx), lineB(idx));
    Opnd p2 = B.mul(lineA(idx + 1), lineB(idx + 1));
    idx += 2;
    pace(p1, step), pace(p2, step);
    Opnd m = B.mul(p1, p2);
    f = B.mul(f, m);
  }
  if (idx != kLinesPerG2) throw std::logic_error("line count");
  // ---- final exponentiation (pairing.h final_exponentiation)
  Opnd g = B.mul(Builder::conj(f), B.inv(f));  // f^(p^6 - 1)
  g = B.mul(B.frob(g, 2), g);                  // ^(p^2 + 1)
  Opnd fx = B.exp_by_x(g), fx2 = B.exp_by_x(fx), fx3 = B.exp_by_x(fx2);
  Opnd y0 = B.mul(B.mul(B.frob(g, 1), B.frob(g, 2)), B.frob(g, 3));
  Opnd y1 = Builder::conj(g);
  Opnd y2 = B.frob(fx2, 2);
  Opnd y3 = Builder::conj(B.frob(fx, 1));
  Opnd y4 = Builder::conj(B.mul(fx, B.frob(fx2, 1)));
  Opnd y5 = Builder::conj(fx2);
  Opnd y6 = Builder::conj(B.mul(fx3, B.frob(fx3, 1)));
  Opnd t0 = B.mul(B.mul(B.mul(y6, y6), y4), y5);
  Opnd t1 = B.mul(B.mul(y3, y5), t0);
  t0 = B.mul(t0, y2);
  t1 = B.mul(B.mul(t1, t1), t0);
  t1 = B.mul(t1, t1);
  t0 = B.mul(t1, y1);
  t1 = B.mul(t1, y0);
  Opnd res = B.materialise(B.mul(B.mul(t0, t0), t1));

  // ---- list scheduling on two duos, one operation per duo and round, by critical path
  auto& N = B.n;
  const int M = (int)N.size();
  for (int i = M - 1; i >= 0; --i) {
    N[i].prio = std::max(N[i].prio, 1);
    for (int d : N[i].deps) N[d].prio = std::max(N[d].prio, N[i].prio + 1);
  }
  WtProgram P;
  P.critical_path = 0;
  for (auto& x : N) P.critical_path = std::max(P.critical_path, x.prio);
  int done = 0, round = 0;
  while (done < M) {
    int pick[2] = {-1, -1};
    for (int slot = 0; slot < 2; ++slot) {
      int best = -1;
      for (int i = 0; i < M; ++i) {
        if (N[i].round >= 0 || i == pick[0]) continue;
        bool ready = true;
        for (int d : N[i].deps) ready = ready && N[d].round >= 0 && N[d].round < round;
        for (int d : N[i].after) ready = ready && N[d].round >= 0 && N[d].round <= round;
        if (!ready) continue;
        if (best < 0 || N[i].prio > N[best].prio) best = i;
      }
      pick[slot] = best;
    }
    // the inversion occupies its duo for ~20 rounds' worth of time: nothing else is worth pairing with it, the other
    // duo just waits at the barrier
    for (int slot = 0; slot < 2; ++slot)
      if (pick[slot] >= 0) {
        N[pick[slot]].round = round;
        N[pick[slot]].duo = slot;
        ++done;
      }
    if (pick[0] < 0 && pick[1] < 0) throw std::logic_error("scheduler stalled");
    ++round;
  }
  P.rounds = round;
  // ---- registers by liveness: a value occupies its register from the round it is written to the last round it is read
  for (int i = 0; i < M; ++i) {
    N[i].last_use = N[i].round;
    if (i == res.id) N[i].last_use = round;  // the result stays
  }
  for (int i = 0; i < M; ++i)
    for (int d : N[i].deps) N[d].last_use = std::max(N[d].last_use, N[i].round);
  std::vector<int> order(M);
  for (int i = 0; i < M; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int x, int y) { return N[x].round < N[y].round || (N[x].round == N[y].round && x < y); });
  std::vector<int> free_at(kWtRegs, -1);  // register r is free for a value written in round > free_at[r]
  for (int i : order) {
    if (N[i].kind == WT_FQ2INV) continue;  // writes the scalar register
    int r = -1;
    for (int q = 0; q < kWtRegs; ++q)
      if (free_at[q] < N[i].round && (r < 0 || free_at[q] > free_at[r])) r = q;  // best fit: the most recently freed
    if (r < 0) throw std::runtime_error("decide program: out of LDS registers (shrink the line window)");
    N[i].reg = r;
    free_at[r] = N[i].last_use;
    P.regs_used = std::max(P.regs_used, r + 1);
  }
  // ---- emit
  P.ops.assign((size_t)2 * round, WtOp{0, 0, 0, WT_IDLE, 0});
  auto index_of = [&](const Opnd& o) -> int { return o.kind == Opnd::VAL ? N[o.id].reg * kWtDense : o.id; };
  for (int i = 0; i < M; ++i) {
    WtOp op;
    op.kind = N[i].kind;
    op.flags = N[i].flags;
    op.a = (uint16_t)index_of(N[i].a);
    op.b = (uint16_t)index_of(N[i].b);
    op.dst = (uint16_t)(N[i].kind == WT_FQ2INV ? kWtScalar : N[i].reg * kWtDense);
    P.ops[(size_t)2 * N[i].round + N[i].duo] = op;
  }
  P.result = N[res.id].reg * kWtDense;
  return P;
}

}  // namespace snarkv
