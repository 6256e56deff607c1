"""Generates csrc/swimmer_row_fused.h: the inline-asm blocks of the row kernel (n = 4..8) that use
gfx950's `v_fmac_f64_dpp ... row_newbcast:k` (acc += [lane k of the row].src * mul), one
instruction instead of a 64-bit DPP move plus an FMA.  hipcc does not form this instruction
itself, and it knows nothing about DPP hazards inside inline asm, so every block is ONE asm
statement whose instruction order is fixed here:

  * a DPP operand must not have been written by one of the two preceding instructions
    ("VALU write -> DPP read: 2 wait states"); the comments in the generated blocks say which
    instruction wrote each DPP source and how far back;
  * dependent accumulators are interleaved so that a chain never waits on its own result.

Run:  python scripts/gen_row_fused.py   (rewrites the header in place)
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "safe-exploration-with-simulator-in-rl-algorithms_amd", "csrc",
                   "swimmer_row_fused.h")
DPP = "row_newbcast:{k} row_mask:0xf bank_mask:0xf"


def fmac(acc, src, mul, k):
    return f"v_fmac_f64_dpp %{acc}, %{src}, %{mul} " + DPP.format(k=k)


def asm_block(lines, outs, ins, indent="        "):
    body = "\n".join(f'{indent}    "{l}\\n"' for l in lines)
    o = ", ".join(f'"+v"({x})' for x in outs)
    i = ", ".join(f'"v"({x})' for x in ins)
    return f"{indent}asm volatile(\n{body}\n{indent}    : {o}\n{indent}    : {i});\n"


def gen(n):
    d = 2 * n + 2
    s = []
    s.append(f"template <>\nstruct RowFused<{n}> {{\n")
    # ---- policy ----
    outs = ["tq0", "tq1"]
    ins = ["th", "thd"] + [f"V[{2 + j}]" for j in range(2 * n)] + ["after"]
    lines = []
    for k in range(n):
        lines.append(fmac(0, 2, 4 + 2 * k, k))
        lines.append(fmac(1, 3, 5 + 2 * k, k))
    s.append(f"""    // tq0 += sum_k th_k V[2+2k], tq1 += sum_k thd_k V[3+2k]   (th_k = lane k's theta).
    // DPP sources th, thd: written by the previous step's Euler update; `after` (this step's
    // sin) only orders the block behind the ~30 instructions of sincos_fast.
    static __device__ __forceinline__ void policy(double &tq0, double &tq1, double th, double thd,
                                                  const double (&V)[{d}], double after)
    {{
""")
    s.append(asm_block(lines, outs, ins))
    s.append("    }\n\n")
    # ---- g / r(sq) ----
    outs = ["g", "r"]
    ins = ["thd", "sq"] + [f"vc[{k}]" for k in range(n)] + [f"tc[{k}]" for k in range(n)]
    gk = [fmac(0, 2, 4 + k, k) for k in range(n)]
    rk = [fmac(1, 3, 4 + n + k, k) for k in range(n)]
    lines = []
    for k in range(n):
        lines.append(gk[k])
        if k >= 1:
            lines.append(rk[k - 1])
    lines.append(rk[n - 1])
    s.append(f"""    // g += sum_k thd_k vc[k]   (normal velocity);   r += sum_k thd_k^2 tc[k]   (centripetal part
    // of the right-hand side; sq = own thd^2).  The two chains alternate; the block ENDS with two
    // r instructions so that g's last write is 2 instructions old when `sums` reads g by DPP.
    static __device__ __forceinline__ void velocity(double &g, double &r, double thd, double sq,
                                                    const double (&vc)[{n}], const double (&tc)[{n}])
    {{
""")
    s.append(asm_block(lines, outs, ins))
    s.append("    }\n\n")
    # ---- sums ----
    outs = ["sx", "sy", "r"]
    ins = ["g"] + [f"sk[{k}]" for k in range(n)] + [f"ck[{k}]" for k in range(n)] + \
          [f"ac[{k}]" for k in range(n)]
    lines = []
    for k in range(n):
        lines.append(fmac(0, 3, 4 + k, k))
        lines.append(fmac(1, 3, 4 + n + k, k))
        lines.append(fmac(2, 3, 4 + 2 * n + k, k))
    s.append(f"""    // sx += sum_k g_k sin_k, sy += sum_k g_k cos_k (barycentre acceleration, canonical order, so
    // bit-identical on every lane);   r += sum_k g_k ac[k]   (friction part of the right-hand side).
    // DPP source g: last written 2 instructions before the end of `velocity`.
    static __device__ __forceinline__ void sums(double &sx, double &sy, double &r, double g,
                                                const double (&sk)[{n}], const double (&ck)[{n}],
                                                const double (&ac)[{n}])
    {{
""")
    s.append(asm_block(lines, outs, ins))
    s.append("    }\n\n")
    # ---- ordering point ----
    allrows = ", ".join(f'"+v"(a[{k}])' for k in range(n))
    s.append(f"""    // No instruction: an ordering point.  Every a[k] is written before it; the first pivot's
    // broadcast, the sel / nm arithmetic and the block's own multiplies (>= 4 + N instructions)
    // come after it, so the DPP reads of eliminate<0> are never closer than that to the writes
    // of the matrix row.
    static __device__ __forceinline__ void fence(double (&a)[{n}])
    {{
        asm volatile("" : {allrows});
    }}

""")
    # ---- elimination step (division-free) ----
    s.append(f"""    // Elimination step J, division-free: row_i <- sel * row_i + nm * [lane J].row   with
    // sel = pivot Q_JJ on lanes i > J (1 elsewhere) and nm = -Q_iJ on lanes i > J (0 elsewhere; lane J
    // itself multiplies by 1 and adds 0).  The rows span the same space as with the usual multiplier
    // Q_iJ / Q_JJ, but no step needs 1 / pivot: the serial pivot chain carries no v_rcp_f64 (16 issue
    // cycles each), every lane takes ONE reciprocal of its own scaled pivot after the last step.
    // All multiplies first, then the fused broadcast-FMAs: no instruction waits on its neighbour.
    // DPP sources a[k], r: written by the previous step's block, behind this step's pivot broadcast
    // and the two instructions that form sel / nm (the first block sits behind `fence`).  a[J+1] --
    // the next pivot, which the compiler's own DPP move reads next -- is written >= 2 instructions
    // before the end (the two-entry block pads with s_nop 0).
    template <int J>
    static __device__ __forceinline__ void eliminate(double (&a)[{n}], double &r, double sel, double nm)
    {{
""")
    for j in range(n - 1):
        ents = [f"a[{k}]" for k in range(j + 1, n)] + ["r"]
        ne = len(ents)
        # operands: outputs t0..t(ne-1) (early clobber), inputs e0..e(ne-1), sel, nm
        lines = [f"v_mul_f64 %{i}, %{ne + i}, %{2 * ne}" for i in range(ne)]
        lines += [f"v_fmac_f64_dpp %{i}, %{ne + i}, %{2 * ne + 1} " + DPP.format(k=j) for i in range(ne)]
        if ne == 2:
            lines.append("s_nop 0")
        kw = "if" if j == 0 else "else if"
        s.append(f"        {kw} constexpr (J == {j}) {{\n")
        s.append("            double " + ", ".join(f"t{i}" for i in range(ne)) + ";\n")
        body = "\n".join(f'                "{l}\\n"' for l in lines)
        o = ", ".join(f'"=&v"(t{i})' for i in range(ne))
        i_ = ", ".join(f'"v"({x})' for x in ents + ["sel", "nm"])
        s.append(f"            asm volatile(\n{body}\n                : {o}\n                : {i_});\n")
        for i, e in enumerate(ents):
            s.append(f"            {e} = t{i};\n")
        s.append("        }\n")
    s.append("    }\n\n")
    # ---- back-substitution ----
    # y_i = r_i / Q_ii - sum_{k>i} (Q_ik / Q_ii) x_k, processed from the last row up: level J
    # broadcasts x_J = [lane J].y (final by then) and every lane above adds nu[J] * x_J with
    # nu[J] = -(Q_iJ / Q_ii) on lanes i < J, 0 elsewhere.  nu[J] = (nabove[J] * a[J]) * rq is two
    # multiplies; they sit in the two slots the DPP hazard wants between consecutive levels
    # (m1 two levels ahead, m2 one level ahead, so nothing waits on its neighbour).
    T = 4                                   # rotating temporaries
    outs = ["y"] + [f"t{i}" for i in range(T)]
    ins = ["rq", "nu_top", "na_next"] + [f"a[{k}]" for k in range(1, n)] + [f"nab[{k}]" for k in range(1, n)]
    o_idx = {name: i for i, name in enumerate(outs)}
    i_idx = {name: len(outs) + i for i, name in enumerate(ins)}
    lines = []
    free = [f"t{i}" for i in range(T)]
    na_reg = {n - 2: "na_next"}            # na(J) = nabove[J] * a[J], given for J = n-2
    nu_reg = {n - 1: "nu_top"}             # nu(J), given for J = n-1
    def op(name):
        return "%" + str(o_idx[name] if name in o_idx else i_idx[name])
    def alloc():
        return free.pop(0)
    def release(name):
        if name in o_idx and name != "y":
            free.append(name)
    for J in range(n - 1, 0, -1):
        gap = []
        # m2(J-1): nu(J-1) = na(J-1) * rq
        if J - 1 >= 1:
            t = alloc()
            gap.append(f"v_mul_f64 {op(t)}, {op(na_reg[J - 1])}, {op('rq')}")
            release(na_reg[J - 1])
            nu_reg[J - 1] = t
        # m1(J-2): na(J-2) = nabove[J-2] * a[J-2]
        if J - 2 >= 1:
            t = alloc()
            gap.append(f"v_mul_f64 {op(t)}, {op(f'nab[{J - 2}]')}, {op(f'a[{J - 2}]')}")
            na_reg[J - 2] = t
        while len(gap) < 2 and J < n - 1 and False:
            gap.append("s_nop 0")
        lines.extend(gap)
        if J < n - 1 or True:
            # pad to two instructions between consecutive DPP levels / after y's producer
            pad = 2 - len(gap)
            if pad == 1:
                lines.append("s_nop 0")
            elif pad == 2:
                lines.append("s_nop 1")
        lines.append(f"v_fmac_f64_dpp {op('y')}, {op('y')}, {op(nu_reg[J])} " + DPP.format(k=J))
        release(nu_reg[J])
    s.append(f"""    // Back-substitution on the scaled right-hand side y = r / Q_ii: level J adds nu[J] x_J to the
    // lanes above J, x_J = [lane J].y being final by then (D == S0 again; lane J adds 0).
    // nu[J] = (nabove[J] a[J]) rq; nu_top = nu[{n - 1}] and na_next = nabove[{n - 2}] a[{n - 2}] come from
    // the caller, the rest is computed in the two slots the DPP hazard needs between levels.
    static __device__ __forceinline__ void backsub(double &y, double rq, double nu_top, double na_next,
                                                   const double (&a)[{n}], const double (&nab)[{n}])
    {{
        double t0, t1, t2, t3;
""")
    body = "\n".join(f'            "{l}\\n"' for l in lines)
    o = '"+v"(y), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)'
    i = ", ".join(f'"v"({x})' for x in ins)
    s.append(f"        asm volatile(\n{body}\n            : {o}\n            : {i});\n")
    s.append("    }\n};\n\n")
    return "".join(s)


HEADER = '''// swimmer_row_fused.h -- GENERATED by scripts/gen_row_fused.py; do not edit by hand.
//
// Inline-asm blocks of the row kernel (swimmer_row.h) built on gfx950's
//     v_fmac_f64_dpp D, S0, S1 row_newbcast:k      D += [lane k of the 16-lane row].S0 * S1
// the only DP-ALU instruction besides v_mov_b64 that takes a DPP operand
// (scripts/ubench/dpp64_fused_test.hip).  It replaces "64-bit DPP move + FMA" pairs, which
// hipcc never fuses.  The compiler does not see DPP hazards inside inline asm, so each block
// is one asm statement with a fixed instruction order; see the generator for the rules.
#pragma once

namespace sw {

template <int N>
struct RowFused;   // specialised for N = 4 .. 8 below

'''


def render():
    return HEADER + "".join(gen(n) for n in range(4, 9)) + "}  // namespace sw\n"


def main():
    text = render()
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT, len(text.splitlines()), "lines")


if __name__ == "__main__":
    main()
