#!/usr/bin/env python3
"""Generates gmmvi_amd/csrc/subst_asm_gen.h: hand-scheduled gfx950 code for the two triangular substitutions
y = L^-T L^-1 (x - mu) of one sample per lane against one packed component block (layout: common.h Pack<DP>).

Why generated assembly: the component block is wave-uniform, so L is fed through scalar loads (SGPR operands of v_fma);
scalar memory returns out of order, so the only usable wait is s_waitcnt lgkmcnt(0), and the compiler places it right
after each load -- ~50 exposed scalar-cache round trips per pass.  Here the loads are double-buffered by hand: the 16
scalars of chunk c+1 are requested before the 16 FMAs of chunk c are issued and waited for after them.

In place on the DP vector registers: x -> t -> z (forward, rows ascending, columns ascending: the order of
forward_subst_s) -> y (backward, columns descending, rows descending within a column).
"""
import sys

import os

BUF_A, BUF_B, RD0 = 40, 56, 72          # SGPR bases: two 16-dword buffers, 1/diag (DP <= 20)


def gen(dp, part="both"):
    """part: 'fwd' (x -> z), 'bwd' (z -> y) or 'both'."""
    assert dp <= 24 and dp >= 2
    T = dp * (dp - 1) // 2
    MU, RD, LROW, LCOL = 0, dp, 2 * dp, 2 * dp + T
    lines = []
    emit = lines.append
    P = f"%{dp}"                                                   # the pointer operand follows the dp vector operands

    def sload(dst, n, float_off):
        emit(f"s_load_dwordx{n} s[{dst}:{dst + n - 1}], {P}, {hex(4 * float_off)}" if n > 1 else
             f"s_load_dword s{dst}, {P}, {hex(4 * float_off)}")

    def load_span(dst, float_off, count):
        """count <= 16 consecutive floats into s[dst...] with the fewest aligned-width loads (dst is 4-aligned)."""
        done = 0
        while done < count:
            rem = count - done
            n = 16 if rem >= 16 else 8 if rem >= 8 else 4 if rem >= 4 else 2 if rem >= 2 else 1
            while (dst + done) % min(n, 4) != 0:
                n //= 2
            sload(dst + done, n, float_off + done)
            done += n

    # ---- mu into A|B, 1/diag into RD ---------------------------------------------------------------------------
    if part != "bwd":
        load_span(BUF_A, MU, min(dp, 16))
        if dp > 16:
            load_span(BUF_B, MU + 16, dp - 16)
    load_span(RD0, RD, min(dp, 16))
    if dp > 16:
        load_span(RD0 + 16, RD + 16, dp - 16)
    emit("s_waitcnt lgkmcnt(0)")
    if part != "bwd":
        for i in range(dp):
            src = BUF_A + i if i < 16 else BUF_B + i - 16
            emit(f"v_subrev_f32_e32 %{i}, s{src}, %{i}")           # x_i - mu_i
        emit(f"v_mul_f32_e32 %0, s{RD0}, %0")                     # z_0

    # ---- forward: stream LROW ascending in chunks of 16 --------------------------------------------------------
    def stream(n_elems, base_off, descending, consume):
        nch = (n_elems + 15) // 16
        def span(c):                                               # (first float offset, count, first stream index)
            lo = 16 * c
            cnt = min(16, n_elems - lo)
            if not descending:
                return base_off + lo, cnt, lo
            return base_off + n_elems - lo - cnt, cnt, lo          # memory window of stream indices lo..lo+cnt-1
        bufs = (BUF_A, BUF_B)
        off, cnt, _ = span(0)
        load_span(bufs[0], off, cnt)
        emit("s_waitcnt lgkmcnt(0)")
        for c in range(nch):
            if c + 1 < nch:
                off, cnt, _ = span(c + 1)
                load_span(bufs[(c + 1) & 1], off, cnt)
            off, cnt, lo = span(c)
            for e in range(cnt):
                # stream index lo + e; in a descending stream it is memory element (window end - e)
                reg = bufs[c & 1] + (e if not descending else cnt - 1 - e)
                consume(lo + e, reg)
            if c + 1 < nch:
                emit("s_waitcnt lgkmcnt(0)")

    rows = [(i, j) for i in range(1, dp) for j in range(i)]       # LROW order
    def fwd(idx, reg):
        i, j = rows[idx]
        emit(f"v_fma_f32 %{i}, -s{reg}, %{j}, %{i}")
        if j == i - 1:
            emit(f"v_mul_f32_e32 %{i}, s{RD0 + i}, %{i}")
    if part != "bwd":
        stream(T, LROW, False, fwd)
    if part == "fwd":
        return lines

    # ---- backward: LCOL (column i holds rows i+1..dp-1, ascending) read from its end ----------------------------
    cols = [(i, j) for i in range(dp - 1) for j in range(i + 1, dp)]   # memory order of LCOL
    emit(f"v_mul_f32_e32 %{dp - 1}, s{RD0 + dp - 1}, %{dp - 1}")       # y_{dp-1}
    def bwd(idx, reg):
        i, j = cols[T - 1 - idx]
        emit(f"v_fma_f32 %{i}, -s{reg}, %{j}, %{i}")
        if j == i + 1:
            emit(f"v_mul_f32_e32 %{i}, s{RD0 + i}, %{i}")
    stream(T, LCOL, True, bwd)
    return lines


PK_CH = int(os.environ.get("PK_CH", "16")) if "os" in dir() else 16


def gen_pk(dp, part="both"):
    """Two samples per lane (float2 operands, v_pk_fma_f32): every scalar of L feeds two FMAs in ONE instruction -- the only
    way to the packed fp32 rate of the vector unit.  fwd + bwd in place, same element order as gen()."""
    assert 2 <= dp <= 24
    CH = PK_CH
    A0, B0, R0 = (40, 56, 72) if CH == 16 else ((32, 56, 80) if CH == 24 else (16, 48, 80))
    T = dp * (dp - 1) // 2
    MU, RD, LROW, LCOL = 0, dp, 2 * dp, 2 * dp + T
    lines = []
    emit = lines.append
    P = f"%{dp}"

    def sload(dst, n, float_off):
        emit(f"s_load_dwordx{n} s[{dst}:{dst + n - 1}], {P}, {hex(4 * float_off)}" if n > 1 else
             f"s_load_dword s{dst}, {P}, {hex(4 * float_off)}")

    def load_span(dst, float_off, count):
        done = 0
        while done < count:
            rem = count - done
            n = 16 if rem >= 16 else 8 if rem >= 8 else 4 if rem >= 4 else 2 if rem >= 2 else 1
            while (dst + done) % min(n, 4) != 0:
                n //= 2
            sload(dst + done, n, float_off + done)
            done += n

    def spair(reg):                                   # (aligned SGPR pair holding reg, which half)
        return f"s[{reg & ~1}:{(reg & ~1) + 1}]", reg & 1

    def pk_fma(i, reg, j):                            # v_i -= s_reg * v_j   (both samples)
        sp, h = spair(reg)
        emit(f"v_pk_fma_f32 %{i}, {sp}, %{j}, %{i} op_sel:[{h},0,0] op_sel_hi:[{h},1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]")

    def pk_mul(i, reg):                               # v_i *= s_reg
        sp, h = spair(reg)
        emit(f"v_pk_mul_f32 %{i}, %{i}, {sp} op_sel:[0,{h}] op_sel_hi:[1,{h}]")

    def pk_sub(i, reg):                               # v_i -= s_reg
        sp, h = spair(reg)
        emit(f"v_pk_add_f32 %{i}, %{i}, {sp} op_sel:[0,{h}] op_sel_hi:[1,{h}] neg_lo:[0,1] neg_hi:[0,1]")

    load_span(A0, MU, dp)                           # mu transits through buffer A (dp <= 20 <= CH + slack)
    load_span(R0, RD, dp)
    emit("s_waitcnt lgkmcnt(0)")
    for i in range(dp):
        pk_sub(i, A0 + i)
    pk_mul(0, R0)

    def stream(n_elems, base_off, descending, consume):
        nch = (n_elems + CH - 1) // CH
        def span(c):
            lo = CH * c
            cnt = min(CH, n_elems - lo)
            if not descending:
                return base_off + lo, cnt, lo
            return base_off + n_elems - lo - cnt, cnt, lo
        bufs = (A0, B0)
        off, cnt, _ = span(0)
        load_span(bufs[0], off, cnt)
        emit("s_waitcnt lgkmcnt(0)")
        for c in range(nch):
            if c + 1 < nch:
                off, cnt, _ = span(c + 1)
                load_span(bufs[(c + 1) & 1], off, cnt)
            off, cnt, lo = span(c)
            for e in range(cnt):
                reg = bufs[c & 1] + (e if not descending else cnt - 1 - e)
                consume(lo + e, reg)
            if c + 1 < nch:
                emit("s_waitcnt lgkmcnt(0)")

    rows = [(i, j) for i in range(1, dp) for j in range(i)]
    def fwd(idx, reg):
        i, j = rows[idx]
        pk_fma(i, reg, j)
        if j == i - 1:
            pk_mul(i, R0 + i)
    stream(T, LROW, False, fwd)
    if part == "fwd":
        return lines
    cols = [(i, j) for i in range(dp - 1) for j in range(i + 1, dp)]
    pk_mul(dp - 1, R0 + dp - 1)
    def bwd(idx, reg):
        i, j = cols[T - 1 - idx]
        pk_fma(i, reg, j)
        if j == i + 1:
            pk_mul(i, R0 + i)
    stream(T, LCOL, True, bwd)
    return lines


def main(out):
    w = []
    w.append("// GENERATED by tools/gen_subst_asm.py -- do not edit.  See that file for the scheme.")
    w.append("#pragma once")
    w.append("#include <hip/hip_runtime.h>")
    w.append("")
    w.append("template <int DP> struct SubstAsm { static constexpr bool available = false; };")
    for dp in (10, 12, 16, 20, 24):
        ops = ", ".join(f'"+v"(v[{i}])' for i in range(dp))
        clob = ", ".join(f'"s{r}"' for r in range(BUF_A, RD0 + 24))
        w.append("")
        w.append(f"template <> struct SubstAsm<{dp}> {{")
        w.append("    static constexpr bool available = true;")
        for name, part, doc in (("run", "both", "v = x  ->  v = Sigma^-1 (x - mu) = L^-T L^-1 (x - mu)"),
                                ("forward", "fwd", "v = x  ->  v = z = L^-1 (x - mu)"),
                                ("backward", "bwd", "v = z  ->  v = L^-T z")):
            w.append(f"    // {doc}   (one sample per lane, block at P)")
            w.append(f"    static __device__ __forceinline__ void {name}(const float* P, float (&v)[{dp}]) {{")
            w.append("        asm volatile(")
            for ln in gen(dp, part):
                w.append(f'            "{ln}\\n"')
            w.append(f"            : {ops}")
            w.append('            : "s"(P)')
            w.append(f"            : {clob});")
            w.append("    }")
        w.append("};")
    w.append("")
    w.append("typedef float gmmvi_f32x2 __attribute__((ext_vector_type(2)));")
    w.append("template <int DP> struct SubstAsmPk { static constexpr bool available = false; };")
    for dp in (10, 12, 16, 20, 24):
        ops = ", ".join(f'"+v"(v[{i}])' for i in range(dp))
        lo_c, hi_c = (40, 96) if PK_CH == 16 else ((32, 100) if PK_CH == 24 else (16, 100))
        clob = ", ".join(f'"s{r}"' for r in range(lo_c, hi_c))
        w.append("")
        w.append(f"template <> struct SubstAsmPk<{dp}> {{")
        w.append("    static constexpr bool available = true;")
        for name, part, doc in (("run", "both", "Sigma^-1 (x - mu)"), ("forward", "fwd", "z = L^-1 (x - mu)")):
            w.append(f"    // v[i] = (x_i of sample A, x_i of sample B)  ->  {doc} of both, one v_pk_fma_f32 per element of L")
            w.append(f"    static __device__ __forceinline__ void {name}(const float* P, gmmvi_f32x2 (&v)[{dp}]) {{")
            w.append("        asm volatile(")
            for ln in gen_pk(dp, part):
                w.append(f'            "{ln}\\n"')
            w.append(f"            : {ops}")
            w.append('            : "s"(P)')
            w.append(f"            : {clob});")
            w.append("    }")
        w.append("};")
    open(out, "w").write("\n".join(w) + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gmmvi_amd/csrc/subst_asm_gen.h")
