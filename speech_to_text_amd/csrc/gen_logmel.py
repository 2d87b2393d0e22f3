#!/usr/bin/env python3
"""Generates the straight-line per-frame log-mel codelet for the HIP frontend.

One GPU lane computes one 400-sample frame entirely in registers.  The frame is
viewed as two interleaved real sequences e[n] = x[2n], o[n] = x[2n+1]
(n = 0..199) held in the two halves of a packed-float register pair, so that a
single real-input 200-point DFT program runs on both at once with packed
arithmetic and wave-uniform scalar constants only:

    X[k]       = E[k] + W400^k O[k]             k = 0..100
    |X[200-k]| = |E[k] - W400^k O[k]|           k = 0..99

The 200-point real DFT is radix 8 x 25 (25 = 5 x 5): eight real 25-point DFTs
over n2 (n = n1 + 8 n2), a twiddle, then 8-point complex DFTs over n1 for the
13 residues k2 = 0..12; Hermitian symmetry supplies the rest.

The program is built as an SSA graph over real values with zero/sign folding,
common-subexpression elimination and dead-code elimination, checked against
numpy's FFT, and written out as C++ on ``v2f`` (ext_vector_type(2) float).
Every constant (twiddles, window, mel weights) is read from one float table so
that the compiler fetches them with batched scalar loads.

Outputs (next to this file):
    logmel_codelet.inc   device code
    logmel_tables.h      host-side description of the constant table
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from speech_to_text_amd.audio.mel_basis import mel_filterbank  # noqa: E402

N_FFT = 400
NH = 200          # length of each interleaved real sequence


# --------------------------------------------------------------------------
# SSA graph over real values
# --------------------------------------------------------------------------
class Graph:
    def __init__(self):
        self.nodes = []          # (op, a, b, const_index)
        self.cse = {}
        self.consts = []         # table of positive constants
        self.const_ix = {}

    def const(self, value):
        key = float(np.float32(value))
        if key not in self.const_ix:
            self.const_ix[key] = len(self.consts)
            self.consts.append(key)
        return self.const_ix[key]

    def node(self, op, a, b=None, c=None):
        key = (op, a, b, c)
        if key not in self.cse:
            self.cse[key] = len(self.nodes)
            self.nodes.append(key)
        return self.cse[key]


class R:
    """A real value: sign * node, or exact zero."""
    __slots__ = ("g", "n", "s")

    def __init__(self, g, n, s=1):
        self.g, self.n, self.s = g, n, s

    @property
    def zero(self):
        return self.n is None

    def __neg__(self):
        return R(self.g, self.n, -self.s)

    def __add__(self, o):
        if self.zero:
            return o
        if o.zero:
            return self
        g = self.g
        if self.n == o.n:
            if self.s == o.s:
                return self * 2.0
            return R(g, None)
        if self.s == o.s:
            a, b = sorted((self.n, o.n))
            return R(g, g.node("add", a, b), self.s)
        if self.s > 0:
            return R(g, g.node("sub", self.n, o.n), 1)
        return R(g, g.node("sub", o.n, self.n), 1)

    def __sub__(self, o):
        return self + (-o)

    def __mul__(self, c):
        c = float(c)
        if self.zero or c == 0.0:
            return R(self.g, None)
        if abs(c) == 1.0:
            return R(self.g, self.n, self.s * (1 if c > 0 else -1))
        g = self.g
        return R(g, g.node("mul", self.n, None, g.const(abs(c))), self.s * (1 if c > 0 else -1))


class C:
    """A complex value made of two R."""
    __slots__ = ("re", "im")

    def __init__(self, re, im):
        self.re, self.im = re, im

    def __add__(self, o):
        return C(self.re + o.re, self.im + o.im)

    def __sub__(self, o):
        return C(self.re - o.re, self.im - o.im)

    def scale(self, c):
        return C(self.re * c, self.im * c)

    def mul_const(self, w):
        c, d = _snap(w.real), _snap(w.imag)
        return C(self.re * c - self.im * d, self.re * d + self.im * c)

    def conj(self):
        return C(self.re, -self.im)

    def mul_neg_i(self):          # (-i) * z
        return C(self.im, -self.re)


def _snap(v):
    for t in (0.0, 1.0, -1.0):
        if abs(v - t) < 1e-15:
            return t
    return v


def _w(n, k):
    return complex(math.cos(2 * math.pi * k / n), -math.sin(2 * math.pi * k / n))


# --------------------------------------------------------------------------
# DFT building blocks
# --------------------------------------------------------------------------
def bf2(x):
    return [x[0] + x[1], x[0] - x[1]]


def bf4(x):
    t0, t1 = x[0] + x[2], x[0] - x[2]
    t2, t3 = x[1] + x[3], (x[1] - x[3]).mul_neg_i()
    return [t0 + t2, t1 + t3, t0 - t2, t1 - t3]


_C5A = (math.cos(2 * math.pi / 5) + math.cos(4 * math.pi / 5)) / 2     # -0.25
_C5B = (math.cos(2 * math.pi / 5) - math.cos(4 * math.pi / 5)) / 2
_S51 = math.sin(2 * math.pi / 5)
_S52 = math.sin(4 * math.pi / 5)


def bf5(x):
    t1, t2 = x[1] + x[4], x[2] + x[3]
    t3, t4 = x[1] - x[4], x[2] - x[3]
    t5 = t1 + t2
    y0 = x[0] + t5
    m0 = x[0] + t5.scale(_C5A)
    m1 = (t1 - t2).scale(_C5B)
    s1, s2 = m0 + m1, m0 - m1
    m3 = (t3.scale(_S51) + t4.scale(_S52)).mul_neg_i()
    m4 = (t3.scale(_S52) - t4.scale(_S51)).mul_neg_i()
    return [y0, s1 + m3, s2 + m4, s2 - m4, s1 - m3]


def dft(x):
    """Complex DFT, natural order in and out (decimation in time)."""
    n = len(x)
    if n == 1:
        return list(x)
    if n == 2:
        return bf2(x)
    if n == 4:
        return bf4(x)
    if n == 5:
        return bf5(x)
    r = 2 if n % 2 == 0 else 5
    m = n // r
    subs = [dft(x[j::r]) for j in range(r)]
    out = [None] * n
    for k2 in range(m):
        t = [subs[j][k2].mul_const(_w(n, j * k2)) for j in range(r)]
        y = dft(t)
        for q in range(r):
            out[k2 + m * q] = y[q]
    return out


def rdft(x, radices):
    """DFT of a real sequence: returns bins 0..n//2.  ``radices`` lists the
    decimation radix used at each level, outermost first."""
    n = len(x)
    g = x[0].g
    if n == 1:
        return [C(x[0], R(g, None))]
    r = radices[0]
    m = n // r
    subs = [rdft(x[j::r], radices[1:]) for j in range(r)]

    def sub_bin(j, k2):
        return subs[j][k2] if k2 <= m // 2 else subs[j][m - k2].conj()

    fam = {}
    for k2 in range(m // 2 + 1):
        t = [sub_bin(j, k2).mul_const(_w(n, j * k2)) for j in range(r)]
        fam[k2] = dft(t)
    out = []
    for k in range(n // 2 + 1):
        k2 = k % m
        if k2 <= m // 2:
            out.append(fam[k2][k // m])
        else:
            kk = n - k
            out.append(fam[kk % m][kk // m].conj())
    return out


# --------------------------------------------------------------------------
# build + check + emit
# --------------------------------------------------------------------------
def build_fft():
    """Top level written out (radix 8 over n1) so that the two passes are visible."""
    g = Graph()
    xs = [R(g, g.node("in", i)) for i in range(NH)]
    r, m, n = 8, 25, NH
    subs = [rdft(xs[j::r], [5, 5]) for j in range(r)]

    def sub_bin(j, k2):
        return subs[j][k2] if k2 <= m // 2 else subs[j][m - k2].conj()

    fams = {}
    for k2 in range(m // 2 + 1):
        t = [sub_bin(j, k2).mul_const(_w(n, j * k2)) for j in range(r)]
        fams[k2] = dft(t)
    bins = []
    for k in range(n // 2 + 1):
        k2 = k % m
        if k2 <= m // 2:
            bins.append(fams[k2][k // m])
        else:
            kk = n - k
            bins.append(fams[kk % m][kk // m].conj())
    return g, bins, subs, fams


def live_nodes(g, roots):
    live = set()
    stack = [r for r in roots if r is not None]
    while stack:
        n = stack.pop()
        if n in live:
            continue
        live.add(n)
        op, a, b, _ = g.nodes[n]
        if op != "in":
            stack.append(a)
            if b is not None:
                stack.append(b)
    return live


def evaluate(g, live, inputs):
    """inputs: [NH, ...] float64 array -> dict node -> value."""
    val = {}
    for n, (op, a, b, c) in enumerate(g.nodes):
        if n not in live:
            continue
        if op == "in":
            val[n] = inputs[a]
        elif op == "add":
            val[n] = val[a] + val[b]
        elif op == "sub":
            val[n] = val[a] - val[b]
        elif op == "mul":
            val[n] = val[a] * g.consts[c]
    return val


def self_check(g, bins, live):
    rng = np.random.default_rng(1)
    x = rng.standard_normal(NH)
    val = evaluate(g, live, x)

    def get(r):
        return 0.0 if r.zero else r.s * val[r.n]
    got = np.array([complex(get(b.re), get(b.im)) for b in bins])
    ref = np.fft.rfft(x)
    err = np.abs(got - ref).max()
    assert err < 2e-5, "generated DFT disagrees with numpy: %g" % err
    return err


def rname(r):
    if r.zero:
        return "v2f{0.f, 0.f}"
    return ("-" if r.s < 0 else "") + "t%d" % r.n


def mel_structure(n_mels):
    """[(bin, filter)] of the non-zeros of the Whisper filterbank, per filter in
    ascending bin order (the summation order of a dense matmul row)."""
    fb = mel_filterbank(16000, N_FFT, n_mels, 0.0, 8000.0)
    pairs = []
    for m in range(n_mels):
        ks = np.nonzero(fb[:, m])[0]
        assert len(ks) > 0 and (np.diff(ks) == 1).all()
        for k in ks:
            pairs.append((int(k), m))
    return pairs


def park_name(slot):
    return "ag[%d]" % slot


def main():
    g, bins, subs, fams = build_fft()
    roots = []
    for b in bins:
        roots += [b.re.n, b.im.n]
    live = live_nodes(g, roots)
    err = self_check(g, bins, live)

    # ---- pass A: the eight 25-point real sub-DFTs; their outputs are the boundary
    boundary = {}            # node -> ("ag", slot) | ("sv", slot)
    sub_nodes = []
    n_ag = n_sv = 0
    for j in range(8):
        rts = []
        for b in subs[j]:
            rts += [b.re.n, b.im.n]
        nodes = sorted(live_nodes(g, rts) & live)
        sub_nodes.append(nodes)
        for n in sorted(set(r for r in rts if r is not None) & live):
            if n in boundary:
                continue
            if j < N_SUBS_IN_AGPR:
                boundary[n] = ("ag", n_ag)
                n_ag += 2
            else:
                boundary[n] = ("sv", n_sv)
                n_sv += 1
    pass_a_nodes = set(n for nodes in sub_nodes for n in nodes)

    n_fft_consts_holder = [None]

    out = []
    emit = out.append

    def emit_node(n, names):
        op, a, b, c = g.nodes[n]
        if op == "add":
            emit("  const v2f t%d = %s + %s;" % (n, names(a), names(b)))
        elif op == "sub":
            emit("  const v2f t%d = %s - %s;" % (n, names(a), names(b)))
        elif op == "mul":
            emit("  const v2f t%d = %s * ct[%d];" % (n, names(a), OFF_FFT + c))

    # constant table layout (floats)
    OFF_FFT = 0
    body_a = []
    emit = body_a.append

    def emit_ins(j, loader):
        """Window multiplies of sub-DFT j.  loader(n2) -> C++ expression of the raw v2f pair."""
        for n in sub_nodes[j]:
            if g.nodes[n][0] != "in":
                continue
            a = g.nodes[n][1]
            n2 = (a - j) // 8
            # the window table is stored in the order pass A consumes it (sub-DFT j, then
            # input n2), so one sub-DFT's 50 values are a few wide scalar loads
            pos = 2 * (25 * (a % 8) + a // 8)
            emit("  const v2f t%d = %s * v2f{ct[OFF_WIN + %d], ct[OFF_WIN + %d]};"
                 % (n, loader(n2), pos, pos + 1))

    def emit_ops(j):
        nodes = sub_nodes[j]
        for n in nodes:
            if g.nodes[n][0] != "in":
                emit(_fmt_node(g, n, OFF_FFT))
        for n in nodes:
            if n in boundary:
                kind, slot = boundary[n]
                if kind == "ag":
                    emit("  ag[%d] = park(t%d.x); ag[%d] = park(t%d.y);" % (slot, n, slot + 1, n))
                else:
                    emit("  sv[%d] = t%d;" % (slot, n))

    def lds_off(nn):
        return (nn // 160) * (164 * 4) + (nn % 160) * 4

    def emit_loads(r):
        item = PASS_A_ORDER[r]
        if isinstance(item, tuple):
            for n2 in range(25):
                emit("  const v4f q%d_%d = *(const v4f*)(base + %d);"
                     % (r, n2, lds_off(2 * (item[0] + 8 * n2))))
        else:
            for n2 in range(25):
                emit("  const v2f q%d_%d = *(const v2f*)(base + %d);"
                     % (r, n2, lds_off(2 * (item + 8 * n2))))

    # LM_PREFETCH_A=1 issues the LDS reads of a region while the previous region's last sub-DFT
    # is still in its butterflies (its inputs are consumed by then, so the registers are free).
    # Measured null (166 vs 167 us per hour of audio): LDS reads and scalar loads share one
    # counter, and the wait for the next window values drains the prefetch anyway.
    prefetch = os.environ.get("LM_PREFETCH_A", "0") == "1"
    emit_loads(0)
    for r, item in enumerate(PASS_A_ORDER):
        if isinstance(item, tuple):
            jj = item[0]
            units = [(jj, lambda n2, r=r: "v2f{q%d_%d.x, q%d_%d.y}" % (r, n2, r, n2)),
                     (jj + 1, lambda n2, r=r: "v2f{q%d_%d.z, q%d_%d.w}" % (r, n2, r, n2))]
        else:
            units = [(item, lambda n2, r=r: "q%d_%d" % (r, n2))]
        for u, (j, loader) in enumerate(units):
            emit("  // sub-DFT %d" % j)
            emit_ins(j, loader)
            if u == len(units) - 1 and r + 1 < len(PASS_A_ORDER):
                if prefetch:
                    emit("  __builtin_amdgcn_sched_barrier(0);")
                    emit_loads(r + 1)
                    emit("  __builtin_amdgcn_sched_barrier(0);")
                    emit_ops(j)
                else:
                    emit_ops(j)
                    emit("  __builtin_amdgcn_sched_barrier(0);")
                    emit_loads(r + 1)
            else:
                emit_ops(j)
            emit("  __builtin_amdgcn_sched_barrier(0);")

    # ---- pass B: families k2 = 0..12
    body_b = []
    emit = body_b.append
    fam_of_bin = {}
    for k in range(101):
        k2 = k % 25
        fam_of_bin.setdefault(min(k2, 25 - k2), []).append(k)
    emitted = set(pass_a_nodes)
    for k2 in range(13):
        rts = []
        for k in fam_of_bin[k2]:
            rts += [bins[k].re.n, bins[k].im.n]
        nodes = sorted((live_nodes(g, rts) & live) - emitted)
        emitted |= set(nodes)
        used_boundary = set()
        for n in nodes:
            op, a, b, c = g.nodes[n]
            for x in (a, b):
                if x is not None and x in boundary:
                    used_boundary.add(x)
        for k in fam_of_bin[k2]:
            for r in (bins[k].re, bins[k].im):
                if not r.zero and r.n in boundary:
                    used_boundary.add(r.n)
        emit("  {   // family k2 = %d: bins %s" % (k2, fam_of_bin[k2]))
        for n in sorted(used_boundary):
            kind, slot = boundary[n]
            if kind == "ag":
                emit("    const v2f t%d = v2f{unpark(ag[%d]), unpark(ag[%d])};" % (n, slot, slot + 1))
            else:
                emit("    const v2f t%d = sv[%d];" % (n, slot))
        for n in nodes:
            emit("  " + _fmt_node(g, n, OFF_FFT))
        # X[k] = E + W^k O and |X[200-k]| = |E - W^k O| with W = exp(-2 pi i / 400); br/bi hold
        # (E, O) in their two halves and cs = (cos, sin) of 2 pi k / 400.  Six packed
        # instructions per bin, with operand-half selects and sign modifiers chosen by hand:
        #   t  = (c O.re, s O.re)
        #   T  = (s O.im + t.lo, c O.im - t.hi)            = W^k O
        #   R  = (E.re + T.re, E.re - T.re),  I = (E.im + T.im, E.im - T.im)
        #   M  = R R + I I                                 = (|X[k]|^2, |X[200-k]|^2)
        # Two bins share one asm statement with their instructions interleaved: a packed result
        # may not be consumed by the very next instruction, and the compiler does not pad
        # hazards inside (or right after) an asm statement.
        def neg(mask3):
            return "[%s]" % ",".join(str(int(x)) for x in mask3)

        def bin_instrs(tag, sr, si):
            nr, ni = sr < 0, si < 0
            return [
                "v_pk_mul_f32 %%[t%s], %%[cs%s], %%[br%s] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:%s neg_hi:%s"
                % (tag, tag, tag, neg((0, nr)), neg((0, nr))),
                "v_pk_fma_f32 %%[t%s], %%[cs%s], %%[bi%s], %%[t%s] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:%s neg_hi:%s"
                % (tag, tag, tag, tag, neg((0, ni, 0)), neg((0, ni, 1))),
                "v_pk_add_f32 %%[m%s], %%[br%s], %%[t%s] op_sel:[0,0] op_sel_hi:[0,0] neg_lo:%s neg_hi:%s"
                % (tag, tag, tag, neg((nr, 0)), neg((nr, 1))),
                "v_pk_add_f32 %%[i%s], %%[bi%s], %%[t%s] op_sel:[0,1] op_sel_hi:[0,1] neg_lo:%s neg_hi:%s"
                % (tag, tag, tag, neg((ni, 0)), neg((ni, 1))),
                "v_pk_mul_f32 %%[m%s], %%[m%s], %%[m%s]" % (tag, tag, tag),
                "v_pk_fma_f32 %%[m%s], %%[i%s], %%[i%s], %%[m%s]" % (tag, tag, tag, tag),
            ]

        def val(r):
            return "v2f{0.f, 0.f}" if r.zero else "t%d" % r.n

        ks = fam_of_bin[k2]
        for g0 in range(0, len(ks), 2):
            grp = ks[g0:g0 + 2]
            lines = ["s_nop 0"]      # the inputs may come straight from a packed instruction
            seqs = [bin_instrs("ab"[i], 1 if bins[k].re.zero else bins[k].re.s,
                               1 if bins[k].im.zero else bins[k].im.s) for i, k in enumerate(grp)]
            if len(grp) == 2:
                for x, y in zip(*seqs):
                    lines += [x, y]
                lines.append("s_nop 0")
            else:
                for x in seqs[0]:
                    lines += [x, "s_nop 0"]
            emit("    {")
            outs, ins = [], []
            for i, k in enumerate(grp):
                tag = "ab"[i]
                emit("      v2f m%s, t%s, i%s;" % (tag, tag, tag))
                outs += ['[m%s] "=&v"(m%s)' % (tag, tag), '[t%s] "=&v"(t%s)' % (tag, tag),
                         '[i%s] "=&v"(i%s)' % (tag, tag)]
                ins += ['[cs%s] "s"(v2f{ct[OFF_TW + %d], ct[OFF_TW + %d]})' % (tag, 2 * k, 2 * k + 1),
                        '[br%s] "v"(%s)' % (tag, val(bins[k].re)), '[bi%s] "v"(%s)' % (tag, val(bins[k].im))]
            emit('      asm("%s"' % "\\n\\t".join(lines))
            emit("          : %s" % ", ".join(outs))
            emit("          : %s);" % ", ".join(ins))
            for i, k in enumerate(grp):
                tag = "ab"[i]
                emit("      mag[%d] = park(m%s.x);" % (k, tag))
                if k < 100:
                    emit("      mag[%d] = park(m%s.y);" % (200 - k, tag))
            emit("    }")
        emit("  }")
        emit("  __builtin_amdgcn_sched_barrier(0);")
        emit("  slot(%d);" % k2)
        if os.environ.get("LM_SLOT_BARRIER", "1") == "1":
            emit("  __builtin_amdgcn_sched_barrier(0);")

    n_fft_consts = len(g.consts)
    off_fft = 0
    off_win = off_fft + n_fft_consts           # 400 window values
    off_tw = off_win + N_FFT                   # (cos, sin) of 2 pi k / 400, k = 0..100
    off_mel80 = off_tw + 2 * 101
    pairs80 = mel_structure(80)
    off_mel128 = off_mel80 + len(pairs80)
    pairs128 = mel_structure(128)
    table_len = off_mel128 + len(pairs128)

    out = []
    emit = out.append
    emit("// GENERATED by gen_logmel.py -- do not edit.")
    emit("// real 200-point DFT on two interleaved sequences: %d live ops, %d constants"
         % (sum(1 for n in live if g.nodes[n][0] != "in"), n_fft_consts))
    emit("// self-check vs numpy rfft (float64 evaluation): max abs err %.3g" % err)
    emit("")
    emit("#define OFF_FFT %d" % off_fft)
    emit("#define OFF_WIN %d" % off_win)
    emit("#define OFF_TW %d" % off_tw)
    emit("#define OFF_MEL80 %d" % off_mel80)
    emit("#define OFF_MEL128 %d" % off_mel128)
    emit("#define LOGMEL_TABLE_LEN %d" % table_len)
    emit("#define LOGMEL_N_AG %d" % max(n_ag, 1))
    emit("#define LOGMEL_N_SV %d" % max(n_sv, 1))
    emit("")
    emit("// Pass A.  base = this lane's frame start in LDS (rows of 164 floats).  Writes the")
    emit("// eight 25-point sub-DFTs: parked in accumulator registers (ag) or kept (sv).")
    emit("__device__ __forceinline__ void logmel_pass_a(const unsigned char* base, "
         "lm_const_ptr ct, float (&ag)[LOGMEL_N_AG], v2f (&sv)[LOGMEL_N_SV]) {")
    out.extend(body_a)
    emit("}")
    emit("")
    emit("// Pass B: twiddles, 8-point DFTs, E/O combine; mag[k] = |X[k]|^2 (parked), k = 0..200.")
    emit("// slot(k2) is called after family k2 (13 calls): the caller's hook for work to interleave.")
    emit("template <class Slot>")
    emit("__device__ __forceinline__ void logmel_pass_b(lm_const_ptr ct, "
         "const float (&ag)[LOGMEL_N_AG], const v2f (&sv)[LOGMEL_N_SV], float (&mag)[201], "
         "Slot&& slot) {")
    out.extend(body_b)
    emit("}")
    emit("")
    for n_mels, pairs, off in ((80, pairs80, "OFF_MEL80"), (128, pairs128, "OFF_MEL128")):
        fb = mel_filterbank(16000, N_FFT, n_mels, 0.0, 8000.0)
        emit("__device__ __forceinline__ void logmel_project_%d(const float (&mag)[201], "
             "lm_const_ptr ct, float (&mel)[%d]) {" % (n_mels, n_mels))
        i = 0
        defined = set()
        for m in range(n_mels):
            terms = []
            while i < len(pairs) and pairs[i][1] == m:
                k = pairs[i][0]
                if k not in defined:
                    # the square root is taken after the value has left the accumulator file: an
                    # asm statement must not read a transcendental result directly (the compiler
                    # pads that hazard only for its own instructions)
                    emit("  const float m%d = __builtin_amdgcn_sqrtf(unpark(mag[%d]));" % (k, k))
                    defined.add(k)
                if LITERALS & 4:
                    terms.append("%s * m%d" % (_lit(fb[k, m]), k))
                else:
                    terms.append("ct[%s + %d] * m%d" % (off, i, k))
                i += 1
            emit("  mel[%d] = %s;" % (m, " + ".join(terms)))
        emit("}")
        emit("")
    with open(os.path.join(HERE, "logmel_codelet.inc"), "w") as f:
        f.write("\n".join(out))

    # host-side table description
    h = []
    h.append("// GENERATED by gen_logmel.py -- do not edit.")
    h.append("#pragma once")
    h.append("#define LOGMEL_OFF_FFT %d" % off_fft)
    h.append("#define LOGMEL_OFF_WIN %d" % off_win)
    h.append("#define LOGMEL_OFF_TW %d" % off_tw)
    h.append("#define LOGMEL_OFF_MEL80 %d" % off_mel80)
    h.append("#define LOGMEL_OFF_MEL128 %d" % off_mel128)
    h.append("#define LOGMEL_TABLE_LEN %d" % table_len)
    h.append("#define LOGMEL_N_FFT_CONSTS %d" % n_fft_consts)
    h.append("static const float logmel_fft_consts[%d] = {" % n_fft_consts)
    h.append(",\n".join("  %.9ef" % v for v in g.consts))
    h.append("};")
    for n_mels, pairs in ((80, pairs80), (128, pairs128)):
        h.append("#define LOGMEL_NNZ_%d %d" % (n_mels, len(pairs)))
        h.append("// (bin, filter) of every weight slot, in table order")
        h.append("static const unsigned short logmel_nz_%d[%d][2] = {" % (n_mels, len(pairs)))
        h.append(",\n".join("  {%d, %d}" % p for p in pairs))
        h.append("};")
    with open(os.path.join(HERE, "logmel_tables.h"), "w") as f:
        f.write("\n".join(h) + "\n")
    print("fft ops: %d  consts: %d  table floats: %d  ag %d  sv %d  self-check err %.3g"
          % (sum(1 for n in live if g.nodes[n][0] != "in"), n_fft_consts, table_len, n_ag, n_sv, err))


def _lit(v):
    return "%.9ef" % float(np.float32(v))


def _fmt_node(g, n, off_fft):
    op, a, b, c = g.nodes[n]
    if op == "mul" and LITERALS & 1:
        return "  const v2f t%d = t%d * %s;" % (n, a, _lit(g.consts[c]))
    if op == "add":
        return "  const v2f t%d = t%d + t%d;" % (n, a, b)
    if op == "sub":
        return "  const v2f t%d = t%d - t%d;" % (n, a, b)
    if op == "mul":
        return "  const v2f t%d = t%d * ct[OFF_FFT + %d];" % (n, a, c)
    raise ValueError(op)


N_SUBS_IN_AGPR = 5
HANN = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(400) / 400)).astype(np.float32)   # periodic Hann
LITERALS = int(os.environ.get("LOGMEL_LITERALS", "0"))   # 1 fft constants, 2 twiddles, 4 mel weights
PASS_A_ORDER = [(0,), (2,), (4,), 6, 7]     # (j,) = the pair j, j+1 from one 16-byte load


if __name__ == "__main__":
    main()
