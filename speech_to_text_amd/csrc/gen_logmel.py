#!/usr/bin/env python3
"""Generates the straight-line log-mel codelet in which TWO lanes share one 400-sample frame.

A frame is viewed as two interleaved real sequences e[n] = x[2n], o[n] = x[2n+1] (n = 0..199)
held in the two halves of a packed-float register pair, and that packed sequence s[n] is decimated once more across a lane pair
(lane l and lane l + 32 of a wave):

    lane half p computes the real 100-point DFT  S_p[k] of s[2m + p], m = 0..99   (k = 0..50)
    S[k]       = S_0[k] + W200^k S_1[k]
    S[100 - k] = conj(S_0[k] - W200^k S_1[k])
    X[k]       = E[k] + W400^k O[k],   |X[200-k]| = |E[k] - W400^k O[k]|     (E, O = halves of S)

so that the state a lane carries between the passes is 200 floats instead of 400 and two waves
fit on a SIMD.  The 100-point real DFT is radix 4 x 25 (25 = 5 x 5).  After family k2 of the
second pass has produced S_p[k] for k in {k2, 25-k2, 25+k2, 50-k2}, one v_permlane32_swap per
float leaves (S_0[kappa], S_1[kappa]) in the lower lanes for kappa <= 25 and in the upper lanes
for kappa = 50 - k (>= 25): each lane then finishes its own kappa with per-lane twiddles read
from a small LDS table and ends up with |X| of the four bins kappa, 200-kappa, 100-kappa,
100+kappa.  The mel projection is two programs, one per lane half (filters below / from SPLIT),
run one after the other under the half's execution mask; the magnitudes a half needs but does
not own are exchanged with the same swap instruction first.

Outputs (next to this file):
    logmel_codelet.inc   device code
    logmel_tables.h      host-side description of the tables
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from speech_to_text_amd.audio.mel_basis import mel_filterbank  # noqa: E402

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


N_FFT = 400
NL = 100                   # length of the packed sequence one lane transforms
SPLIT = {80: 48, 128: 64}  # first filter of the upper half's projection program


def build_fft():
    g = Graph()
    xs = [R(g, g.node("in", i)) for i in range(NL)]
    r, m, n = 4, 25, NL
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
    return g, bins, subs


def kappa_slots():
    """[(family, kappa of the lower half, kappa of the upper half or None)] in generation order."""
    slots = [(0, 0, 50), (0, 25, None)]
    for f in range(1, 13):
        slots += [(f, f, 50 - f), (f, 25 - f, 25 + f)]
    return slots


def bins_of(kappa):
    return [kappa, 200 - kappa, 100 - kappa, 100 + kappa]


def twiddles(kappa):
    """(cos, sin) pairs a lane needs for its kappa: W200^kappa, then the E/O twiddle of S[kappa]
    and of conj(S[100 - kappa]) (conjugated, so the same instruction sequence serves both)."""
    a = 2 * math.pi * kappa / 200
    b = 2 * math.pi * kappa / 400
    c = 2 * math.pi * (100 - kappa) / 400
    return [(math.cos(a), math.sin(a)), (math.cos(b), math.sin(b)), (math.cos(c), -math.sin(c))]


def exchange_plan(n_mels):
    """Which magnitude registers the two halves swap before the projection.

    Register mg[4 i + c] holds bin bins_of(kappa)[c] of the half's i-th kappa.  Returns
    (moves, loc_lower, loc_upper, pairs_lower, pairs_upper): moves = [(vdst, src0, copy_vdst,
    copy_src0)] register indices for lm_swap (upper's vdst <-> lower's src0; a copy flag says
    the register is duplicated into a scratch register first because its owner still needs it),
    loc_* = bin -> register name after the moves."""
    fb = mel_filterbank(16000, N_FFT, n_mels, 0.0, 8000.0)
    split = SPLIT[n_mels]
    pairs = [[], []]
    for m in range(n_mels):
        ks = np.nonzero(fb[:, m])[0]
        assert len(ks) > 0 and (np.diff(ks) == 1).all()
        for k in ks:
            pairs[0 if m < split else 1].append((int(k), m))
    need = [sorted({k for k, _ in pairs[h]}) for h in (0, 1)]
    own = [{}, {}]
    for i, (_, kl, ku) in enumerate(kappa_slots()):
        for c, b in enumerate(bins_of(kl)):
            own[0].setdefault(b, 4 * i + c)
        if ku is not None:
            for c, b in enumerate(bins_of(ku)):
                own[1].setdefault(b, 4 * i + c)
    loc = [{b: "mg[%d]" % own[0][b] for b in need[0] if b in own[0]},
           {b: "mg[%d]" % own[1][b] for b in need[1] if b in own[1]}]
    to_lower = [b for b in need[0] if b not in own[0]]       # owned by the upper half
    to_upper = [b for b in need[1] if b not in own[1]]       # owned by the lower half
    used = [set(own[0][b] for b in need[0] if b in own[0]), set(own[1][b] for b in need[1] if b in own[1])]
    # registers whose content the half itself does not need: free to receive
    free = [[r for r in range(104) if r not in used[0]], [r for r in range(104) if r not in used[1]]]
    moves, n_tmp = [], 0
    while to_lower or to_upper:
        a = to_lower.pop(0) if to_lower else None            # upper -> lower
        b = to_upper.pop(0) if to_upper else None            # lower -> upper
        # vdst: the register whose UPPER copy is sent (holds a) and receives b in the upper half
        # src0: the register whose LOWER copy is sent (holds b) and receives a in the lower half
        if a is not None:
            vdst = own[1][a]
            copy_v = vdst in used[1]                          # the upper half needs bin a itself
        else:
            vdst = next((r for r in free[1] if r not in [m[0] for m in moves]), None)
            copy_v = False
        if b is not None:
            src0 = own[0][b]
            copy_s = src0 in used[0]
        else:
            src0 = next((r for r in free[0] if r not in [m[1] for m in moves]), None)
            copy_s = False
        vname, sname = "mg[%s]" % vdst, "mg[%s]" % src0
        if vdst is None:            # nothing free to receive in: a scratch register of its own
            vname = "xt[%d]" % n_tmp
            n_tmp += 1
        if src0 is None:
            sname = "xt[%d]" % n_tmp
            n_tmp += 1
        if copy_v:
            vname = "xt[%d]" % n_tmp
            n_tmp += 1
        if copy_s:
            sname = "xt[%d]" % n_tmp
            n_tmp += 1
        # a register may take part in one move only (its content changes)
        if not copy_v and vdst is not None:
            if vdst in free[1]:
                free[1].remove(vdst)
        if not copy_s and src0 is not None:
            if src0 in free[0]:
                free[0].remove(src0)
        moves.append((vdst, src0, vname if (copy_v or vdst is None) else None, sname if (copy_s or src0 is None) else None))
        if a is not None:
            loc[0][a] = sname                                 # lower finds bin a where it sent from
        if b is not None:
            loc[1][b] = vname
    # sanity: a register that receives must not be needed (under its own name) by the receiver
    for vdst, src0, cv, cs in moves:
        if cs is None:
            assert src0 not in used[0], "lower register %d overwritten while needed" % src0
        if cv is None:
            assert vdst not in used[1], "upper register %d overwritten while needed" % vdst
    assert len({m[0] for m in moves if m[2] is None}) == len([m for m in moves if m[2] is None])
    assert len({m[1] for m in moves if m[3] is None}) == len([m for m in moves if m[3] is None])
    return moves, loc, pairs, n_tmp


def lds_off(nn):
    return (nn // 160) * (164 * 4) + (nn % 160) * 4


def fmt_node(g, n):
    op, a, b, c = g.nodes[n]
    if op == "add":
        return "  const v2f t%d = t%d + t%d;" % (n, a, b)
    if op == "sub":
        return "  const v2f t%d = t%d - t%d;" % (n, a, b)
    if op == "mul":
        return "  const v2f t%d = t%d * ct[OFF_FFT + %d];" % (n, a, c)
    raise ValueError(op)


def rexpr(r):
    if r.zero:
        return "v2f{0.f, 0.f}"
    return ("-" if r.s < 0 else "") + "t%d" % r.n


# --------------------------------------------------------------------------
# numpy emulation of the lane pair (checks the plan and the tables)
# --------------------------------------------------------------------------
def emulate(x, n_mels, hann, fb):
    """x: [400] float64 frame -> mel [n_mels] by the lane-pair dataflow."""
    g, bins, _ = build_fft()
    roots = []
    for b in bins:
        roots += [b.re.n, b.im.n]
    live = live_nodes(g, [r for r in roots if r is not None])
    xw = x * hann
    s = xw[0::2] + 1j * 0          # packed halves handled as two real sequences
    e, o = xw[0::2], xw[1::2]
    S = []
    for p in (0, 1):
        out = []
        for seq in (e, o):
            val = evaluate(g, live, seq[p::2])
            def get(r):
                return 0.0 if r.zero else r.s * val[r.n]
            out.append(np.array([complex(get(b.re), get(b.im)) for b in bins]))
        S.append(out)              # S[p][half e/o][k]
    mg = [np.zeros(104), np.zeros(104)]
    for i, (_, kl, ku) in enumerate(kappa_slots()):
        for h, kap in ((0, kl), (1, ku)):
            if kap is None:
                continue
            tw = twiddles(kap)
            w200 = complex(tw[0][0], -tw[0][1])
            U, V = [], []
            for half in (0, 1):
                T = w200 * S[1][half][kap]
                U.append(S[0][half][kap] + T)
                V.append(S[0][half][kap] - T)
            wu = complex(tw[1][0], -tw[1][1])
            wv = complex(tw[2][0], -tw[2][1])
            mg[h][4 * i + 0] = abs(U[0] + wu * U[1])
            mg[h][4 * i + 1] = abs(U[0] - wu * U[1])
            mg[h][4 * i + 2] = abs(V[0] + wv * V[1])
            mg[h][4 * i + 3] = abs(V[0] - wv * V[1])
    moves, loc, pairs, n_tmp = exchange_plan(n_mels)
    xt = [np.zeros(max(n_tmp, 1)), np.zeros(max(n_tmp, 1))]
    regs = [{"mg": mg[0], "xt": xt[0]}, {"mg": mg[1], "xt": xt[1]}]

    def ref(h, name):
        arr, idx = name[:2], int(name[3:-1])
        return regs[h][arr], idx
    for vdst, src0, cv, cs in moves:
        vname, sname = cv or "mg[%d]" % vdst, cs or "mg[%d]" % src0
        for h in (0, 1):
            if cv and vdst is not None:
                a, i = ref(h, cv); a[i] = regs[h]["mg"][vdst]
            if cs and src0 is not None:
                a, i = ref(h, cs); a[i] = regs[h]["mg"][src0]
        au, iu = ref(1, vname)
        al, il = ref(0, sname)
        au[iu], al[il] = al[il], au[iu]
    mel = np.zeros(n_mels)
    for h in (0, 1):
        for k, m in pairs[h]:
            a, i = ref(h, loc[h][k])
            mel[m] += fb[k, m] * a[i]
    return mel


def self_check():
    hann = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(400) / 400)
    rng = np.random.default_rng(3)
    worst = 0.0
    for n_mels in (80, 128):
        fb = mel_filterbank(16000, N_FFT, n_mels, 0.0, 8000.0).astype(np.float64)
        for _ in range(3):
            x = rng.standard_normal(400)
            want = np.abs(np.fft.rfft(x * hann)) @ fb
            got = emulate(x, n_mels, hann, fb)
            worst = max(worst, float(np.abs(got - want).max() / np.abs(want).max()))
    assert worst < 1e-6, "lane-pair dataflow disagrees with numpy: %g" % worst
    return worst


# --------------------------------------------------------------------------
# emit
# --------------------------------------------------------------------------
def main():
    err = self_check()
    g, bins, subs = build_fft()
    roots = []
    for b in bins:
        roots += [b.re.n, b.im.n]
    live = live_nodes(g, [r for r in roots if r is not None])

    # ---- pass A: four 25-point real sub-DFTs
    boundary, sub_nodes, n_sa = {}, [], 0
    for j in range(4):
        rts = []
        for b in subs[j]:
            rts += [b.re.n, b.im.n]
        rts = [r for r in rts if r is not None]
        nodes = sorted(live_nodes(g, rts) & live)
        sub_nodes.append(nodes)
        for n in sorted(set(rts) & live):
            if n not in boundary:
                boundary[n] = n_sa
                n_sa += 1
    pass_a_nodes = set(n for nodes in sub_nodes for n in nodes)

    body_a = []
    emit = body_a.append
    win_pos = []                 # [j][position in consumption order] -> m2
    for j in range(4):
        emit("  // sub-DFT %d: packed inputs s[2 (j + 4 m2) + p] = frame samples 4 (j + 4 m2) + 2 p, +1;" % j)
        emit("  // every input and its window pair are fetched right before the first use")
        # consumption order of the inputs
        order = []
        for n in sub_nodes[j]:
            op, a, b, c = g.nodes[n]
            if op == "in":
                continue
            for x in (a, b):
                if x is not None and g.nodes[x][0] == "in" and x not in order:
                    order.append(x)
        for n in sub_nodes[j]:           # inputs that are outputs themselves (none expected)
            if g.nodes[n][0] == "in" and n not in order:
                order.append(n)
        assert len(order) == 25
        win_pos.append([(g.nodes[n][1] - j) // 4 for n in order])
        loaded_w, done_in = set(), set()

        def need_input(n):
            if n in done_in:
                return
            done_in.add(n)
            pos = order.index(n)
            m2 = (g.nodes[n][1] - j) // 4
            emit("  const v2f q%d_%d = *(const v2f*)(base + %d);" % (j, m2, lds_off(4 * (j + 4 * m2))))
            wi = pos // 2
            if wi not in loaded_w:
                loaded_w.add(wi)
                if wi < 12:
                    emit("  const v4f w%d_%d = *(const v4f*)(wt + %d);" % (j, wi, (26 * j + 2 * wi) * 8))
                else:
                    emit("  const v2f w%d_%d = *(const v2f*)(wt + %d);" % (j, wi, (26 * j + 2 * wi) * 8))
            if wi == 12:
                w = "w%d_12" % j
            elif pos % 2 == 0:
                w = "v2f{w%d_%d.x, w%d_%d.y}" % (j, wi, j, wi)
            else:
                w = "v2f{w%d_%d.z, w%d_%d.w}" % (j, wi, j, wi)
            emit("  const v2f t%d = q%d_%d * %s;" % (n, j, m2, w))

        for n in sub_nodes[j]:
            op, a, b, c = g.nodes[n]
            if op == "in":
                continue
            for x in (a, b):
                if x is not None and g.nodes[x][0] == "in":
                    need_input(x)
            emit(fmt_node(g, n))
        for n in order:
            need_input(n)
        for n in sub_nodes[j]:
            if n in boundary:
                emit("  sa[%d] = t%d;" % (boundary[n], n))
        # pin the results here: pure arithmetic would otherwise be sunk past the workgroup barrier
        # to its first use, where all four sub-DFTs would then be live at once
        for n in sub_nodes[j]:
            if n in boundary:
                emit("  asm volatile(\"\" : \"+v\"(sa[%d]));" % boundary[n])
        emit("  __builtin_amdgcn_sched_barrier(0);")

    # ---- pass B + combine, family by family
    body_b = []
    emit = body_b.append
    emitted = set(pass_a_nodes)
    slots = kappa_slots()
    for f in range(13):
        ks = sorted({k for k in range(51) if k % 25 in (f, 25 - f)})
        rts = []
        for k in ks:
            rts += [bins[k].re.n, bins[k].im.n]
        rts = [r for r in rts if r is not None]
        nodes = sorted((live_nodes(g, rts) & live) - emitted)
        emitted |= set(nodes)
        used_boundary = set()
        for n in nodes:
            op, a, b, c = g.nodes[n]
            for x in (a, b):
                if x is not None and x in boundary:
                    used_boundary.add(x)
        for k in ks:
            for r in (bins[k].re, bins[k].im):
                if not r.zero and r.n in boundary:
                    used_boundary.add(r.n)
        emit("  {   // family %d: S_p[k] for k in %s" % (f, ks))
        for n in sorted(used_boundary):
            emit("    const v2f t%d = sa[%d];" % (n, boundary[n]))
        for n in nodes:
            emit("  " + fmt_node(g, n))
        mine = [(i, kl, ku) for i, (ff, kl, ku) in enumerate(slots) if ff == f]
        for j, (i, kl, ku) in enumerate(mine):
            # registers a (vdst) = S_p[kl], b (src0) = S_p[ku]; after the swap a = S_0[kappa], b = S_1[kappa]
            kb = ku if ku is not None else kl
            # kappa 25 pairs with itself: the partner's value arrives in a copy
            for comp, ra, rb in (("r", bins[kl].re, bins[kb].re), ("i", bins[kl].im, bins[kb].im)):
                for h in ("x", "y"):
                    emit("    float a%s%s%d = (%s).%s, b%s%s%d = (%s).%s;"
                         % (comp, h, i, rexpr(ra), h, comp, h, i, rexpr(rb), h))
            emit("    lm_swap4(arx%d, brx%d, ary%d, bry%d, aix%d, bix%d, aiy%d, biy%d);" % ((i,) * 8))
            emit("    const v2f ar%d = v2f{arx%d, ary%d}, ai%d = v2f{aix%d, aiy%d};" % (i, i, i, i, i, i))
            emit("    const v2f br%d = v2f{brx%d, bry%d}, bi%d = v2f{bix%d, biy%d};" % (i, i, i, i, i, i))
            # this kappa's three twiddle pairs: 24 bytes at tt + 48 f + 24 j
            o = f * 48 + 24 * j
            emit("    const v2f w%d = *(const v2f*)(tt + %d), cu%d = *(const v2f*)(tt + %d), cv%d = *(const v2f*)(tt + %d);"
                 % (i, o, i, o + 8, i, o + 16))
            emit("    const v2f Tr%d = br%d * w%d.x + bi%d * w%d.y, Ti%d = bi%d * w%d.x - br%d * w%d.y;"
                 % (i, i, i, i, i, i, i, i, i, i))
            emit("    const v2f Ur%d = ar%d + Tr%d, Ui%d = ai%d + Ti%d, Vr%d = ar%d - Tr%d, Vi%d = ai%d - Ti%d;"
                 % (i, i, i, i, i, i, i, i, i, i, i, i))
            emit("    v2f mu%d, mv%d;" % (i, i))
            emit("    lm_combine2(cu%d, Ur%d, Ui%d, mu%d, cv%d, Vr%d, Vi%d, mv%d);" % (i, i, i, i, i, i, i, i))
            emit("    mg[%d] = __builtin_amdgcn_sqrtf(mu%d.x); mg[%d] = __builtin_amdgcn_sqrtf(mu%d.y);"
                 % (4 * i, i, 4 * i + 1, i))
            emit("    mg[%d] = __builtin_amdgcn_sqrtf(mv%d.x); mg[%d] = __builtin_amdgcn_sqrtf(mv%d.y);"
                 % (4 * i + 2, i, 4 * i + 3, i))
            if j == 0:
                emit("    __builtin_amdgcn_sched_barrier(0);")
        emit("  }")
        emit("  __builtin_amdgcn_sched_barrier(0);")
        emit("  slot(%d);" % f)
        emit("  __builtin_amdgcn_sched_barrier(0);")

    # ---- tables
    n_fft_consts = len(g.consts)
    off_fft = 0
    off_mel = {}
    cursor = off_fft + n_fft_consts
    plans = {}
    for n_mels in (80, 128):
        plans[n_mels] = exchange_plan(n_mels)
        off_mel[n_mels] = cursor
        cursor += len(plans[n_mels][2][0]) + len(plans[n_mels][2][1])
    table_len = cursor

    out = []
    emit = out.append
    emit("// GENERATED by gen_logmel.py -- do not edit.")
    emit("// real 100-point DFT per lane on packed (even, odd) pairs: %d live ops, %d constants"
         % (sum(1 for n in live if g.nodes[n][0] != "in"), n_fft_consts))
    emit("// lane-pair dataflow checked against numpy (float64): max rel err %.3g" % err)
    emit("")
    emit("#define OFF_FFT %d" % off_fft)
    emit("#define OFF_MEL80 %d" % off_mel[80])
    emit("#define OFF_MEL128 %d" % off_mel[128])
    emit("#define LOGMEL_TABLE_LEN %d" % table_len)
    emit("#define LOGMEL_N_SA %d" % n_sa)
    emit("#define LOGMEL_WIN_BYTES 1664      // [half][sub-DFT j][26] pairs (w[4m+2p], w[4m+2p+1]), m = j + 4 m2 (m2 = 25 is padding)")
    emit("#define LOGMEL_TW_BYTES 1248       // [half][family][6] (cos, sin)")
    emit("")
    emit("// Pass A.  base = LDS address of this lane's frame start + 8 * half; wt = LDS address of the")
    emit("// half's window table.  Writes the four 25-point sub-DFTs to sa.")
    emit("__device__ __forceinline__ void logmel_pass_a(const unsigned char* base, const unsigned char* wt,")
    emit("                                               lm_const_ptr ct, v2f (&sa)[LOGMEL_N_SA]) {")
    out.extend(body_a)
    emit("}")
    emit("")
    emit("// Pass B, the lane-pair exchange and both combines: mg[4 i + c] = |X[b]| for the c-th bin of")
    emit("// this half's i-th kappa (see kappa_slots in the generator).  tt = LDS address of the half's")
    emit("// twiddle table.  slot(f) is called after family f (13 calls).")
    emit("template <class Slot>")
    emit("__device__ __forceinline__ void logmel_pass_b(lm_const_ptr ct, const unsigned char* tt,")
    emit("                                               const v2f (&sa)[LOGMEL_N_SA], float (&mg)[104], Slot&& slot) {")
    out.extend(body_b)
    emit("}")
    emit("")
    for n_mels in (80, 128):
        moves, loc, pairs, n_tmp = plans[n_mels]
        split = SPLIT[n_mels]
        na = max(split, n_mels - split)
        emit("#define LOGMEL_SPLIT_%d %d" % (n_mels, split))
        emit("#define LOGMEL_NACC_%d %d" % (n_mels, na))
        emit("// acc[i] = filter i (lower half) / filter %d + i (upper half; entries past %d repeat acc[0])"
             % (split, n_mels - split))
        emit("__device__ __forceinline__ void logmel_project_%d(float (&mg)[104], lm_const_ptr ct, bool upper, "
             "float (&acc)[%d]) {" % (n_mels, na))
        emit("  float xt[%d];" % max(n_tmp, 1))
        for vdst, src0, cv, cs in moves:
            if cv:
                emit("  %s = %s;" % (cv, "mg[%d]" % vdst if vdst is not None else "0.f"))
            if cs:
                emit("  %s = %s;" % (cs, "mg[%d]" % src0 if src0 is not None else "0.f"))
        names = [(cv or "mg[%d]" % vdst, cs or "mg[%d]" % src0) for vdst, src0, cv, cs in moves]
        # four swaps per asm statement; a register may appear only once in a statement (each operand
        # of an asm statement is read before and written after it, independently of the others)
        while names:
            grp, rest, seen = [], [], set()
            for n in names:
                if len(grp) < 4 and n[0] not in seen and n[1] not in seen and n[0] != n[1]:
                    grp.append(n)
                    seen.update(n)
                else:
                    rest.append(n)
            # order must be kept between moves that share a register: only take a prefix-safe group
            safe = []
            blocked = set()
            for n in names:
                if n in grp and n[0] not in blocked and n[1] not in blocked:
                    safe.append(n)
                else:
                    blocked.update(n)
            grp = safe
            names = [n for n in names if n not in grp]
            if len(grp) == 4:
                emit("  lm_swap4(%s);" % ", ".join("%s, %s" % n for n in grp))
            else:
                for n in grp:
                    emit("  lm_swap(%s, %s);" % n)
        off = "OFF_MEL%d" % n_mels
        pos = 0
        for h in (0, 1):
            emit("  if (%supper) {" % ("" if h else "!"))
            i = 0
            ps = pairs[h]
            first = 0 if h == 0 else split
            count = split if h == 0 else n_mels - split
            for m in range(first, first + count):
                terms = []
                while i < len(ps) and ps[i][1] == m:
                    terms.append("ct[%s + %d] * %s" % (off, pos, loc[h][ps[i][0]]))
                    i += 1
                    pos += 1
                emit("    acc[%d] = %s;" % (m - first, " + ".join(terms)))
            for m in range(count, na):
                emit("    acc[%d] = acc[0];" % m)
            emit("  }")
        emit("}")
        emit("")
    with open(os.path.join(HERE, "logmel_codelet.inc"), "w") as f:
        f.write("\n".join(out))

    # ---- host-side description
    h = []
    h.append("// GENERATED by gen_logmel.py -- do not edit.")
    h.append("#pragma once")
    h.append("#define LOGMEL_OFF_FFT %d" % off_fft)
    h.append("#define LOGMEL_OFF_MEL80 %d" % off_mel[80])
    h.append("#define LOGMEL_OFF_MEL128 %d" % off_mel[128])
    h.append("#define LOGMEL_TABLE_LEN %d" % table_len)
    h.append("#define LOGMEL_N_FFT_CONSTS %d" % n_fft_consts)
    h.append("static const float logmel_fft_consts[%d] = {" % n_fft_consts)
    h.append(",\n".join("  %.9ef" % v for v in g.consts))
    h.append("};")
    # twiddle table [half][family][6][2], as doubles' nearest floats
    tw = np.zeros((2, 13, 6, 2), np.float64)
    for f in range(13):
        mine = [(kl, ku) for ff, kl, ku in slots if ff == f]
        for j, (kl, ku) in enumerate(mine):
            for hh, kap in ((0, kl), (1, ku if ku is not None else kl)):
                for p, (c, s) in enumerate(twiddles(kap)):
                    tw[hh, f, 3 * j + p] = (c, s)
    h.append("// window pair at position i of sub-DFT j's LDS table belongs to input m2 = logmel_win_pos[j][i]")
    h.append("static const unsigned char logmel_win_pos[4][25] = {")
    h.append(",\n".join("  {%s}" % ", ".join(str(v) for v in row) for row in win_pos))
    h.append("};")
    h.append("static const float logmel_twiddles[%d] = {" % tw.size)
    h.append(",\n".join("  %.9ef" % v for v in tw.reshape(-1)))
    h.append("};")
    for n_mels in (80, 128):
        pairs = plans[n_mels][2]
        allp = pairs[0] + pairs[1]
        h.append("#define LOGMEL_NNZ_%d %d" % (n_mels, len(allp)))
        h.append("// (bin, filter) of every weight slot, in table order")
        h.append("static const unsigned short logmel_nz_%d[%d][2] = {" % (n_mels, len(allp)))
        h.append(",\n".join("  {%d, %d}" % p for p in allp))
        h.append("};")
    with open(os.path.join(HERE, "logmel_tables.h"), "w") as f:
        f.write("\n".join(h) + "\n")
    print("fft ops: %d  consts: %d  table floats: %d  sa %d  swaps %d/%d (tmp %d/%d)  self-check err %.3g"
          % (sum(1 for n in live if g.nodes[n][0] != "in"), n_fft_consts, table_len, n_sa,
             len(plans[80][0]), len(plans[128][0]), plans[80][3], plans[128][3], err))


if __name__ == "__main__":
    main()
