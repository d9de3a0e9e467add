"""A SECOND, independent restatement of the reference's decoders, used only to pin oracle/.

Written directly from the reference's Rust sources -- /root/reference/src/decoder.rs,
src/decoder/flooding.rs, src/decoder/horizontal_layered.rs, src/decoder/arithmetic.rs,
src/decoder/factory.rs and src/sparse.rs (citations at each function) -- and NOT from oracle/: it
shares no code, no data structure and no author-time reading with the C oracle (numpy arrays
vectorised over frames here, per-frame AoS slots there), so a misreading would have to be made
twice, independently, to go unnoticed.  tests/test_independent_pin.py asserts bit equality of the
two on every reference implementation name.

Transcendentals: Rust's f32/f64 `tanh`, `ln`, `exp`, `ln_1p` are the platform libm's `tanhf`/`tanh`,
`logf`/`log`, `expf`/`exp`, `log1pf`/`log1p`; `atanh` is std's own formula
`0.5 * ((2.0 * x) / (1.0 - x)).ln_1p()`.  The libm calls are made through a 20-line C shim
(tests/libm_map.c: `for i: y[i] = tanhf(x[i])`, compiled with -fno-builtin) because numpy's own
tanh/log/exp are SIMD re-implementations that differ from libm in the last ulp.

Vectorisation: axis 0 is the frame.  All frames are stepped together; a frame's result is the
snapshot taken at its first zero syndrome (flooding.rs:69-79), so the extra steps the others need
never touch it.

Round 6: the four Minsum names (the headline rule, absent from the reference) are restated here too -- from SURVEY.md
Appendix A.6's definition as a literal fold, see class Minsum -- so that all 40 names have two independent readers.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBM = None


def _libm():
    global _LIBM
    if _LIBM is None:
        out = os.path.join(tempfile.gettempdir(), f"ldpc_libm_map_{os.getuid()}.so")
        src = os.path.join(_HERE, "libm_map.c")
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            subprocess.run(["gcc", "-O1", "-fno-builtin", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", out, "-lm"],
                           check=True)
        _LIBM = C.CDLL(out)
    return _LIBM


def _map(name, x):
    x = np.ascontiguousarray(x)
    y = np.empty_like(x)
    fn = getattr(_libm(), name + ("_f32" if x.dtype == np.float32 else "_f64"))
    fn(C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), C.c_size_t(x.size))
    return y


def m_tanh(x): return _map("map_tanh", x)
def m_ln(x): return _map("map_log", x)
def m_exp(x): return _map("map_exp", x)
def m_ln_1p(x): return _map("map_log1p", x)


def m_atanh(x):
    """Rust std f32/f64::atanh: 0.5 * ((2.0 * self) / (1.0 - self)).ln_1p()"""
    t = x.dtype.type
    return t(0.5) * m_ln_1p((t(2.0) * x) / (t(1.0) - x))


# ---- sparse.rs ------------------------------------------------------------------------------

def from_alist(text):
    """SparseMatrix::from_alist (sparse.rs:352-389): only the column lists are read, in order;
    insert(row, col) appends col to rows[row] and row to cols[col] unless present (sparse.rs:114-119).
    -> (rows, cols): iter_row(r) = rows[r], iter_col(c) = cols[c] (sparse.rs:240-248)"""
    lines = text.split("\n")
    ncols, nrows = (int(t) for t in lines[0].split()[:2])
    rows = [[] for _ in range(nrows)]
    cols = [[] for _ in range(ncols)]
    for c in range(ncols):
        for tok in lines[4 + c].split():
            r = int(tok)
            if r != 0 and (r - 1) not in cols[c]:
                rows[r - 1].append(c)
                cols[c].append(r - 1)
    return rows, cols


# ---- decoder.rs:157-174 ---------------------------------------------------------------------

def check_llrs(rows, hard):
    """true where every row has an even number of ones; hard: [B][n] bool"""
    ok = np.ones(hard.shape[0], dtype=bool)
    for r in rows:
        if r:
            ok &= (hard[:, r].sum(axis=1) % 2 == 0)
    return ok


# ---- arithmetic.rs: the float rules (one class per macro) -----------------------------------

class FloatArithmetic:
    """common part of impl_phif / impl_tanhf / impl_minstarapproxf / impl_aminstarf"""
    def __init__(self, f):
        self.f = f                                   # np.float32 / np.float64

    def input_llr_quantize(self, llr):               # `llr as $f`
        return llr.astype(self.f)

    def llr_hard_decision(self, llr):                # llr <= 0.0
        return llr <= 0

    def llr_to_var_message(self, llr): return llr
    def llr_to_var_llr(self, llr): return llr
    def var_llr_to_llr(self, v): return v
    def zero_message(self, shape): return np.zeros(shape, dtype=self.f)

    def send_var_messages(self, input_llr, msgs):
        """send_var_messages_no_clip (arithmetic.rs:140-156): llr = input + sum(m) with Rust's float
        Sum (a fold from -0.0 in std >= 1.83: library/core/src/iter/traits/accum.rs), messages llr - m"""
        s = np.full(input_llr.shape, -0.0, dtype=self.f)
        for j in range(msgs.shape[1]):
            s = s + msgs[:, j]
        llr = input_llr + s
        return llr, llr[:, None] - msgs


class Phi(FloatArithmetic):
    def phi(self, x):                                # arithmetic.rs:180-186
        x = np.maximum(x, self.f(1e-30))
        return -(m_ln(m_tanh(self.f(0.5) * x)))

    def send_check_messages(self, x):                # arithmetic.rs:214-246; x: [B][d] in slot order
        B, d = x.shape
        sign = np.zeros(B, dtype=bool)
        total = np.zeros(B, dtype=self.f)
        phis = np.empty_like(x)
        for j in range(d):
            p = self.phi(np.abs(x[:, j]))
            phis[:, j] = p
            total = total + p
            sign ^= x[:, j] < 0
        out = np.empty_like(x)
        for j in range(d):
            y = self.phi(total - phis[:, j])
            s = np.where(x[:, j] < 0, ~sign, sign)
            out[:, j] = np.where(s, -y, y)
        return out

    def update_check_messages_and_vars(self, r, q):  # arithmetic.rs:260-292; r: [B][d] Rcv, q: [B][d] Qv of the row
        x = q - r
        rcv = self.send_check_messages(x)            # same two loops on x = vars - msg
        return rcv, x + rcv                          # msg.value = rcv; vars = x + rcv


class Tanh(FloatArithmetic):
    def __init__(self, f):
        super().__init__(f)
        self.clampv = f(18.0) if f == np.float64 else f(9.0)   # arithmetic.rs:431-435

    def _tanhs(self, x):
        h = self.f(0.5) * x
        h = np.where(h < -self.clampv, -self.clampv, np.where(h > self.clampv, self.clampv, h))  # f32::clamp
        return m_tanh(h.astype(self.f))

    def _excl_products(self, t):
        B, d = t.shape
        out = np.empty_like(t)
        for i in range(d):
            p = np.full(B, 1.0, dtype=self.f)        # Product for floats folds from 1.0
            for j in range(d):
                if j != i:
                    p = p * t[:, j]
            out[:, i] = self.f(2.0) * m_atanh(p)
        return out

    def send_check_messages(self, x):                # arithmetic.rs:347-379
        return self._excl_products(self._tanhs(x))

    def update_check_messages_and_vars(self, r, q):  # arithmetic.rs:393-426
        rcv = self._excl_products(self._tanhs(q - r))
        return rcv, q + (rcv - r)                    # vars += rcv - old; msg = rcv


class Minstarapprox(FloatArithmetic):
    def _step(self, x, y):                           # (x.min(y) - (-(x - y).abs()).exp().ln_1p()).max(0.0)
        return np.maximum(np.minimum(x, y) - m_ln_1p(m_exp(-np.abs(x - y))), self.f(0.0))

    def _all(self, x):
        B, d = x.shape
        if d < 2:
            raise ValueError("only one variable message connected to check node")   # the reference's expect()
        out = np.empty_like(x)
        for i in range(d):
            sign = np.zeros(B, dtype=bool)
            acc = None
            for j in range(d):
                if j == i:
                    continue
                v = x[:, j]
                sign ^= v < 0
                v = np.abs(v)
                acc = v if acc is None else self._step(v, acc)
            out[:, i] = np.where(sign, -acc, acc)
        return out

    def send_check_messages(self, x):                # arithmetic.rs:487-521
        return self._all(x)

    def update_check_messages_and_vars(self, r, q):  # arithmetic.rs:535-574
        ms = self._all(q - r)
        return ms, q + (ms - r)


class Aminstar(FloatArithmetic):
    def _full(self, x, y):                           # x.min(y) - ln_1p(exp(-|x-y|)) + ln_1p(exp(-(x+y)))
        return (np.minimum(x, y) - m_ln_1p(m_exp(-np.abs(x - y)))) + m_ln_1p(m_exp(-(x + y)))

    def _core(self, x):
        B, d = x.shape
        if d < 1:
            raise ValueError("var_messages is empty")
        if d < 2:
            raise ValueError("var_messages_empty")
        argmin = np.abs(x).argmin(axis=1)            # Iterator::min_by keeps the FIRST of equal minima
        sign = np.zeros(B, dtype=bool)
        delta = np.zeros(B, dtype=self.f)
        have = np.zeros(B, dtype=bool)
        with np.errstate(all="ignore"):
            for j in range(d):
                v = x[:, j]
                sign ^= v < 0
                use = argmin != j
                a = np.abs(v)
                new = np.where(have, self._full(a, delta), a)
                delta = np.where(use, new, delta).astype(self.f)
                have |= use
            xmin = x[np.arange(B), argmin]
            first = np.where(sign ^ (xmin < 0), -delta, delta)
            vmin = np.abs(xmin)
            delta2 = self._full(delta, vmin)         # delta.min(vmin) - ... + ...
        out = np.empty_like(x)
        for j in range(d):
            other = np.where(sign ^ (x[:, j] < 0), -delta2, delta2)
            out[:, j] = np.where(argmin == j, first, other)
        return out

    def send_check_messages(self, x):                # arithmetic.rs:942-999
        return self._core(x)

    def update_check_messages_and_vars(self, r, q):  # arithmetic.rs:1013-1066
        x = q - r
        rcv = self._core(x)
        return rcv, x + rcv


class Minsum(FloatArithmetic):
    """NOT in the reference: the rule BASELINE.json's metric names, defined by SURVEY.md Appendix A.6 as
    "A.4 with `acc = min(a, acc)` (no correction, no clamp)" -- i.e. the Minstarapprox macro's loops
    (arithmetic.rs:487-521 flooding, 535-574 layered) with the fold step `(x.min(y) - ...).max(0.0)` reduced to
    `x.min(y)`.  Written from that sentence and from the macro's Rust text, not from oracle/rules.inc: the literal
    O(d^2) fold per excluded edge (no min1 / min2 / argmin shortcut), so that the closed form the oracle and the GPU
    kernels use is checked against the definition itself.  `f32::min` / `f64::min` are IEEE minNum (a NaN operand
    yields the other one): numpy's `fmin`, not `minimum`.

    `start`: the one place where the sentence leaves a choice.  "first" = the macro's `None => first magnitude`;
    "inf" = a fold from +inf, which is what the build states as ITS definition (oracle/rules.inc
    says so in its header comment).  The two are the same function of every NaN-free input -- asserted by
    test_minsum_fold_on_special_values -- and differ in exactly one corner: an excluded edge whose other inputs are ALL
    NaN (only reachable through inf - inf after infinite channel LLRs) gets NaN from "first" and +inf from "inf".
    test_minsum_nan_corner_is_the_documented_one pins that corner; decode() uses the build's definition."""
    def __init__(self, f, start="inf"):
        super().__init__(f)
        self.start = start

    def _all(self, x):
        B, d = x.shape
        if d < 2:
            raise ValueError("only one variable message connected to check node")   # the macro's expect()
        out = np.empty_like(x)
        for i in range(d):
            sign = np.zeros(B, dtype=bool)
            acc = np.full(B, np.inf, dtype=self.f) if self.start == "inf" else None
            for j in range(d):
                if j == i:
                    continue
                v = x[:, j]
                sign ^= v < 0                        # `x < 0.0`: false for -0.0 and for NaN
                a = np.abs(v)
                acc = a if acc is None else np.fmin(a, acc)
            out[:, i] = np.where(sign, -acc, acc)
        return out

    def send_check_messages(self, x):
        return self._all(x)

    def update_check_messages_and_vars(self, r, q):  # the tail of arithmetic.rs:535-574: vars += new - old; msg = new
        ms = self._all(q - r)
        return ms, q + (ms - r)


# ---- arithmetic.rs: the 8-bit rules ----------------------------------------------------------

def _clip(x):                                        # impl_8bitquant::clip (arithmetic.rs:609-617), i16 -> i8
    return np.where(x >= 127, 127, np.where(x <= -127, -127, x))


class I8Arithmetic:
    """impl_minstarapproxi8 / impl_aminstari8 with their three option closures.  Integers are held in
    int32 arrays (every intermediate of the reference fits i8 / i16, so no wrap can differ)."""
    QUANTIZER_C = 8.0

    def __init__(self, aminstar, jones, hardlimit, deg1clip):
        self.aminstar, self.jones, self.hardlimit, self.deg1clip = aminstar, jones, hardlimit, deg1clip
        table = []                                   # arithmetic.rs:588-598
        for t in range(128):
            v = self.QUANTIZER_C * float(_map("map_log1p", _map("map_exp", np.array([-(t / self.QUANTIZER_C)])))[0])
            xq = int(np.floor(abs(v) + 0.5) * (1 if v >= 0 else -1))     # f64::round: half away from zero
            if xq > 0:
                table.append(xq)
            else:
                break
        self.table = np.array(table + [0] * (300 - len(table)), dtype=np.int32)   # lookup(): 0 beyond the table

    def lookup(self, x):
        return self.table[np.minimum(x, 299)]

    def input_llr_quantize(self, llr):               # arithmetic.rs:678-687
        x = self.QUANTIZER_C * llr.astype(np.float64)
        r = np.sign(x) * np.floor(np.abs(x) + 0.5)   # round half away from zero
        return np.where(x >= 127.0, 127, np.where(x <= -127.0, -127, r)).astype(np.int32)

    def llr_hard_decision(self, llr): return llr <= 0
    def llr_to_var_message(self, llr): return llr
    def llr_to_var_llr(self, llr): return llr        # i16::from
    def var_llr_to_llr(self, v): return _clip(v)
    def zero_message(self, shape): return np.zeros(shape, dtype=np.int32)

    def _hl(self, x):                                # partial_hard_limit! (arithmetic.rs:803-815)
        if not self.hardlimit:
            return x
        return np.where(x <= -100, -127, np.where(x >= 100, 127, x))

    def send_var_messages(self, input_llr, msgs):    # impl_send_var_messages_i8 (arithmetic.rs:621-653)
        degree_one = msgs.shape[1] == 1
        inp = input_llr
        if self.deg1clip and degree_one:             # degree_one_clipping! (arithmetic.rs:817-833)
            inp = np.where(inp <= -116, -116, np.where(inp >= 116, 116, inp))
        llr = inp + msgs.sum(axis=1)
        if self.jones:
            llr = _clip(llr)
        return _clip(llr), _clip(llr[:, None] - msgs)

    def _minstar_all(self, x):                       # arithmetic.rs:707-737 / 752-776
        B, d = x.shape
        if d < 2:
            raise ValueError("only one variable message connected to check node")
        out = np.empty_like(x)
        for i in range(d):
            sign = np.zeros(B, dtype=bool)
            acc = None
            for j in range(d):
                if j == i:
                    continue
                v = x[:, j]
                sign ^= v < 0
                v = np.abs(v)
                acc = v if acc is None else np.maximum(np.minimum(v, acc) - self.lookup(np.abs(v - acc)), 0)
            out[:, i] = self._hl(np.where(sign, -acc, acc))
        return out

    def _aminstar_all(self, x):                      # arithmetic.rs:1134-1190 / 1201-1251
        B, d = x.shape
        if d < 2:
            raise ValueError("var_messages_empty")
        argmin = np.abs(x).argmin(axis=1)            # min_by_key: first minimum
        sign = np.zeros(B, dtype=bool)
        delta = np.zeros(B, dtype=np.int32)
        have = np.zeros(B, dtype=bool)

        def full(a, b):                              # saturating_add on i8 operands in [0, 127]
            return np.maximum(np.minimum(a, b) - self.lookup(np.abs(a - b)) + self.lookup(np.minimum(a + b, 127)), 0)
        for j in range(d):
            v = x[:, j]
            sign ^= v < 0
            use = argmin != j
            a = np.abs(v)
            new = np.where(have, full(a, delta), a)
            delta = np.where(use, new, delta)
            have |= use
        xmin = x[np.arange(B), argmin]
        dh = self._hl(delta)
        first = np.where(sign ^ (xmin < 0), -dh, dh)
        d2 = self._hl(full(delta, np.abs(xmin)))
        out = np.empty_like(x)
        for j in range(d):
            other = np.where(sign ^ (x[:, j] < 0), -d2, d2)
            out[:, j] = np.where(argmin == j, first, other)
        return out

    def send_check_messages(self, x):
        return self._aminstar_all(x) if self.aminstar else self._minstar_all(x)

    def update_check_messages_and_vars(self, r, q):
        if self.aminstar:                            # arithmetic.rs:1201-1251: vars = (vars - msg) + rcv, unclipped
            rcv = self._aminstar_all(_clip(q - r))
            return rcv, (q - r) + rcv
        ms = self._minstar_all(_clip(q - r))         # arithmetic.rs:778-781: vars += minstar - msg
        return ms, q + (ms - r)


# ---- factory.rs:240-277 ----------------------------------------------------------------------

def build(name):
    """-> (arithmetic, layered?) for one of the reference's 36 implementation names or the four Minsum names"""
    layered = name.startswith("HL")
    base = name[2:] if layered else name
    for rule, cls in (("Phi", Phi), ("Tanh", Tanh), ("Minstarapprox", Minstarapprox), ("Aminstar", Aminstar), ("Minsum", Minsum)):
        for suffix, f in (("f64", np.float64), ("f32", np.float32)):
            if base == rule + suffix:
                return cls(f), layered
    for rule in ("Minstarapproxi8", "Aminstari8"):
        if base.startswith(rule):
            opts = base[len(rule):]
            jones = opts.startswith("Jones")
            opts = opts[5:] if jones else opts
            hard = opts.startswith("PartialHardLimit")
            opts = opts[16:] if hard else opts
            deg1 = opts == "Deg1Clip"
            if opts not in ("", "Deg1Clip"):
                break
            if layered and (jones or deg1):
                break                                # the reference has only 4 layered i8 names
            return I8Arithmetic(rule == "Aminstari8", jones, hard, deg1), layered
    raise ValueError("invalid decoder implementation")


REFERENCE_NAMES = (
    [r + s for r in ("Phi", "Tanh", "Minstarapprox", "Aminstar") for s in ("f64", "f32")]
    + [b + j + h + d for b in ("Minstarapproxi8", "Aminstari8") for j in ("", "Jones")
       for h in ("", "PartialHardLimit") for d in ("", "Deg1Clip")]
    + ["HL" + r + s for r in ("Phi", "Tanh", "Minstarapprox", "Aminstar") for s in ("f64", "f32")]
    + ["HLMinstarapproxi8", "HLMinstarapproxi8PartialHardLimit", "HLAminstari8", "HLAminstari8PartialHardLimit"])
assert len(REFERENCE_NAMES) == 36
# the rule the reference lacks, named after the pattern of factory.rs:240-277
MINSUM_NAMES = ["Minsumf64", "Minsumf32", "HLMinsumf64", "HLMinsumf32"]
ALL_NAMES = REFERENCE_NAMES + MINSUM_NAMES


# ---- flooding.rs / horizontal_layered.rs -----------------------------------------------------

def decode(alist, name, llrs, max_iterations):
    """llrs [B][n] f64 -> (bits [B][n] u8, iterations [B] i32 with -1 = Err, final LLRs [B][n] f64)
    per frame exactly what `Decoder::decode` returns (flooding.rs:51-86, horizontal_layered.rs:49-88)"""
    A, layered = build(name)
    rows, cols = from_alist(alist)
    llrs = np.ascontiguousarray(llrs, dtype=np.float64)
    B, n = llrs.shape
    assert n == len(cols)
    raw_hard = llrs <= 0.0
    done = check_llrs(rows, raw_hard)                # "No bit errors case": iterations 0, hard decisions of the input
    iters = np.where(done, 0, -1).astype(np.int32)
    bits = raw_hard.astype(np.uint8)
    with np.errstate(all="ignore"):
        inp = A.input_llr_quantize(llrs)
        if layered:
            final, it_done, bits_run = _layered(A, rows, cols, inp, max_iterations, ~done)
        else:
            final, it_done, bits_run = _flooding(A, rows, cols, inp, max_iterations, ~done)
    run = ~done
    iters[run] = it_done[run]
    bits[run] = bits_run[run]
    out_llr = np.asarray(final, dtype=np.float64)
    out_llr[done] = llrs[done]                       # not defined by the reference (no decoder state): the input
    return bits, iters, out_llr


def _flooding(A, rows, cols, inp, max_iterations, run):
    B, n = inp.shape
    pos_in_row = [{v: j for j, v in enumerate(r)} for r in rows]
    pos_in_col = [{c: j for j, c in enumerate(col)} for col in cols]
    # Messages::from_iter: per destination, slots in iter_row / iter_col order, default values
    v2c = [A.zero_message((B, len(r))) for r in rows]
    c2v = [A.zero_message((B, len(col))) for col in cols]
    output = A.zero_message((B, n))                  # output_llrs: Default until the first variable pass
    for v, col in enumerate(cols):                   # initialize (flooding.rs:88-100)
        for c in col:
            v2c[c][:, pos_in_row[c][v]] = A.llr_to_var_message(inp[:, v])
    final = output.copy()
    iters = np.full(B, -1, dtype=np.int32)
    bits = A.llr_hard_decision(output).astype(np.uint8)    # max_iterations = 0: hard(Default)
    live = run.copy()
    for it in range(1, max_iterations + 1):
        for c, r in enumerate(rows):                 # process_check_nodes (flooding.rs:102-109)
            if not r:
                continue
            out = A.send_check_messages(v2c[c])
            for j, v in enumerate(r):
                c2v[v][:, pos_in_col[v][c]] = out[:, j]
        for v, col in enumerate(cols):               # process_variable_nodes (flooding.rs:111-125)
            llr, out = A.send_var_messages(inp[:, v], c2v[v])
            output[:, v] = llr
            for j, c in enumerate(col):
                v2c[c][:, pos_in_row[c][v]] = out[:, j]
        hard = A.llr_hard_decision(output)
        ok = check_llrs(rows, hard) & live
        final[ok] = output[ok]
        bits[ok] = hard[ok].astype(np.uint8)
        iters[ok] = it
        live &= ~ok
        if not live.any():
            break
    final[live] = output[live]                       # Err: hard decisions of the last output_llrs
    bits[live] = A.llr_hard_decision(output)[live].astype(np.uint8)
    return final, iters, bits


def _layered(A, rows, cols, inp, max_iterations, run):
    B, n = inp.shape
    q = A.llr_to_var_llr(inp).copy()                 # Qv (horizontal_layered.rs:90-103)
    rcv = [A.zero_message((B, len(r))) for r in rows]
    iters = np.full(B, -1, dtype=np.int32)
    final = np.asarray(A.var_llr_to_llr(q)).copy()
    bits = A.llr_hard_decision(A.var_llr_to_llr(q)).astype(np.uint8)   # max_iterations = 0: hard(input)
    live = run.copy()
    for it in range(1, max_iterations + 1):
        for c, r in enumerate(rows):                 # process_check_nodes: rows in order (horizontal_layered.rs:105-110)
            if not r:
                continue
            new_r, new_q = A.update_check_messages_and_vars(rcv[c], q[:, r])
            rcv[c] = new_r
            q[:, r] = new_q
        llr = A.var_llr_to_llr(q)
        hard = A.llr_hard_decision(llr)
        ok = check_llrs(rows, hard) & live
        final[ok] = llr[ok]
        bits[ok] = hard[ok].astype(np.uint8)
        iters[ok] = it
        live &= ~ok
        if not live.any():
            break
    llr = A.var_llr_to_llr(q)
    final[live] = llr[live]
    bits[live] = A.llr_hard_decision(llr)[live].astype(np.uint8)
    return final, iters, bits
