"""Lowering of PyREMOT ``reaction-rates`` lambdas to a HIP device function.

The reference evaluates the user's VARS / RATES dicts node by node in Python
(reactionRateExe, PyREMOT/docs/rmtReaction.py:11-61): VARS entries are evaluated in insertion
order into one namespace that already holds ``R_CONST, T, P, MoFri, SpCoi``; entries that are
functions are called with the partially filled namespace, everything else is copied as a
constant; then every RATES lambda is called with the full namespace.

Here the same ordered evaluation is performed ONCE, symbolically: each lambda is rebuilt with
its ``math`` / ``numpy`` globals (and closure cells) replaced by tracing shims and is called on
a namespace of symbols.  The result is an expression DAG over ``T, P, MoFri[i], SpCoi[i]`` with
hash-consed common subexpressions and folded constants, which is printed as straight-line HIP
C++ (one ``const real vN = ...;`` per node) for the fused RHS kernel.

Python raises on ``log(<=0)``, ``sqrt(<0)``, ``x/0``, ``exp`` overflow and ``pow`` domain
errors; the generated code sets bits of a per-reactor status word instead (see FLAG_*), which
the host turns back into the matching Python exception.

Constructs that cannot be traced (data-dependent ``if``, calls into C extensions other than
math/numpy elementwise functions, ...) raise ``LoweringError`` - there is no CPU fallback.
"""
import hashlib
import math
import os
import types

import numpy as np

FLAG_DOMAIN = 1       # math domain error   (log/sqrt/pow of a bad argument)
FLAG_DIV0 = 2         # float division by zero
FLAG_OVERFLOW = 4     # math range error    (exp/pow overflow)
FLAG_NONFINITE = 8    # a derivative came out NaN/Inf without one of the above
FLAG_STEP = 16        # adaptive step size underflow (rk45)

_EXP_MAX = 709.782712893384  # math.exp raises OverflowError above this


class LoweringError(Exception):
    pass


class Graph:
    """Hash-consed expression DAG.  Node = (op, a, b); ids are creation-ordered (topological)."""

    def __init__(self):
        self.nodes = []
        self._index = {}

    def _mk(self, op, a=None, b=None):
        key = (op, a, b)
        i = self._index.get(key)
        if i is None:
            i = len(self.nodes)
            self.nodes.append(key)
            self._index[key] = i
        return Sym(self, i)

    def const(self, v):
        v = float(v)
        return self._mk("const", v.hex() if math.isfinite(v) else repr(v))

    def inp(self, name):
        return self._mk("in", name)

    def is_const(self, i):
        return self.nodes[i][0] == "const"

    def cval(self, i):
        s = self.nodes[i][1]
        return float.fromhex(s) if s not in ("inf", "-inf", "nan") else float(s)


def _num(v):
    return isinstance(v, (int, float, np.floating, np.integer)) and not isinstance(v, bool)


class Sym:
    __slots__ = ("g", "i")
    __array_priority__ = 1000

    def __init__(self, g, i):
        self.g, self.i = g, i

    # ---- helpers
    def _lift(self, o):
        if isinstance(o, Sym):
            return o
        if _num(o) or isinstance(o, bool):
            return self.g.const(float(o))
        if isinstance(o, np.ndarray) and o.ndim == 0:
            return self.g.const(float(o))
        raise LoweringError("cannot combine a traced value with %r" % (type(o),))

    def _bin(self, op, a, b, pyop):
        g = self.g
        if g.is_const(a.i) and g.is_const(b.i):
            try:
                return g.const(pyop(g.cval(a.i), g.cval(b.i)))
            except (ZeroDivisionError, OverflowError, ValueError):
                pass  # keep it symbolic: the device flags it at run time like Python would raise
        if op in ("add", "mul") and a.i > b.i:      # canonical order for commutative ops
            a, b = b, a
        return g._mk(op, a.i, b.i)

    def _un(self, op, pyfn):
        g = self.g
        if g.is_const(self.i):
            try:
                return g.const(pyfn(g.cval(self.i)))
            except (ZeroDivisionError, OverflowError, ValueError):
                pass
        return g._mk(op, self.i)

    # ---- arithmetic
    def __add__(self, o): return self._bin("add", self, self._lift(o), lambda a, b: a + b)
    def __radd__(self, o): return self._bin("add", self._lift(o), self, lambda a, b: a + b)
    def __sub__(self, o): return self._bin("sub", self, self._lift(o), lambda a, b: a - b)
    def __rsub__(self, o): return self._bin("sub", self._lift(o), self, lambda a, b: a - b)
    def __mul__(self, o): return self._bin("mul", self, self._lift(o), lambda a, b: a*b)
    def __rmul__(self, o): return self._bin("mul", self._lift(o), self, lambda a, b: a*b)
    def __truediv__(self, o): return self._bin("div", self, self._lift(o), lambda a, b: a/b)
    def __rtruediv__(self, o): return self._bin("div", self._lift(o), self, lambda a, b: a/b)
    def __neg__(self): return self._un("neg", lambda a: -a)
    def __pos__(self): return self
    def __abs__(self): return self._un("abs", abs)
    def __pow__(self, o): return _pow(self, self._lift(o))
    def __rpow__(self, o): return _pow(self._lift(o), self)

    # ---- things a trace cannot follow
    def __bool__(self):
        raise LoweringError("data-dependent branching on a traced value is not supported")

    def _cmp(self, o):
        raise LoweringError("comparisons of traced values are not supported (use math-only "
                            "rate expressions, or max/min via numpy.maximum/minimum)")
    __lt__ = __le__ = __gt__ = __ge__ = _cmp

    def __float__(self):
        if self.g.is_const(self.i):
            return self.g.cval(self.i)
        raise LoweringError("a traced value was passed to a function that needs a real number "
                            "(a C function other than the math/numpy elementwise set?)")

    __hash__ = object.__hash__


def _pow(a, b):
    g = a.g
    if g.is_const(a.i) and g.is_const(b.i):
        try:
            return g.const(math.pow(g.cval(a.i), g.cval(b.i)))
        except (ZeroDivisionError, OverflowError, ValueError):
            pass
    if g.is_const(b.i):
        e = g.cval(b.i)
        if e == 1.0:
            return a
        if e == 0.0:
            return g.const(1.0)
        if e == 0.5:
            return a._un("sqrt", math.sqrt)
        if float(e).is_integer() and abs(e) <= 16:
            return g._mk("powi", a.i, int(e))
    if g.is_const(a.i):
        base = g.cval(a.i)
        if base == 10.0:
            return g._mk("exp10", b.i)
        if base == 2.0:
            return g._mk("exp2", b.i)
        if base == math.e:
            return g._mk("exp", b.i)
    return g._mk("pow", a.i, b.i)


class SymVector:
    """MoFri / SpCoi: integer-indexable, iterable vector of input symbols."""

    def __init__(self, syms):
        self._s = list(syms)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return SymVector(self._s[k])
        if isinstance(k, (int, np.integer)):
            return self._s[int(k)]
        raise LoweringError("only integer indexing of MoFri/SpCoi can be lowered")

    def __len__(self):
        return len(self._s)

    def __iter__(self):
        return iter(self._s)


def _fn1(op, pyfn):
    def f(x, *rest):
        if rest:
            raise LoweringError("%s() with extra arguments is not supported" % op)
        if isinstance(x, Sym):
            return x._un(op, pyfn)
        return pyfn(x)
    f.__name__ = op
    return f


class _ShimMath:
    """Replacement for the ``math`` module (and the elementwise subset of numpy) while tracing."""
    pi, e, inf, nan, tau = math.pi, math.e, math.inf, math.nan, math.tau
    exp = staticmethod(_fn1("exp", math.exp))
    log10 = staticmethod(_fn1("log10", math.log10))
    log2 = staticmethod(_fn1("log2", math.log2))
    sqrt = staticmethod(_fn1("sqrt", math.sqrt))
    fabs = staticmethod(_fn1("abs", math.fabs))
    abs = absolute = fabs
    sin = staticmethod(_fn1("sin", math.sin))
    cos = staticmethod(_fn1("cos", math.cos))
    tan = staticmethod(_fn1("tan", math.tan))
    tanh = staticmethod(_fn1("tanh", math.tanh))
    sinh = staticmethod(_fn1("sinh", math.sinh))
    cosh = staticmethod(_fn1("cosh", math.cosh))
    atan = arctan = staticmethod(_fn1("atan", math.atan))
    expm1 = staticmethod(_fn1("expm1", math.expm1))
    log1p = staticmethod(_fn1("log1p", math.log1p))

    @staticmethod
    def log(x, base=None):
        r = x._un("log", math.log) if isinstance(x, Sym) else math.log(x)
        if base is None:
            return r
        lb = base._un("log", math.log) if isinstance(base, Sym) else math.log(base)
        return r/lb

    @staticmethod
    def pow(x, y):
        if isinstance(x, Sym):
            return x**y
        if isinstance(y, Sym):
            return y.__rpow__(x)
        return math.pow(x, y)
    power = float_power = pow

    @staticmethod
    def maximum(a, b):
        return _minmax("max", a, b)

    @staticmethod
    def minimum(a, b):
        return _minmax("min", a, b)

    @staticmethod
    def sum(v):
        tot = 0.0
        for s in v:
            tot = tot + s
        return tot

    @staticmethod
    def array(v, *a, **k):
        return SymVector(list(v)) if any(isinstance(s, Sym) for s in v) else np.array(v, *a, **k)

    def __getattr__(self, name):
        raise LoweringError("math/numpy function %r cannot be lowered to the device" % name)


def _minmax(op, a, b):
    s = a if isinstance(a, Sym) else b
    if not isinstance(s, Sym):
        return max(a, b) if op == "max" else min(a, b)
    a, b = s._lift(a), s._lift(b)
    return s.g._mk(op, min(a.i, b.i), max(a.i, b.i))


_SHIM = _ShimMath()
_MATH_NAMES = {n for n in dir(math) if not n.startswith("_")}


def _shim_value(v, memo):
    """Map a global / closure value to its tracing counterpart."""
    if v is math or v is np:
        return _SHIM
    if isinstance(v, types.BuiltinFunctionType) and getattr(v, "__module__", None) == "math":
        return getattr(_SHIM, v.__name__)
    if isinstance(v, np.ufunc):
        return getattr(_SHIM, v.__name__)
    if isinstance(v, types.FunctionType):
        return rebind(v, memo)
    return v


def rebind(fn, memo=None):
    """Rebuild ``fn`` so that math / numpy references (globals and closure cells, recursively
    through helper functions) resolve to the tracing shims."""
    if memo is None:
        memo = {}
    if id(fn) in memo:
        return memo[id(fn)]
    mod = getattr(fn, "__module__", "") or ""
    if mod == __name__:
        return fn
    if mod == "math" or mod.split(".")[0] == "numpy":
        return getattr(_SHIM, fn.__name__)
    memo[id(fn)] = fn                  # provisional entry: guards against reference cycles
    names = set()
    stack = [fn.__code__]
    while stack:                       # names used by nested code objects too
        co = stack.pop()
        names.update(co.co_names)
        stack.extend(c for c in co.co_consts if isinstance(c, types.CodeType))
    g = {}
    for k in names:
        if k in fn.__globals__:
            g[k] = _shim_value(fn.__globals__[k], memo)
    g["__builtins__"] = fn.__globals__.get("__builtins__", __builtins__)
    cells = None
    if fn.__closure__:
        cells = tuple(types.CellType(_shim_value(c.cell_contents, memo)) for c in fn.__closure__)
    new = types.FunctionType(fn.__code__, g, fn.__name__, fn.__defaults__, cells)
    new.__kwdefaults__ = fn.__kwdefaults__
    for k, v in g.items():
        if v is fn:
            g[k] = new
    memo[id(fn)] = new
    return new


class Lowered:
    """Result of trace(): DAG + output node ids + op statistics."""

    def __init__(self, graph, outputs, nspecies):
        self.g, self.outputs, self.S = graph, outputs, nspecies
        self.live = self._live()

    def _live(self):
        live, stack = set(), list(self.outputs)
        while stack:
            i = stack.pop()
            if i in live:
                continue
            live.add(i)
            op, a, b = self.g.nodes[i]
            if op in ("const", "in"):
                continue
            stack.append(a)
            if b is not None and op != "powi":
                stack.append(b)
        return live

    def stats(self):
        st = {}
        for i in sorted(self.live):
            op = self.g.nodes[i][0]
            st[op] = st.get(op, 0) + 1
        return st

    def uses(self, name):
        return any(self.g.nodes[i][:2] == ("in", name) for i in self.live)

    def digest(self):
        h = hashlib.sha256()
        for i in sorted(self.live):
            h.update(repr(self.g.nodes[i]).encode())
        h.update(repr(self.outputs).encode())
        return h.hexdigest()

    # ---- evaluation on the host (used by tests to check the trace itself, not by the product path)
    def evaluate(self, T, P, x, C, U=()):
        env = {}
        for i in sorted(self.live):
            op, a, b = self.g.nodes[i]
            if op == "const":
                env[i] = self.g.cval(i)
            elif op == "in":
                env[i] = {"T": T, "P": P}.get(a) if a in ("T", "P") else (
                    x[int(a[1:])] if a[0] == "x" else (U[int(a[1:])] if a[0] == "u" else C[int(a[1:])]))
            elif op == "powi":
                env[i] = env[a]**b
            elif op in ("add", "sub", "mul", "div", "pow", "min", "max"):
                u, v = env[a], env[b]
                env[i] = {"add": lambda: u + v, "sub": lambda: u - v, "mul": lambda: u*v,
                          "div": lambda: u/v, "pow": lambda: math.pow(u, v),
                          "min": lambda: min(u, v), "max": lambda: max(u, v)}[op]()
            else:
                fn = {"neg": lambda u: -u, "abs": abs, "exp10": lambda u: 10.0**u,
                      "exp2": lambda u: 2.0**u, "rcp": lambda u: 1.0/u,
                      "expn": lambda u: math.exp(-u), "exp10n": lambda u: 10.0**(-u),
                      "exp2n": lambda u: 2.0**(-u), "sign": lambda u: (u > 0) - (u < 0),
                      "step": lambda u: 1.0 if u >= 0 else 0.0}.get(op) or getattr(math, op)
                env[i] = fn(env[a])
        return [env[o] for o in self.outputs]

    # ---- algebraic strength reduction (device build only; changes results by a few ulp)
    def optimize(self):
        """Cheaper but equivalent forms for the fp64 VALU (costs measured on gfx950):
          a/exp(z)           -> a*exp(-z)            when exp(z) is only ever a divisor
          a/b, c/b, ...      -> a*r, c*r, r = 1/b    when b divides at least twice
          log10(x)           -> log(x)*(1/ln 10)     (one log serves both bases)
        The Python-exception conditions of the original expressions are still flagged."""
        g = self.g
        uses = {}          # node -> list of (user, slot)
        for i in sorted(self.live):
            op, a, b = g.nodes[i]
            if op in ("const", "in"):
                continue
            uses.setdefault(a, []).append((i, 0))
            if b is not None and op != "powi":
                uses.setdefault(b, []).append((i, 1))
        for o in self.outputs:
            uses.setdefault(o, []).append((-1, 0))
        def only_divides(n):
            return bool(uses.get(n)) and all(u >= 0 and g.nodes[u][0] == "div" and slot == 1
                                             for (u, slot) in uses[n])
        # b "divides" directly (x/b) or through an integer power that itself only divides (x/b^n)
        div_uses = {}
        for n, us in uses.items():
            c = 0
            for (u, slot) in us:
                if u < 0:
                    continue
                uop = g.nodes[u][0]
                if uop == "div" and slot == 1:
                    c += 1
                elif uop == "powi" and g.nodes[u][2] > 0 and only_divides(u):
                    c += len(uses[u])
            div_uses[n] = c
        g2 = Graph()
        new = {}
        LOG10E = 1.0/math.log(10.0)
        NEGEXP = {"exp": "expn", "exp10": "exp10n", "exp2": "exp2n"}

        def mul(x, y):
            return x._bin("mul", x, y, lambda p, q: p*q)

        for i in sorted(self.live):
            op, a, b = g.nodes[i]
            if op == "const":
                new[i] = g2.const(g.cval(i))
            elif op == "in":
                new[i] = g2.inp(a)
            elif op == "powi":
                new[i] = None if (b > 0 and only_divides(i) and div_uses.get(a, 0) >= 2
                                  and not g.is_const(a)) else g2._mk("powi", new[a].i, b)
            elif op == "div":
                bop, ba, bb = g.nodes[b]
                if bop in NEGEXP and only_divides(b):
                    new[i] = mul(new[a], g2._mk(NEGEXP[bop], new[ba].i))
                elif bop == "powi" and (new[b] is None or (bb > 0 and div_uses.get(ba, 0) >= 2
                                                           and not g.is_const(ba))):
                    r = g2._mk("rcp", new[ba].i)          # 1/y exists anyway: x/y^n = x*(1/y)^n
                    new[i] = mul(new[a], g2._mk("powi", r.i, bb))
                elif div_uses.get(b, 0) >= 2 and not g.is_const(b):
                    new[i] = mul(new[a], g2._mk("rcp", new[b].i))
                else:
                    new[i] = g2._mk("div", new[a].i, new[b].i)
            elif op == "log10":
                lg = g2._mk("log", new[a].i)
                new[i] = mul(lg, g2.const(LOG10E))
            elif op in NEGEXP and only_divides(i):
                new[i] = None          # 1/exp(z) is emitted as exp(-z) at its uses
            elif b is None:
                new[i] = g2._mk(op, new[a].i)
            else:
                x, y = new[a].i, new[b].i
                if op in ("add", "mul", "min", "max") and x > y:
                    x, y = y, x
                new[i] = g2._mk(op, x, y)
        return Lowered(g2, [new[o].i for o in self.outputs], self.S).reassociate()

    def reassociate(self):
        """Products only: flatten every multiplication tree whose inner products have no other
        user, fold ALL its constant factors into one, drop factors 1.0, and rebuild it as
        ((K * inputs...) * others...) so that sub-products such as 1e-5*P are shared between the
        species (hash-consing).  Changes rounding at the 1e-16 level, never the exception
        behaviour (Python's float product does not raise)."""
        g = self.g
        nuse = {}
        for i in sorted(self.live):
            op, a, b = g.nodes[i]
            if op in ("const", "in"):
                continue
            nuse[a] = nuse.get(a, 0) + 1
            if b is not None and op != "powi":
                nuse[b] = nuse.get(b, 0) + 1
        for o in self.outputs:
            nuse[o] = nuse.get(o, 0) + 1

        def factors(i, root):
            op, a, b = g.nodes[i]
            if op == "mul" and (root or nuse.get(i, 0) == 1):
                return factors(a, False) + factors(b, False)
            return [i]

        inner = set()        # mul nodes absorbed into a parent product
        for i in sorted(self.live):
            op, a, b = g.nodes[i]
            if op == "mul":
                for c in (a, b):
                    if g.nodes[c][0] == "mul" and nuse.get(c, 0) == 1:
                        inner.add(c)
        g2 = Graph()
        new = {}
        for i in sorted(self.live):
            op, a, b = g.nodes[i]
            if op == "const":
                new[i] = g2.const(g.cval(i))
            elif op == "in":
                new[i] = g2.inp(a)
            elif op == "mul":
                if i in inner:
                    continue
                fs = factors(i, True)
                K, rest = 1.0, []
                for f in fs:
                    if g.is_const(f):
                        K *= g.cval(f)
                    else:
                        rest.append(f)
                if not math.isfinite(K) or K == 0.0 or not rest:
                    x, y = new[a].i if a in new else None, new[b].i if b in new else None
                    if x is None or y is None:        # operands were absorbed: rebuild plainly
                        acc = None
                        for f in fs:
                            acc = new[f] if acc is None else acc._bin("mul", acc, new[f], lambda p, q: p*q)
                        new[i] = acc
                    else:
                        new[i] = g2._mk("mul", min(x, y), max(x, y))
                    continue
                ins = sorted([f for f in rest if g.nodes[f][0] == "in"], key=lambda f: g.nodes[f][1])
                oth = [f for f in rest if g.nodes[f][0] != "in"]
                order = ins + oth
                acc = None if K == 1.0 else g2.const(K)
                for f in order:
                    acc = new[f] if acc is None else acc._bin("mul", acc, new[f], lambda p, q: p*q)
                new[i] = acc
            elif op == "powi":
                new[i] = g2._mk("powi", new[a].i, b)
            elif op == "rcp" and g2.nodes[new[a].i][0] == "mul":
                # 1/(K*T) -> (1/K)*(1/T): the kernel hands 1/T to the kinetics for free (it needs it
                # for M/T anyway), so this reciprocal becomes one multiplication
                mo, ma, mb = g2.nodes[new[a].i]
                kk, xx = (ma, mb) if g2.is_const(ma) else (mb, ma)
                if g2.is_const(kk) and g2.nodes[xx] == ("in", "T", None) and g2.cval(kk) != 0.0:
                    rt = g2._mk("rcp", xx)
                    new[i] = rt._bin("mul", g2.const(1.0/g2.cval(kk)), rt, lambda p, q: p*q)
                else:
                    new[i] = g2._mk(op, new[a].i)
            elif b is None:
                new[i] = g2._mk(op, new[a].i)
            else:
                x, y = new[a].i, new[b].i
                if op in ("add", "min", "max") and x > y:
                    x, y = y, x
                new[i] = g2._mk(op, x, y)
        return Lowered(g2, [new[o].i for o in self.outputs], self.S)

    # ---- symbolic gradient (sparse forward mode on the DAG)
    def gradient(self, wrt=None):
        """d outputs / d inputs as a DAG: every node carries the dict {input: derivative node} of its
        NON-ZERO partials (one-hot seeds, so an expression that does not depend on an input never gets
        a tangent for it), built with constant folding and the identities x*0, x*1, x+0 - the result
        is close to what one would write by hand.  ``wrt``: input names to differentiate by (default:
        T and every x_i / C_i the expressions use; P is a frozen parameter of the node Jacobian).
        Returns a ``Gradient`` (primal outputs + partials) for emission / evaluation."""
        g = self.g
        g2 = Graph()
        new = {}
        # primal part first: its node ids stay below n_primal, where the exception tests are kept
        for i in sorted(self.live):
            op, a, b = g.nodes[i]
            if op == "const":
                new[i] = g2.const(g.cval(i))
            elif op == "in":
                new[i] = g2.inp(a)
            elif op == "powi":
                new[i] = g2._mk("powi", new[a].i, b)
            elif b is None:
                new[i] = g2._mk(op, new[a].i)
            else:
                new[i] = g2._mk(op, new[a].i, new[b].i)
        n_primal = len(g2.nodes)
        if wrt is None:
            wrt = sorted({g.nodes[i][1] for i in self.live if g.nodes[i][0] == "in" and g.nodes[i][1] != "P"
                          and g.nodes[i][1][0] != "u"})        # u<k>: per-reactor parameters, frozen like P
        wrt = list(wrt)

        def c(v):
            return g2.const(v)

        def isc(x, v=None):
            return g2.is_const(x.i) and (v is None or g2.cval(x.i) == v)

        def add(x, y):
            if x is None:
                return y
            if y is None:
                return x
            if isc(x, 0.0):
                return y
            if isc(y, 0.0):
                return x
            return x._bin("add", x, y, lambda p, q: p + q)

        def sub(x, y):
            if y is None:
                return x
            if x is None:
                return neg(y)
            if isc(y, 0.0):
                return x
            return x._bin("sub", x, y, lambda p, q: p - q)

        def mul(x, y):
            if x is None or y is None:
                return None
            if isc(x, 0.0) or isc(y, 0.0):
                return None
            if isc(x, 1.0):
                return y
            if isc(y, 1.0):
                return x
            if isc(x, -1.0):
                return neg(y)
            if isc(y, -1.0):
                return neg(x)
            return x._bin("mul", x, y, lambda p, q: p*q)

        def neg(x):
            if x is None:
                return None
            if g2.nodes[x.i][0] == "neg":
                return Sym(g2, g2.nodes[x.i][1])
            return x._un("neg", lambda p: -p)

        def un(op, x):
            return g2._mk(op, x.i)

        LN10, LN2 = math.log(10.0), math.log(2.0)
        d = {}
        for i in sorted(self.live):
            op, a, b = g.nodes[i]
            u = new[i]
            if op == "const":
                d[i] = {}
                continue
            if op == "in":
                d[i] = {a: c(1.0)} if a in wrt else {}
                continue
            A = new[a]
            da = d[a]
            if op == "powi":
                n = b
                f = mul(c(float(n)), A if n == 2 else (c(1.0) if n == 1 else g2._mk("powi", A.i, n - 1)))
                d[i] = {k: mul(f, v) for k, v in da.items()}
                continue
            if b is not None:
                B = new[b]
                db = d[b]
                keys = list(dict.fromkeys(list(da) + list(db)))
                out = {}
                for k in keys:
                    x, y = da.get(k), db.get(k)
                    if op == "add":
                        v = add(x, y)
                    elif op == "sub":
                        v = sub(x, y)
                    elif op == "mul":
                        v = add(mul(x, B), mul(A, y))
                    elif op == "div":
                        v = mul(sub(x, mul(u, y)), un("rcp", B)) if (x is not None or y is not None) else None
                    elif op == "pow":
                        t1 = mul(y, un("log", A))
                        t2 = mul(mul(B, x), un("rcp", A))
                        v = mul(u, add(t1, t2))
                    elif op in ("max", "min"):
                        sel = un("step", sub(A, B) if op == "max" else sub(B, A))
                        v = add(mul(sel, x), mul(sub(c(1.0), sel), y))
                    else:
                        raise LoweringError("no derivative rule for op %r" % op)
                    if v is not None and not isc(v, 0.0):
                        out[k] = v
                d[i] = out
                continue
            # unary
            if op == "neg":
                f = c(-1.0)
            elif op == "abs":
                f = un("sign", A)
            elif op == "rcp":
                f = neg(mul(u, u))
            elif op == "sqrt":
                f = mul(c(0.5), un("rcp", u))
            elif op == "exp":
                f = u
            elif op == "exp10":
                f = mul(c(LN10), u)
            elif op == "exp2":
                f = mul(c(LN2), u)
            elif op == "expn":
                f = neg(u)
            elif op == "exp10n":
                f = mul(c(-LN10), u)
            elif op == "exp2n":
                f = mul(c(-LN2), u)
            elif op == "log":
                f = un("rcp", A)
            elif op == "log10":
                f = mul(c(1.0/LN10), un("rcp", A))
            elif op == "log2":
                f = mul(c(1.0/LN2), un("rcp", A))
            elif op == "log1p":
                f = un("rcp", add(c(1.0), A))
            elif op == "expm1":
                f = add(u, c(1.0))
            elif op == "sin":
                f = un("cos", A)
            elif op == "cos":
                f = neg(un("sin", A))
            elif op == "tan":
                f = add(c(1.0), mul(u, u))
            elif op == "tanh":
                f = sub(c(1.0), mul(u, u))
            elif op == "sinh":
                f = un("cosh", A)
            elif op == "cosh":
                f = un("sinh", A)
            elif op == "atan":
                f = un("rcp", add(c(1.0), mul(A, A)))
            elif op in ("sign", "step"):
                f = None
            else:
                raise LoweringError("no derivative rule for op %r" % op)
            out = {}
            for k, v in da.items():
                w = mul(f, v)
                if w is not None and not isc(w, 0.0):
                    out[k] = w
            d[i] = out
        partial = [{k: v.i for k, v in d[o].items()} for o in self.outputs]
        return Gradient(g2, [new[o].i for o in self.outputs], partial, self.S, n_primal, wrt)

    # ---- HIP C++ emission
    def _emitter(self, nocheck_from=None):
        """-> emit(i, name, declare=True, nocheck=False): lines of HIP C++ for live node i (appending its C expression
        to ``name``).  ``declare`` False prints an assignment to an already declared ``v<i>`` instead of a
        ``const real`` definition (nodes evaluated inside a branch and used after it)."""
        g = self.g
        seen_checks = set()
        table = getattr(self, "_const_table", None)

        def lit(v):
            if math.isnan(v):
                return "real(__builtin_nan(\"\"))"
            if math.isinf(v):
                return "real(__builtin_inf())" if v > 0 else "real(-__builtin_inf())"
            v = float(v)
            # experiment (emit(const_table=...)): constants that are not hardware inline constants come from a
            # __constant__ table (scalar loads) instead of s_mov literal pairs
            if table is not None and v not in (0.0, 0.5, -0.5, 1.0, -1.0, 2.0, -2.0, 4.0, -4.0):
                if v not in table:
                    table[v] = len(table)
                return "real(RMT_KTAB[%d])" % table[v]
            return "real(%s)" % repr(v)

        def emit(i, name, declare=True, nocheck=False):
            lines = []
            op, a, b = g.nodes[i]
            if op == "const":
                name[i] = lit(g.cval(i))
                return lines
            if op == "in":
                name[i] = a if a in ("T", "P") else ("x[%s]" % a[1:] if a[0] == "x" else (
                    "U[%s]" % a[1:] if a[0] == "u" else "C[%s]" % a[1:]))
                return lines
            v = "v%d" % i
            A = name[a]
            B = name[b] if (b is not None and op != "powi") else None
            pre = []
            if op == "add":
                e = "%s + %s" % (A, B)
            elif op == "sub":
                e = "%s - %s" % (A, B)
            elif op == "mul":
                e = "%s * %s" % (A, B)
            elif op == "div":
                pre.append("RMT_CHECK_DEN(flag, %s);" % B)
                e = "rmt_div(%s, %s)" % (A, B)
            elif op == "rcp":
                pre.append("RMT_CHECK_DEN(flag, %s);" % A)
                e = "invT" if A == "T" else "rmt_rcp(%s)" % A     # 1/T comes with the node state
            elif op == "expn":      # stands for 1/exp(A): Python raises if exp(A) overflows or is 0
                pre.append("RMT_CHECK_EXP(flag, rmt_abs(%s));" % A)
                e = "rmt_exp(-%s)" % A
            elif op in ("exp10n", "exp2n"):
                k = {"exp10n": math.log(10.0), "exp2n": math.log(2.0)}[op]
                pre.append("RMT_CHECK_EXP(flag, rmt_abs(%s) * real(%r));" % (A, k))
                e = "rmt_%s(-%s)" % (op[:-1], A)
            elif op == "neg":
                e = "-%s" % A
            elif op == "abs":
                e = "rmt_abs(%s)" % A
            elif op == "powi":
                n = abs(b)
                terms, sq, cur = [], n, A
                # binary powering with named squares
                k = 0
                sqname = A
                while sq:
                    if sq & 1:
                        terms.append(sqname)
                    sq >>= 1
                    if sq:
                        k += 1
                        nm = "%s_s%d" % (v, k)
                        pre.append("const real %s = %s * %s;" % (nm, sqname, sqname))
                        sqname = nm
                prod = " * ".join(terms)
                if b < 0:
                    pre.append("RMT_CHECK_DEN(flag, %s);" % A)
                    e = "rmt_rcp(%s)" % prod
                else:
                    e = prod
            elif op == "pow":
                pre.append("RMT_CHECK(flag, %s < real(0) && %s != trunc(%s), %du);" % (A, B, B, FLAG_DOMAIN))
                pre.append("RMT_CHECK(flag, %s == real(0) && %s < real(0), %du);" % (A, B, FLAG_DIV0))
                e = "rmt_pow(%s, %s)" % (A, B)
            elif op in ("log", "log10", "log2", "log1p"):
                pre.append("RMT_CHECK_POS(flag, %s);" % (A if op != "log1p" else "(%s + real(1))" % A))
                e = "rmt_%s(%s)" % (op, A)
            elif op == "sqrt":
                pre.append("RMT_CHECK_NONNEG(flag, %s);" % A)
                e = "rmt_sqrt(%s)" % A
            elif op in ("exp", "exp10", "exp2", "expm1", "sinh", "cosh"):
                lim = {"exp": _EXP_MAX, "expm1": _EXP_MAX, "sinh": 710.4758600739439,
                       "cosh": 710.4758600739439, "exp10": 308.2547155599167, "exp2": 1024.0}[op]
                # one running maximum against exp's limit 709.78: scale the other bases' arguments
                scale = {"exp10": math.log(10.0), "exp2": math.log(2.0)}.get(op)
                arg = A if op not in ("sinh", "cosh") else "rmt_abs(%s)" % A
                pre.append("RMT_CHECK_EXP(flag, %s);" % (arg if scale is None else "%s * real(%r)" % (arg, scale)))
                e = "rmt_%s(%s)" % (op, A)
            elif op in ("sin", "cos", "tan", "tanh", "atan"):
                e = "rmt_%s(%s)" % (op, A)
            elif op in ("min", "max"):
                e = "rmt_%s(%s, %s)" % (op, A, B)
            elif op == "sign":
                e = "(%s > real(0) ? real(1) : (%s < real(0) ? real(-1) : real(0)))" % (A, A)
            elif op == "step":
                e = "(%s >= real(0) ? real(1) : real(0))" % A
            else:
                raise LoweringError("no device emission for op %r" % op)
            for p in pre:
                if p.startswith("RMT_CHECK"):
                    if nocheck or p in seen_checks or (nocheck_from is not None and i >= nocheck_from):
                        continue
                    if declare:               # (a check printed inside a branch does not cover the code after it)
                        seen_checks.add(p)
                lines.append("    " + p)
            lines.append(("    const real %s = %s;" if declare else "    %s = %s;") % (v, e))
            name[i] = v
            return lines
        return emit

    def _emit_body(self, nocheck_from=None):
        """Straight-line code for every live node -> (lines, {node: C expression}).  Nodes with an id
        >= nocheck_from (the derivative part of a gradient DAG) are printed without the
        Python-exception tests: they are not expressions the reference evaluates."""
        emit = self._emitter(nocheck_from)
        lines, name = [], {}
        for i in sorted(self.live):
            lines.extend(emit(i, name))
        return lines, name

    # ---- cache of the temperature-only transcendentals ("K-cache")
    KC_THR = 2.0**-9      # |d| <= 2^-9: the degree-4 Taylor sum of e^d is exact to 2.4e-16, log1p to degree 5 to 1e-17
    _EXP_ROOTS = {"exp": (1.0, 1.0), "expn": (-1.0, 1.0), "exp10": (1.0, math.log(10.0)),
                  "exp10n": (-1.0, math.log(10.0)), "exp2": (1.0, math.log(2.0)), "exp2n": (-1.0, math.log(2.0))}

    def kcache_plan(self, gen=True):
        """Which nodes of the DAG are worth caching per mesh node between right-hand-side evaluations.

        Rate and equilibrium constants depend on the temperature only - K = exp(f(T)), Arrhenius f = -E/(R T) - and
        they are the expensive part of a kinetics evaluation (9 of the 13 transcendentals of the DME mechanism, 29 % of
        the bench kernel's time), while T at a mesh node moves by millikelvins between RK stages and time steps.  With
        K_ref = K(T_ref) kept per node,  K(T) = K_ref e^d,  d = f(T) - f(T_ref)  is a 4-term Taylor sum as long as
        |d| <= 2^-9 (5 fp64 operations instead of 10 + a dependent table look-up); log(T) likewise through log1p.  A
        wave whose lanes all pass that test takes the short path, otherwise the wave evaluates in full and moves its
        reference point (cache refresh) - so the result never depends on the cache beyond the 2.4e-16 of the Taylor
        remainder and the rounding of d (|f| 2e-16 absolute, i.e. ~1e-14 relative in K).

        Returns None when there is nothing to cache, else a dict: ``roots`` (node ids, creation order), ``kind`` per
        root ("log", ("lin", coef) = exponent linear in 1/T with that coefficient, or "gen" = exponent stored),
        ``slot`` / ``fslot`` (cache slot of the value / of a "gen" exponent), ``slots`` (doubles per mesh node),
        ``branch`` (nodes evaluated inside the cached section, roots included), ``prologue`` (their root-free
        T-only ancestors, evaluated before it).  ``gen=False`` leaves the "gen" constants (two slots each) out of the
        cache: they are evaluated in full on either path (their log(T) still comes from the cache).

        ``gen="basis"``: an exponent that is a LINEAR COMBINATION of T^n (|n| <= 4) and log T - the usual form of an
        equilibrium constant, ln K = a/T + b ln T + c T + d T^2 + ... - needs no stored exponent: f(T) - f(T_ref) is the
        same combination of the differences of the basis functions, which follow from T - T_ref, 1/T - 1/T_ref and
        log1p(T/T_ref - 1) in a handful of operations shared by all such constants.  kind = ("basis", {key: coef}) with
        keys ("p", n) / ("log",), one slot each; ``tslot`` = slot of T_ref (when a positive power occurs).  Exponents that
        do not decompose stay out of the cache.  ``outside_exp``: True when some exp-family node of the DAG is NOT a
        root (its evaluation needs the exp table whatever the path)."""
        g = self.g
        live = sorted(self.live)

        def operands(i):
            op, a, b = g.nodes[i]
            if op in ("const", "in"):
                return []
            return [a] + ([b] if (b is not None and op != "powi") else [])
        tonly = {}
        for i in live:
            op, a, b = g.nodes[i]
            tonly[i] = True if op == "const" else ((a == "T") if op == "in" else all(tonly[o] for o in operands(i)))
        def lin_in_invT(i):
            """coefficient c if node i is exactly c/T through products with constants and negations, else None."""
            op, a, b = g.nodes[i]
            if op == "rcp" and g.nodes[a] == ("in", "T", None):
                return 1.0
            if op == "neg":
                c = lin_in_invT(a)
                return None if c is None else -c
            if op == "mul":
                for k, x in ((a, b), (b, a)):
                    if g.is_const(k):
                        c = lin_in_invT(x)
                        if c is not None:
                            return g.cval(k)*c
            if op == "div" and g.is_const(a) and g.nodes[b] == ("in", "T", None):
                return g.cval(a)
            return None
        def basis(i, memo={}):
            """{key: coef} if node i is a linear combination of T^n and log T (constant term under ("c",)), else None."""
            if i in memo:
                return memo[i]
            op, a, b = g.nodes[i]
            out = None
            mono = lambda d: d is not None and len(d) == 1 and next(iter(d))[0] in ("p", "c")
            def scaled(d, c):
                return None if d is None else {k: v*c for k, v in d.items()}
            def times(x, y):                      # monomial x monomial
                (kx, cx), (ky, cy) = next(iter(x.items())), next(iter(y.items()))
                n = (kx[1] if kx[0] == "p" else 0) + (ky[1] if ky[0] == "p" else 0)
                return {(("p", n) if n else ("c",)): cx*cy}
            if op == "const":
                out = {("c",): g.cval(i)}
            elif op == "in":
                out = {("p", 1): 1.0} if a == "T" else None
            elif op == "log":
                out = {("log",): 1.0} if g.nodes[a] == ("in", "T", None) else None
            elif op == "neg":
                out = scaled(basis(a), -1.0)
            elif op in ("add", "sub"):
                x, y = basis(a), basis(b)
                if x is not None and y is not None:
                    out = dict(x)
                    for k, v in y.items():
                        out[k] = out.get(k, 0.0) + (v if op == "add" else -v)
            elif op == "mul":
                x, y = basis(a), basis(b)
                if x is not None and y is not None:
                    if set(x) == {("c",)}:
                        out = scaled(y, x[("c",)])
                    elif set(y) == {("c",)}:
                        out = scaled(x, y[("c",)])
                    elif mono(x) and mono(y):
                        out = times(x, y)
                    elif mono(x) and all(k[0] in ("p", "c") for k in y):
                        out = {}
                        for k, v in y.items():
                            out.update(times(x, {k: v}))
                    elif mono(y) and all(k[0] in ("p", "c") for k in x):
                        out = {}
                        for k, v in x.items():
                            out.update(times(y, {k: v}))
            elif op in ("rcp", "div"):
                den = basis(a if op == "rcp" else b)
                num = {("c",): 1.0} if op == "rcp" else basis(a)
                if mono(den) and num is not None and next(iter(den.values())) != 0.0 \
                        and all(k[0] in ("p", "c") for k in num):
                    (kd, cd), = den.items()
                    inv = {(("p", -kd[1]) if kd[0] == "p" else ("c",)): 1.0/cd}
                    out = {}
                    for k, v in num.items():
                        out.update(times(inv, {k: v}))
            elif op == "powi":
                x = basis(a)
                if mono(x):
                    (kx, cx), = x.items()
                    n = (kx[1] if kx[0] == "p" else 0)*b
                    out = {(("p", n) if n else ("c",)): cx**b}
            if out is not None and not all(math.isfinite(v) for v in out.values()):
                out = None
            memo[i] = out
            return out
        roots = []
        bases, lins = {}, {}
        for i in live:
            op, a, b = g.nodes[i]
            if op in self._EXP_ROOTS and tonly[a] and not g.is_const(a):
                d = basis(a, {})                  # (an additive constant in the exponent drops out of every difference)
                d = None if d is None else {k: v for k, v in d.items() if k != ("c",) and v != 0.0}
                c = lin_in_invT(a)
                if c in (None, 0.0) and d and set(d) == {("p", -1)}:
                    c = d[("p", -1)]              # c/T + const: Arrhenius written relative to a reference temperature
                if c not in (None, 0.0) and math.isfinite(c):
                    lins[i] = c
                    roots.append(i)
                elif gen is True:
                    roots.append(i)
                elif gen == "basis" and d and all(k == ("log",) or 1 <= abs(k[1]) <= 4 for k in d):
                    bases[i] = d
                    roots.append(i)
            elif op == "log" and g.nodes[a] == ("in", "T", None):
                roots.append(i)
        if not roots:
            return None
        rootset = set(roots)
        dep = {}                      # depends on a root (strictly below it)
        for i in live:
            dep[i] = any(dep[o] or o in rootset for o in operands(i))
        need, stack = set(), [g.nodes[r][1] for r in roots]
        while stack:                  # ancestors of the roots' arguments
            i = stack.pop()
            if i in need:
                continue
            need.add(i)
            stack.extend(operands(i))

        kind, slot, fslot, nslots = {}, {}, {}, 1          # slot 0: 1/T_ref
        for r in roots:
            op, a, b = g.nodes[r]
            slot[r] = nslots
            nslots += 1
            if op == "log":
                kind[r] = "log"
                continue
            if r in lins:
                kind[r] = ("lin", lins[r])
            elif r in bases:
                kind[r] = ("basis", bases[r])
            else:
                kind[r] = "gen"
                fslot[r] = nslots
                nslots += 1
        tslot = None
        if any(k[0] == "p" and k[1] > 0 for d in bases.values() for k in d):
            tslot = nslots
            nslots += 1
        branch = [i for i in live if i in rootset or (i in need and dep[i] and not g.is_const(i))]
        prologue = [i for i in live if i in need and not dep[i] and i not in rootset]
        outside = any(g.nodes[i][0] in self._EXP_ROOTS and i not in rootset for i in live)
        return {"roots": roots, "kind": kind, "slot": slot, "fslot": fslot, "slots": nslots, "branch": branch,
                "prologue": prologue, "tslot": tslot, "outside_exp": outside}

    def emit(self, fname="rmt_kinetics", const_table=False, kcache=False, kcache_gen=True, kcache_thr=None):
        """The device function of the rates.  ``kcache``: with the cached section of kcache_plan() for callers that
        pass a cache (template parameter KC with KC::enabled; every other caller passes rmt_nocache_t and gets the
        plain evaluation - the section is discarded at compile time)."""
        self._const_table = {} if const_table else None
        plan = self.kcache_plan(kcache_gen) if kcache else None
        self._kc_thr = self.KC_THR
        if kcache_thr is not None:        # (tests: a smaller range makes the callers' out-of-range handling run)
            if not 0.0 < float(kcache_thr) <= self.KC_THR:
                raise ValueError("the cache's Taylor range must lie in (0, 2^-9]")
            self._kc_thr = float(kcache_thr)
        if plan is None:
            lines, name = self._emit_body()
        else:
            lines, name = self._emit_cached(plan)
        table, self._const_table = self._const_table, None
        body = "\n".join(lines)
        if table:
            vals = sorted(table, key=table.get)
            body = "    // literals through the scalar cache\n" + body
            self._table_decl = ("__constant__ double RMT_KTAB[%d] = {%s};\n"
                                % (len(vals), ", ".join(repr(v) for v in vals)))
        else:
            self._table_decl = ""
        outs = "\n".join("    r[%d] = %s;" % (k, name[o]) for k, o in enumerate(self.outputs))
        head = "#define RMT_KC_SLOTS %d\n" % (plan["slots"] if plan else 0)
        return head + self._table_decl + (
            "template <typename FL, typename KC, int MODE = 0>\n"
            "__device__ __forceinline__ void %s(const real T, const real invT, const real P,\n"
            "        const real* __restrict__ x, const real* __restrict__ C, const real* __restrict__ U,\n"
            "        real* __restrict__ r, FL& flag, KC& kc) {\n"
            "    (void)invT; (void)U; (void)kc;\n%s\n%s\n}\n"
            % (fname, body, outs))

    def _emit_cached(self, plan):
        """Body with the cached section: MODE 2 takes every cached constant from its reference value by a Taylor step
        and tests the range that step serves - |d| <= KC_THR in every exponent, |T/T_ref - 1| <= KC_THR for log(T) - a
        lane out of range only marks its cache (`kc.leave`: there is no second code path, the CALLER discards what it
        computed from such an evaluation, see csrc/kernels/50_rk4.inc rmt_rk4_reg_body); MODE 0 evaluates in full and
        (with a cache) stores the new reference point.  Full evaluations made on behalf of a cache go through
        rmt_exp_sel<KC::enabled> (00_config_math.inc): a caching kernel may keep a smaller exp table than the rest."""
        g = self.g
        emit = self._emitter()
        lines, name = [], {}
        for i in sorted(self.live):                       # constants and inputs have no code: name them first
            if g.nodes[i][0] in ("const", "in"):
                emit(i, name)
        done = set(name)
        for i in plan["prologue"]:
            if i not in done:
                lines.extend(emit(i, name))
                done.add(i)
        for i in plan["branch"]:                          # values that leave the cached section
            lines.append("    real v%d;" % i)
        thr = repr(self._kc_thr)
        taylor = ("k_ + (k_ * d_) * (real(1) + d_ * (real(0.5) + d_ * (real(%r) + d_ * real(%r))))"
                  % (1.0/6.0, 1.0/24.0))
        kinds = plan["kind"]
        lin = [abs(self._EXP_ROOTS[g.nodes[r][0]][0]*self._EXP_ROOTS[g.nodes[r][0]][1]*kinds[r][1])
               for r in plan["roots"] if isinstance(kinds[r], tuple) and kinds[r][0] == "lin"]
        bases = {r: kinds[r][1] for r in plan["roots"] if isinstance(kinds[r], tuple) and kinds[r][0] == "basis"}
        powers = sorted({k[1] for d in bases.values() for k in d if k[0] == "p"})
        want_log = any(k == "log" for k in kinds.values()) or any(("log",) in d for d in bases.values())
        L = lines.append
        L("    if constexpr (KC::enabled && MODE == 2) {      // every constant from its cached value: K = K_ref e^d, d = f(T) - f(T_ref)")
        L("        const real kc_it = kc.get(0);")
        L("        const real kc_di = invT - kc_it;")
        L("        (void)kc_di;")
        tests = []                                        # the range test sits next to the values it shares with the Taylor steps
        if lin:
            tests.append("!(rmt_abs(kc_di) <= real(%r))" % (self._kc_thr/max(lin)))
        if want_log:
            L("        const real kc_s = T * kc_it - real(1);")
            L("        const real kc_dl = kc_s * (real(1) + kc_s * (real(-0.5) + kc_s * (real(%r) + kc_s * (real(-0.25) + kc_s * real(0.2)))));      // log(T) - log(T_ref) = log1p(T/T_ref - 1)"
              % (1.0/3.0))
            tests.append("!(rmt_abs(kc_s) <= real(%r))" % self._kc_thr)
        if tests:
            L("        kc.leave(%s);" % " || ".join(tests))
        # differences of the basis functions of the ("basis", ...) exponents: T^n - T_ref^n from (T - T_ref) or (1/T - 1/T_ref)
        dname = {("log",): "kc_dl"}
        if any(n > 0 for n in powers):
            L("        const real kc_tr = kc.get(%d);" % plan["tslot"])
            L("        const real kc_dt = T - kc_tr;")
        for n in powers:
            x, xr, dx = ("T", "kc_tr", "kc_dt") if n > 0 else ("invT", "kc_it", "kc_di")
            m = abs(n)
            nm = "kc_d%s%d" % ("p" if n > 0 else "m", m)
            dname[("p", n)] = nm
            if m == 1:
                L("        const real %s = %s;" % (nm, dx))
            elif m == 2:
                L("        const real %s = %s * (%s + %s);" % (nm, dx, x, xr))
            elif m == 3:
                L("        const real %s = %s * (%s * (%s + %s) + %s * %s);" % (nm, dx, x, x, xr, xr, xr))
            else:
                L("        const real %s = (%s * (%s + %s)) * (%s * %s + %s * %s);" % (nm, dx, x, xr, x, x, xr, xr))
        # ... consumed at once: only the exponents' changes stay live (one value per constant), with ONE range test for all
        # of them (a NaN exponent implies a NaN temperature, which the tests above catch)
        dmax = []
        for i, d in bases.items():
            sg, lnb = self._EXP_ROOTS[g.nodes[i][0]]
            terms = " + ".join("real(%r) * %s" % (sg*lnb*c, dname[key]) for key, c in d.items())
            L("        const real kc_e%d = %s;" % (i, terms))
            dmax.append("rmt_abs(kc_e%d)" % i)
        if dmax:
            big = dmax[0]
            for t in dmax[1:]:
                big = "rmt_max(%s, %s)" % (big, t)
            L("        kc.leave(!(%s <= real(%s)));" % (big, thr))
        for i in plan["branch"]:
            op, a, b = g.nodes[i]
            v = "v%d" % i
            if i not in kinds:
                for ln in emit(i, name, declare=False, nocheck=True):
                    L("    " + ln)
                continue
            k = kinds[i]
            if k == "log":
                L("        %s = kc.get(%d) + kc_dl;" % (v, plan["slot"][i]))
            elif k[0] == "lin":
                sg, lnb = self._EXP_ROOTS[op]
                L("        {")
                L("            const real d_ = real(%r) * kc_di;" % (sg*lnb*k[1]))
                L("            const real k_ = kc.get(%d);" % plan["slot"][i])
                L("            %s = %s;" % (v, taylor))
                L("        }")
            elif k[0] == "basis":
                L("        {")
                L("            const real d_ = kc_e%d;" % i)
                L("            const real k_ = kc.get(%d);" % plan["slot"][i])
                L("            %s = %s;" % (v, taylor))
                L("        }")
            else:
                sg, lnb = self._EXP_ROOTS[op]
                L("        {")
                L("            const real d_ = real(%r) * (%s - kc.get(%d));" % (sg*lnb, name[a], plan["fslot"][i]))
                L("            kc.leave(!(rmt_abs(d_) <= real(%s)));" % thr)
                L("            const real k_ = kc.get(%d);" % plan["slot"][i])
                L("            %s = %s;" % (v, taylor))
                L("        }")
            name[i] = v
        L("    } else {                 // full evaluation; with a cache its reference point moves here (every constant is")
        L("        if constexpr (KC::enabled) kc.put(0, invT);      // stored as soon as it exists: short live ranges)")
        if plan.get("tslot") is not None:
            L("        if constexpr (KC::enabled) kc.put(%d, T);" % plan["tslot"])
        for i in plan["branch"]:
            for ln in emit(i, name, declare=False):
                if i in kinds and kinds[i] != "log":          # the table-driven exp of whoever owns the table
                    for fn in ("rmt_exp10(", "rmt_exp2(", "rmt_exp("):
                        ln = ln.replace(fn, fn[:-1] + "_sel<KC::enabled>(")
                L("    " + ln)
            done.add(i)
            if i in plan["slot"]:
                L("        if constexpr (KC::enabled) {")
                L("            kc.put(%d, v%d);" % (plan["slot"][i], i))
                if i in plan["fslot"]:
                    L("            kc.put(%d, %s);" % (plan["fslot"][i], name[g.nodes[i][1]]))
                L("        }")
        L("    }")
        for i in sorted(self.live):
            if i not in done:
                lines.extend(emit(i, name))
                done.add(i)
        return lines, name


class Gradient(Lowered):
    """Rates and their non-zero partial derivatives with respect to the node inputs (T, x_i, C_i)."""

    def __init__(self, graph, outputs, partial, nspecies, n_primal, wrt):
        self.partial, self.n_primal, self.wrt = partial, n_primal, wrt
        self.n_rates = len(outputs)
        flat = list(outputs)
        for p in partial:
            flat.extend(p.values())
        Lowered.__init__(self, graph, flat, nspecies)
        self.rate_outputs = list(outputs)

    def evaluate_all(self, T, P, x, C, U=()):
        """(rates, [{input: d rate / d input}]) on the host - for the tests of the gradient itself."""
        vals = self.evaluate(T, P, x, C, U)
        R = self.n_rates
        out, pos = [], R
        for p in self.partial:
            out.append({k: vals[pos + n] for n, k in enumerate(p)})
            pos += len(p)
        return vals[:R], out

    def emit_jac(self, fname="rmt_kinetics_jac", with_p=False):
        """Device function: rates r[R] and the partials drdT[R], drdx[R][S], drdC[R][S] (zeros written
        as literals, which the compiler then removes from the chain rule of the node Jacobian); with_p: also
        drdP[R] (model N1, where the pressure is a state variable - the gradient must have been taken by "P" too)."""
        lines, name = self._emit_body(nocheck_from=self.n_primal)
        S, R = self.S, self.n_rates
        outs = ["    r[%d] = %s;" % (k, name[o]) for k, o in enumerate(self.rate_outputs)]
        for q, p in enumerate(self.partial):
            outs.append("    drdT[%d] = %s;" % (q, name[p["T"]] if "T" in p else "real(0)"))
            if with_p:
                outs.append("    drdP[%d] = %s;" % (q, name[p["P"]] if "P" in p else "real(0)"))
            for i in range(S):
                outs.append("    drdx[%d][%d] = %s;" % (q, i, name[p["x%d" % i]] if ("x%d" % i) in p else "real(0)"))
                outs.append("    drdC[%d][%d] = %s;" % (q, i, name[p["C%d" % i]] if ("C%d" % i) in p else "real(0)"))
        return (
            "template <typename FL>\n"
            "__device__ __forceinline__ void %s(const real T, const real invT, const real P,\n"
            "        const real* __restrict__ x, const real* __restrict__ C, const real* __restrict__ U,\n"
            "        real* __restrict__ r,\n"
            "        real* __restrict__ drdT, real (*__restrict__ drdx)[RMT_S], real (*__restrict__ drdC)[RMT_S],%s FL& flag) {\n"
            "    (void)invT; (void)U;\n%s\n%s\n}\n"
            % (fname, " real* __restrict__ drdP," if with_p else "", "\n".join(lines), "\n".join(outs)))


def is_scalar_constant(v):
    """A VARS entry that reactionRateExe copies as a plain number (rmtReaction.py:44-51)."""
    return _num(v) or (isinstance(v, np.ndarray) and v.ndim == 0 and v.dtype.kind in "fiu")


def trace(VARS, RATES, nspecies, R_CONST=8.314472, fixed=None, params=()):
    """Symbolic run of reactionRateExe (rmtReaction.py:27-58).  ``fixed`` optionally binds inputs
    (e.g. {"T": 523.0}) to literals so that everything depending only on them is folded.
    ``params``: names of scalar VARS entries that stay SYMBOLIC - input ``u<k>`` for params[k] - instead of
    being folded as literals: per-reactor kinetic parameters of an ensemble (a sweep over a catalyst density,
    a pre-exponential factor, an activation energy), read by the kernel from the member row."""
    g = Graph()
    fixed = fixed or {}
    params = list(params)
    for nm in params:
        if nm not in VARS or not is_scalar_constant(VARS[nm]):
            raise LoweringError("VARS[%r] cannot be a per-reactor parameter: it is not a scalar constant" % (nm,))

    def leaf(nm):
        return g.const(fixed[nm]) if nm in fixed else g.inp(nm)

    loopDict = {
        "R_CONST": R_CONST,
        "T": leaf("T"),
        "P": leaf("P"),
        "MoFri": SymVector([g.inp("x%d" % i) for i in range(nspecies)]),
        "SpCoi": SymVector([g.inp("C%d" % i) for i in range(nspecies)]),
    }
    merged = {**loopDict, **VARS}
    memo = {}
    exe = {}
    for k, v in merged.items():
        if isinstance(v, types.FunctionType):
            try:
                exe[k] = rebind(v, memo)(exe)
            except LoweringError:
                raise
            except Exception as e:
                raise LoweringError("cannot trace VARS[%r]: %s: %s" % (k, type(e).__name__, e)) from e
        elif k in params:
            exe[k] = g.inp("u%d" % params.index(k))
        else:
            exe[k] = v
    outs = []
    for k, fn in RATES.items():
        try:
            val = rebind(fn, memo)(exe)
        except LoweringError:
            raise
        except Exception as e:
            raise LoweringError("cannot trace RATES[%r]: %s: %s" % (k, type(e).__name__, e)) from e
        if not isinstance(val, Sym):
            if not _num(val):
                raise LoweringError("RATES[%r] returned %r" % (k, type(val)))
            val = g.const(float(val))
        outs.append(val.i)
    return Lowered(g, outs, nspecies)
