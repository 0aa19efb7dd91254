"""Seeded random stCSP models over the whole constraint language (tests only): small domains, a
few constraints, every operator the front end accepts except the ones the reference itself
mis-normalises (`not`, `@` after `next`) and `/`, `%` by a variable
(SIGFPE in the reference). Used to compare implementations with each other on inputs nobody
hand-picked."""
import importlib

_inst = importlib.import_module("stcsp-solver_amd").instances


class Gen:
    def __init__(self, seed):
        self.r = _inst.SplitMix64(0x5EED0000 + seed)
        self.vars = []
        self.arr_len = 0

    def pick(self, xs):
        return xs[self.r.below(len(xs))]

    def chance(self, pct):
        return self.r.below(100) < pct

    def var(self):
        return self.pick(self.vars)

    def atom(self):
        k = self.r.below(100)
        if k < 50:
            return self.var()
        if k < 70:
            return str(self.r.below(4))
        if k < 82:
            return f"next {self.var()}"
        if k < 90:
            return f"first {self.var()}"
        if k < 95:
            return f"({self.var()} fby {self.var()})"
        return f"({self.var()} @ {1 + self.r.below(2)})"

    def expr(self, depth):
        if depth == 0 or self.chance(35):
            return self.atom()
        k = self.r.below(100)
        a, b = self.expr(depth - 1), self.expr(depth - 1)
        if k < 18:
            return f"({a} + {b})"
        if k < 32:
            return f"({a} - {b})"
        if k < 40:
            return f"({a} * {self.r.below(3)})"
        if k < 44:
            return f"({a} % {2 + self.r.below(2)})"
        if k < 46:
            return f"({a} / {1 + self.r.below(3)})"
        if k < 50:
            return f"(abs {a})"
        if k < 78:
            return f"({a} {self.pick(['lt', 'gt', 'le', 'ge', 'eq', 'ne'])} {b})"
        if k < 88:
            return f"({a} {self.pick(['and', 'or'])} {b})"
        if k < 95:
            return f"(if ({a} {self.pick(['lt', 'eq', 'ge'])} {b}) then {self.atom()} else {self.atom()})"
        if self.arr_len:
            return f"T[({a}) % {self.arr_len}]" if self.chance(50) else f"T[{self.var()}]"
        return f"({a} + {b})"

    def model(self):
        n = 2 + self.r.below(3)
        out = []
        for i in range(n):
            lo = self.r.below(3) - 1
            hi = lo + self.r.below(4)
            self.vars.append(f"v{chr(97 + i)}")
            out.append(f"var {self.vars[-1]} : [{lo}, {hi}];")
        if self.chance(30):
            self.arr_len = 2 + self.r.below(3)
            out.append("arr T : {" + ", ".join(str(self.r.below(4)) for _ in range(self.arr_len)) + "};")
        if self.chance(60):  # an initial condition and a transition, like every shipped model
            v = self.var()
            out.append(f"first {v} == {self.r.below(2)};")
        if self.chance(70):
            v = self.var()
            out.append(f"next {v} {self.pick(['==', '>=', '<=', '!='])} {self.expr(1)};")
        for _ in range(1 + self.r.below(3)):
            if self.chance(8):
                out.append(f"{self.var()} until {self.var()};")
                continue
            op = self.pick(["==", "==", "!=", "<", ">", "<=", ">=", "->"])
            out.append(f"{self.expr(2)} {op} {self.expr(2)};")
        return "\n".join(out) + "\n"


def random_model(seed: int) -> str:
    return Gen(seed).model()


class WideGen(Gen):
    """The same language over domains of 33..128 values (bitset blocks of two and four words per variable: the engine's
    W = 2 / 4 kernels): one or two wide variables, the others small, constants that reach into the wide ranges, and
    transitions that walk through them so that the automata are not trivial."""

    def atom(self):
        k = self.r.below(100)
        if k < 50:
            return self.var()
        if k < 62:
            return str(self.r.below(4))
        if k < 72:
            return str(self.r.below(self.span))
        if k < 84:
            return f"next {self.var()}"
        if k < 92:
            return f"first {self.var()}"
        return f"({self.var()} fby {self.var()})"

    def model(self):
        n = 2 + self.r.below(2)
        n_wide = 1 + self.r.below(2)
        out = []
        self.span = 33 + self.r.below(96 if self.chance(50) else 32)  # 33..64 (W = 2) or 33..128 (W = 4)
        for i in range(n):
            lo = self.r.below(3) - 1
            hi = lo + (self.span - 1 if i < n_wide else self.r.below(4))
            self.vars.append(f"v{chr(97 + i)}")
            out.append(f"var {self.vars[-1]} : [{lo}, {hi}];")
        if self.chance(25):
            self.arr_len = 2 + self.r.below(3)
            out.append("arr T : {" + ", ".join(str(self.r.below(4)) for _ in range(self.arr_len)) + "};")
        w = self.vars[0]
        step = 1 + self.r.below(5)
        k = self.r.below(100)
        if k < 70:  # the wide variable walks: few successors per state, many states
            out.append(f"first {w} == {self.r.below(3)};")
            if self.chance(50):
                out.append(f"next {w} == (if ({w} ge {self.span - step - 2}) then {self.r.below(3)} else ({w} + {step}));")
            else:
                out.append(f"next {w} >= {w} + {step - 1};")
                out.append(f"next {w} <= {w} + {step};")
        elif k < 85:
            out.append(f"next {w} {self.pick(['==', '>=', '<='])} {self.expr(1)};")
        if n_wide > 1:  # tie the second wide variable to the first, or it multiplies every state's edges by its whole domain
            d = self.r.below(4)
            out.append(f"{self.vars[1]} {self.pick(['==', '<=', '>='])} {w} {self.pick(['+', '-'])} {d};")
            if self.chance(70):
                out.append(f"{self.vars[1]} {self.pick(['>=', '<='])} {w} {self.pick(['+', '-'])} {self.r.below(3)};")
        for _ in range(1 + self.r.below(2)):
            if self.chance(6):
                out.append(f"{self.var()} until {self.var()};")
                continue
            op = self.pick(["==", "!=", "<", ">", "<=", ">=", "->"])
            out.append(f"{self.expr(1)} {op} {self.expr(2)};")
        return "\n".join(out) + "\n"


def random_wide_model(seed: int) -> str:
    return WideGen(seed).model()
