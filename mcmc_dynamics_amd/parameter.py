"""Model parameters: ordered collection with bounds, priors, initial-value recipes and constraints.

Host-side counterpart of the reference's ``mcmc_dynamics/parameter.py`` (``Parameters`` / ``Parameter``),
re-implemented without lmfit / asteval / astropy.  Same public surface and the same JSON layout
(``{"unique_symbols": {"rng_seed": ...}, "params": [[name, value, unit, fixed, min, max, label,
initials, lnprior, user_data, expr], ...]}``, parameter.py:844-863, 465-466) so that parameter files
written for the reference load here unchanged.

What the hot path needs from this module: the ordered list of free / fixed
parameters, their units and INCLUSIVE ``[min, max]`` bounds (parameter.py:691), plus vectorised forms
for batches of walkers.
"""
import json
import keyword
import logging
import pathlib
from collections import OrderedDict

import numpy as np

from . import units

logger = logging.getLogger(__name__)

_STATE_FIELDS = ("name", "value", "unit", "fixed", "min", "max", "label", "initials", "lnprior", "user_data", "expr")


def _expression_namespace():
    ns = {}
    for name in ("sin", "cos", "tan", "arcsin", "arccos", "arctan", "arctan2", "sinh", "cosh", "tanh", "exp", "log",
                 "log10", "log2", "sqrt", "abs", "hypot", "where", "minimum", "maximum", "clip", "floor", "ceil",
                 "power", "isfinite", "deg2rad", "rad2deg", "pi", "e", "inf", "nan"):
        ns[name] = getattr(np, name)
    ns.update({"asin": np.arcsin, "acos": np.arccos, "atan": np.arctan, "atan2": np.arctan2, "ln": np.log,
               "min": min, "max": max, "float": float, "int": int, "len": len, "True": True, "False": False,
               "None": None})
    try:                                   # the reference exposes these three (parameter.py:19-21)
        from scipy import stats
        for name in ("uniform", "norm", "lognorm"):
            ns[name] = getattr(stats, name)
    except Exception:                      # pragma: no cover - scipy is present in the target image
        pass
    return ns


class ExpressionError(ValueError):
    """An ``initials`` / ``lnprior`` / ``expr`` string failed to parse or evaluate."""


_ALLOWED_NODES = None


def _allowed_nodes():
    import ast
    global _ALLOWED_NODES
    if _ALLOWED_NODES is None:
        names = ["Expression", "BinOp", "UnaryOp", "BoolOp", "Compare", "IfExp", "Call", "keyword", "Name", "Load", "Constant",
                 "Attribute", "Subscript", "Slice", "Tuple", "List",
                 "Add", "Sub", "Mult", "Div", "FloorDiv", "Mod", "Pow", "USub", "UAdd", "Not", "And", "Or",
                 "Eq", "NotEq", "Lt", "LtE", "Gt", "GtE", "Index"]
        _ALLOWED_NODES = tuple(getattr(ast, n) for n in names if hasattr(ast, n))
    return _ALLOWED_NODES


# Attribute names an expression may use, whatever the object: the sampling methods of ``rng`` (numpy.random.Generator), the
# distribution methods of scipy's ``uniform`` / ``norm`` / ``lognorm``, and value-like attributes / reductions of numbers and
# arrays.  Everything else is refused -- in particular ``rng.bit_generator`` (whose ``.ctypes`` / ``.cffi`` interfaces hand out
# raw function pointers: ``rng.bit_generator.ctypes.next_double(12345)`` calls into an arbitrary address).
_ALLOWED_ATTRIBUTES = frozenset((
    # numpy.random.Generator
    "random", "uniform", "normal", "standard_normal", "lognormal", "exponential", "standard_exponential", "gamma",
    "standard_gamma", "beta", "chisquare", "standard_t", "standard_cauchy", "triangular", "laplace", "logistic", "gumbel",
    "rayleigh", "weibull", "pareto", "power", "vonmises", "wald", "poisson", "binomial", "integers", "choice", "permutation",
    # scipy.stats distributions
    "pdf", "logpdf", "cdf", "logcdf", "sf", "logsf", "ppf", "isf", "rvs", "mean", "median", "std", "var", "interval",
    # numbers and arrays
    "real", "imag", "size", "shape", "ndim", "T", "sum", "prod", "min", "max", "clip", "mean", "std"))


def check_expression(text):
    """Parse ``text`` and accept only plain arithmetic: numbers, names, arithmetic / comparison / boolean operators,
    conditional expressions, indexing, tuples / lists, calls, and the attributes of ``_ALLOWED_ATTRIBUTES``.  The
    reference evaluates these strings with asteval, which refuses dunder attributes and has no ``import`` / ``lambda`` /
    comprehension escape routes into the interpreter; ``eval`` with an empty ``__builtins__`` alone is not a sandbox
    (``().__class__.__base__.__subclasses__()`` reaches ``subprocess.Popen``), so the tree is validated before it is
    compiled.  This is a validator for trusted-but-fallible parameter files, narrower than asteval in what it lets
    through, not a claim of equivalence with asteval's sandbox.  Integer constants under ``**`` become floats (``9**9**9``
    is an OverflowError, not an hour of big-integer arithmetic), and a list / tuple literal cannot be repeated with ``*``.
    Returns the parsed ``ast.Expression``."""
    import ast
    try:
        tree = ast.parse(text.strip(), "<parameter expression>", "eval")
    except SyntaxError as exc:
        raise ExpressionError("cannot parse expression '{0}': {1}".format(text, exc))
    allowed = _allowed_nodes()
    for node in ast.walk(tree):
        if not isinstance(node, allowed):
            raise ExpressionError("'{0}' is not allowed in a parameter expression: '{1}'".format(type(node).__name__, text))
        if isinstance(node, ast.Attribute) and (node.attr.startswith("_") or node.attr not in _ALLOWED_ATTRIBUTES):
            raise ExpressionError("attribute '{0}' is not allowed in a parameter expression: '{1}'".format(node.attr, text))
        if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Mult) and \
                (isinstance(node.left, (ast.List, ast.Tuple)) or isinstance(node.right, (ast.List, ast.Tuple))):
            raise ExpressionError("a list / tuple cannot be repeated in a parameter expression: '{0}'".format(text))
        if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Pow):
            for side in ("left", "right"):
                operand = getattr(node, side)
                if isinstance(operand, ast.Constant) and isinstance(operand.value, int) and not isinstance(operand.value, bool):
                    setattr(node, side, ast.copy_location(ast.Constant(float(operand.value)), operand))
        if isinstance(node, ast.Name) and node.id.startswith("_"):
            raise ExpressionError("name '{0}' is not allowed in a parameter expression: '{1}'".format(node.id, text))
        if isinstance(node, ast.Constant) and isinstance(node.value, (str, bytes)):
            raise ExpressionError("string constants are not allowed in a parameter expression: '{0}'".format(text))
        if isinstance(node, ast.Call):
            root = node.func
            while isinstance(root, ast.Attribute):
                root = root.value
            if not isinstance(root, ast.Name):                      # e.g. (lambda: ...)(), f()(), [..][0]()
                raise ExpressionError("only named functions can be called in a parameter expression: '{0}'".format(text))
    return tree


class _Evaluator(object):
    """Symbol table + evaluation of the small Python expressions found in parameter files (``initials``, ``lnprior``,
    ``expr``).  Expressions are validated by ``check_expression`` and may only call what the symbol table holds: the
    NumPy functions of ``_expression_namespace``, ``rng`` (a ``numpy.random.Generator``), scipy's ``uniform`` / ``norm`` /
    ``lognorm`` and user symbols -- the set the reference registers with asteval (parameter.py:17-21, 73-74)."""

    def __init__(self, rng_seed=None, usersyms=None):
        self._base = _expression_namespace()
        self.symtable = dict(self._base)
        if usersyms:
            self.symtable.update(usersyms)
        self.symtable["rng_seed"] = rng_seed
        self.symtable["rng"] = np.random.default_rng(rng_seed)

    def user_defined_symbols(self):
        return set(k for k in self.symtable if k not in self._base)

    def compile(self, text):
        tree = check_expression(text)
        return compile(tree, "<parameter expression>", "eval")

    def __call__(self, code, **local):
        if isinstance(code, str):
            code = self.compile(code)
        try:
            return eval(code, {"__builtins__": {}}, dict(self.symtable, **local))
        except ExpressionError:
            raise
        except Exception as exc:
            raise ExpressionError("cannot evaluate expression: {0}".format(exc))


def valid_symbol_name(name):
    return isinstance(name, str) and name.isidentifier() and not keyword.iskeyword(name)


class Parameter(object):
    """One model parameter (reference: parameter.py:558-1007)."""

    def __init__(self, name, value=None, unit=None, fixed=False, min=-np.inf, max=np.inf, label=None,
                 initials=None, lnprior=None, expr=None, user_data=None):
        self.name = name
        self.fixed = bool(fixed)
        self.min = min
        self.max = max
        self.user_data = user_data
        self.unit = None
        self._value = None
        self._label = label
        self._eval = None            # set by Parameters.__setitem__
        self._initials = None
        self._lnprior = None
        self._expr = None
        self._codes = {}
        self._set_unit(unit)
        self._set_value(value)
        self._init_bounds()
        self.initials = initials
        self.lnprior = lnprior
        self.expr = expr

    # ------------------------------------------------------------------ attribute plumbing
    def set(self, value=None, unit=None, fixed=None, min=None, max=None, label=None, initials=None, lnprior=None,
            expr=None):
        """Update selected attributes (reference: parameter.py:589-619)."""
        if unit is not None:
            self._set_unit(unit)
        if value is not None:
            self._set_value(value)
        if fixed is not None:
            self.fixed = bool(fixed)
        if min is not None:
            self.min = min
        if max is not None:
            self.max = max
        self._init_bounds()
        if initials is not None:
            self.initials = initials
        if lnprior is not None:
            self.lnprior = lnprior
        if expr is not None:
            self.expr = expr
        if label is not None:
            self._label = label

    def _set_unit(self, unit):
        name = units.unit_name(unit)
        if name is None:
            return
        if self.unit is None:
            self.unit = name
        elif name != self.unit:
            logger.error("Cannot change unit from '%s' to '%s'.", self.unit, name)

    def _set_value(self, val):
        plain, src = units.split(val)
        if src is not None:
            if self.unit is None:
                self.unit = src
            else:
                try:
                    plain = np.asarray(plain, dtype=np.float64) * units.conversion_factor(src, self.unit)
                except ValueError:
                    raise IOError("Unit '{0}' of new value incompatible with existing unit '{1}'.".format(src, self.unit))
        if plain is not None and np.ndim(plain) == 0:
            plain = float(plain)
        self._value = plain
        if self._eval is not None and self.name:
            self._eval.symtable[self.name] = self._value

    def _bound(self, b, default):
        if b is None:
            return default
        plain, src = units.split(b)
        if src is not None:
            if self.unit is None:
                self.unit = src
            try:
                return float(plain) * units.conversion_factor(src, self.unit)
            except ValueError:
                raise IOError("Incompatible units provided for a bound of parameter '{0}'.".format(self.name))
        return float(plain)

    def _init_bounds(self):
        """Self-consistent bounds and a value inside them (reference: parameter.py:773-806)."""
        self.min = self._bound(self.min, -np.inf)
        self.max = self._bound(self.max, np.inf)
        if self.min > self.max:
            self.min, self.max = self.max, self.min
        if np.isclose(self.min, self.max, atol=1e-13, rtol=1e-13):
            raise ValueError("Parameter '{0}' has min == max".format(self.name))
        if self._value is None:
            self._value = 0.5 * (self.min + self.max) if np.isfinite(self.min) and np.isfinite(self.max) else 0.0
        if np.ndim(self._value) == 0:
            self._value = float(np.clip(self._value, self.min, self.max))

    # ------------------------------------------------------------------ expressions
    def _code(self, kind, text):
        if text is None or self._eval is None:
            return None
        key = (kind, text)
        if key not in self._codes:
            self._codes[key] = self._eval.compile(text)
        return self._codes[key]

    @property
    def initials(self):
        return self._initials

    @initials.setter
    def initials(self, val):
        self._initials = val or None
        self._code("initials", self._initials)

    @property
    def lnprior(self):
        return self._lnprior

    @lnprior.setter
    def lnprior(self, val):
        self._lnprior = val or None
        self._code("lnprior", self._lnprior)

    @property
    def expr(self):
        return self._expr

    @expr.setter
    def expr(self, val):
        self._expr = val or None
        if self._expr is not None:
            self.fixed = True            # a constrained parameter is never sampled (parameter.py:725-726)
        self._code("expr", self._expr)

    @property
    def value(self):
        """Current value; a constrained parameter is re-evaluated from its expression."""
        if self._expr is not None and self._eval is not None:
            self._value = self._eval(self._code("expr", self._expr))
        return self._value

    @value.setter
    def value(self, val):
        self._set_value(val)

    def evaluate_initials(self, n):
        """Draw ``n`` starting values (reference: parameter.py:642-661)."""
        if self._initials is not None:
            if self._eval is None:
                raise IOError("Cannot evaluate 'initials' expression: '{0}'".format(self._initials))
            return np.asarray(self._eval(self._code("initials", self._initials), n=int(n)), dtype=np.float64)
        from scipy import stats
        loc, scale = self.value, 1.0
        if self.min == -np.inf and self.max == np.inf:
            fct = stats.norm(loc=loc, scale=scale)
        else:
            fct = stats.truncnorm((self.min - loc) / scale, (self.max - loc) / scale, loc=loc, scale=scale)
        return fct.rvs(int(n))

    def evaluate_lnprior(self, val):
        """0 inside the inclusive bounds, -inf outside, or the ``lnprior`` expression
        (reference: parameter.py:684-705)."""
        val = float(units.to_unit(val, self.unit))
        if val < self.min or val > self.max:
            return -np.inf
        if self._lnprior is None:
            return 0
        if self._eval is None:
            raise IOError("Cannot evaluate expression: '{0}'".format(self._lnprior))
        # the reference hands the value over as 'val={:f}' (parameter.py:698): six decimals
        return self._eval(self._code("lnprior", self._lnprior), val=float("{0:f}".format(val)))

    # ------------------------------------------------------------------ misc
    @property
    def label(self):
        text = self._label if self._label is not None else r"${{\rm {0}}}$".format(self.name)
        return text + ("/" + self.unit if self.unit is not None else "")

    @label.setter
    def label(self, val):
        self._label = val

    def __getstate__(self):
        return (self.name, self._value, self.unit, self.fixed, self.min, self.max, self._label, self._initials,
                self._lnprior, self.user_data, self._expr)

    @classmethod
    def from_state(cls, state):
        s = dict(zip(_STATE_FIELDS, state))
        return cls(s["name"], value=s["value"], unit=s["unit"], fixed=s["fixed"], min=s["min"], max=s["max"],
                   label=s["label"], initials=s["initials"], lnprior=s["lnprior"], expr=s["expr"],
                   user_data=s["user_data"])

    def __float__(self):
        return float(self.value)

    def __repr__(self):
        bits = ["value={0!r}{1}".format(self.value, " (fixed)" if self.fixed and self._expr is None else "")]
        if self.unit is not None:
            bits.append("unit={0}".format(self.unit))
        bits.append("bounds=[{0!r}:{1!r}]".format(self.min, self.max))
        for key in ("initials", "expr", "lnprior"):
            if getattr(self, "_" + key) is not None:
                bits.append("{0}='{1}'".format(key, getattr(self, "_" + key)))
        return "<Parameter '{0}', {1}>".format(self.name, ", ".join(bits))


class Parameters(OrderedDict):
    """Ordered ``name -> Parameter`` mapping with a shared expression evaluator and RNG
    (reference: parameter.py:30-555).  Iteration order defines the order of the free-parameter
    vector that the sampler sees (runner.py:123-127, 162-175)."""

    def __init__(self, usersyms=None, rng_seed=None):
        super().__init__()
        self._asteval = _Evaluator(rng_seed=rng_seed, usersyms=usersyms)

    @property
    def rng(self):
        return self._asteval.symtable["rng"]

    def __setitem__(self, key, par):
        if key not in self and not valid_symbol_name(key):
            raise KeyError("'{0}' is not a valid Parameters name".format(key))
        if not isinstance(par, Parameter):
            raise ValueError("'{0}' is not a Parameter".format(par))
        OrderedDict.__setitem__(self, key, par)
        par.name = key
        par._eval = self._asteval
        self._asteval.symtable[key] = par._value

    def add(self, name, value=None, unit=None, fixed=False, min=-np.inf, max=np.inf, label=None, initials=None,
            lnprior=None, expr=None):
        if isinstance(name, Parameter):
            self[name.name] = name
        else:
            self[name] = Parameter(name, value=value, unit=unit, fixed=fixed, min=min, max=max, label=label,
                                   initials=initials, lnprior=lnprior, expr=expr)

    def add_many(self, *parlist):
        for par in parlist:
            if not isinstance(par, Parameter):
                par = Parameter(*par)
            self[par.name] = par

    def copy(self):
        return self.__deepcopy__(None)

    def __copy__(self):
        return self.__deepcopy__(None)

    def __deepcopy__(self, memo):
        new = Parameters(rng_seed=self._asteval.symtable.get("rng_seed"))
        for key in self._asteval.user_defined_symbols():
            if key not in ("rng", "rng_seed") and key not in self:
                new._asteval.symtable[key] = self._asteval.symtable[key]
        new.add_many(*[Parameter.from_state(p.__getstate__()) for p in self.values()])
        return new

    def __reduce__(self):
        return (_rebuild, (self.dumps(),))

    def eval(self, expr):
        return self._asteval(expr)

    def valuesdict(self):
        return OrderedDict((p.name, p.value) for p in self.values())

    # ------------------------------------------------------------------ JSON (reference layout)
    def dumps(self, **kws):
        rng = self._asteval.symtable["rng"]
        syms = {"rng_seed": self._asteval.symtable.get("rng_seed")}
        for key in self._asteval.user_defined_symbols():
            val = self._asteval.symtable[key]
            if key not in ("rng", "rng_seed") and key not in self and isinstance(val, (int, float, str, bool, type(None))):
                syms[key] = val
        state = _jsonable(rng.bit_generator.state)
        return json.dumps({"unique_symbols": syms, "random_state": state,
                           "params": [list(p.__getstate__()) for p in self.values()]}, **kws)

    def loads(self, s, **kws):
        self.clear()
        tmp = json.loads(s, **kws)
        syms = tmp.get("unique_symbols", {}) or {}
        seed = syms.get("rng_seed")
        self._asteval = _Evaluator(rng_seed=seed, usersyms={k: v for k, v in syms.items() if k != "rng_seed"})
        state = tmp.get("random_state")
        if state is not None:
            try:
                self._asteval.symtable["rng"].bit_generator.state = state
            except Exception:
                logger.warning("Could not restore the random-number generator state from the parameter file.")
        self.add_many(*[Parameter.from_state(st) for st in tmp["params"]])
        return self

    def dump(self, fp, **kws):
        return fp.write(self.dumps(**kws))

    def load(self, fp, **kws):
        if isinstance(fp, (str, pathlib.Path)):
            return self.loads(pathlib.Path(fp).read_text(), **kws)
        return self.loads(fp.read(), **kws)

    # ------------------------------------------------------------------ display
    def pretty_print(self, oneline=False, colwidth=10, precision=4, fmt="g", columns=None):
        if oneline:
            print(OrderedDict.__repr__(self))
            return
        columns = columns or ["value", "unit", "min", "max", "fixed", "initials", "lnprior"]
        width = max(len(k) for k in self) if len(self) else 4
        print(" ".join(["{0:<{1}}".format("Name", width)] + ["{0:>{1}}".format(c.title(), colwidth) for c in columns]))
        for name in sorted(self):
            row = ["{0:<{1}}".format(name, width)]
            for c in columns:
                v = getattr(self[name], c)
                if isinstance(v, float):
                    row.append("{0:>{1}.{2}{3}}".format(v, colwidth, precision, fmt))
                else:
                    row.append("{0!s:>{1}}".format(v, colwidth))
            print(" ".join(row))

    # ------------------------------------------------------------------ batched helpers (new)
    def free_names(self):
        return [k for k, p in self.items() if not p.fixed]

    def bounds(self):
        """(min, max) arrays over ALL parameters in iteration order."""
        return (np.array([p.min for p in self.values()], dtype=np.float64),
                np.array([p.max for p in self.values()], dtype=np.float64))

    def resolve_batch(self, values):
        """(W, P_free) sampler positions -> OrderedDict name -> (W,) array over ALL parameters: free
        values scattered in iteration order, fixed values broadcast, ``expr``-constrained parameters
        evaluated on the arrays (vectorised form of runner.py:143-180 + parameter.py:865-874)."""
        values = np.atleast_2d(np.asarray(values, dtype=np.float64))
        w = values.shape[0]
        free = self.free_names()
        if values.shape[1] != len(free):
            raise AssertionError("Not all parameters used.")      # runner.py:178
        out = OrderedDict()
        i = 0
        for name, par in self.items():
            if par.fixed:
                if par._expr is None:
                    out[name] = np.full(w, float(par._value))
            else:
                out[name] = values[:, i]
                i += 1
        pending = [name for name, par in self.items() if par._expr is not None]
        for _ in range(len(pending) + 1):
            if not pending:
                break
            rest = []
            for name in pending:
                try:
                    val = self._asteval(self[name]._code("expr", self[name]._expr), **out)
                    out[name] = np.broadcast_to(np.asarray(val, dtype=np.float64), (w,)).copy()
                except ExpressionError:
                    rest.append(name)
            if len(rest) == len(pending):
                raise ExpressionError("cannot resolve constrained parameter(s): {0}".format(rest))
            pending = rest
        return OrderedDict((name, out[name]) for name in self)

    def lnprior_batch(self, resolved):
        """Vectorised ``Runner.lnprior``: sum of per-parameter priors, -inf outside inclusive bounds
        (runner.py:206-217: every parameter is checked, fixed ones included)."""
        w = len(next(iter(resolved.values()))) if resolved else 0
        total = np.zeros(w, dtype=np.float64)
        for name, par in self.items():
            col = resolved[name]
            bad = (col < par.min) | (col > par.max) | np.isnan(col)
            total[bad] = -np.inf
            if par._lnprior is not None:
                good = np.flatnonzero(~bad & np.isfinite(total))
                for j in good:                      # expression priors are rare: evaluate per walker
                    total[j] += float(par._eval(par._code("lnprior", par._lnprior),
                                                val=float("{0:f}".format(col[j]))))
        total[~np.isfinite(total)] = -np.inf
        return total


def _jsonable(obj):
    if isinstance(obj, dict):
        return {k: _jsonable(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_jsonable(v) for v in obj]
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, np.generic):
        return obj.item()
    return obj


def _rebuild(text):
    return Parameters().loads(text)
