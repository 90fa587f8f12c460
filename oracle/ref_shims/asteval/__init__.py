"""Stand-in for asteval: a symbol table plus parse/eval of small Python snippets.

Supports what mcmc_dynamics/parameter.py asks of it: `symtable`, `parse`, `eval` /
`__call__` of expressions and of one-line assignments ('n=5', 'val=1.000000'),
`error`, `error_msg`, `raise_exception`, `user_defined_symbols`, and the helpers
`get_ast_names`, `valid_symbol_name`.
"""
import ast
import builtins
import keyword
import math

import numpy as np

_BASE = {}
for _n in ("sin", "cos", "tan", "arcsin", "arccos", "arctan", "arctan2", "exp", "log", "log10",
           "sqrt", "abs", "pi", "e", "inf", "nan", "where", "minimum", "maximum"):
    _BASE[_n] = getattr(np, _n)
for _n in ("min", "max", "float", "int", "len", "sum", "range", "True", "False", "None"):
    if hasattr(builtins, _n):
        _BASE[_n] = getattr(builtins, _n)
_BASE["math"] = math
_BASE["np"] = np


class Interpreter(object):
    def __init__(self, *args, **kwargs):
        self.symtable = dict(_BASE)
        self._builtin_names = set(self.symtable)
        self.error = []
        self.error_msg = None

    def user_defined_symbols(self):
        return set(k for k in self.symtable if k not in self._builtin_names)

    def parse(self, text):
        try:
            return ast.parse(text.strip())
        except SyntaxError as exc:
            self.error.append(exc)
            self.error_msg = str(exc)
            return None

    def run(self, node):
        if node is None:
            return None
        result = None
        try:
            for stmt in node.body:
                if isinstance(stmt, ast.Expr):
                    code = compile(ast.Expression(stmt.value), "<asteval-standin>", "eval")
                    result = eval(code, {"__builtins__": {}}, self.symtable)
                else:
                    code = compile(ast.Module([stmt], []), "<asteval-standin>", "exec")
                    exec(code, {"__builtins__": {}}, self.symtable)
                    result = None
        except Exception as exc:  # asteval collects errors rather than raising
            self.error.append(exc)
            self.error_msg = str(exc)
            return None
        return result

    def eval(self, expr, **kwargs):
        if isinstance(expr, str):
            expr = self.parse(expr)
        return self.run(expr)

    __call__ = eval

    def raise_exception(self, node, exc=None, msg="", **kwargs):
        err = self.error[0] if self.error else RuntimeError(msg or "asteval stand-in error")
        self.error = []
        if isinstance(err, BaseException):
            raise err
        raise RuntimeError(str(err))


def get_ast_names(astnode):
    if astnode is None:
        return []
    return sorted(set(n.id for n in ast.walk(astnode) if isinstance(n, ast.Name)))


def valid_symbol_name(name):
    return isinstance(name, str) and name.isidentifier() and not keyword.iskeyword(name)
