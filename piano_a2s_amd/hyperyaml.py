"""Minimal HyperPyYAML-compatible loader for the recipe's hparams files (host side).

The reference loads ``hparams/*.yaml`` with ``hyperpyyaml.load_hyperpyyaml(fin, overrides)`` (reference pretrain.py:259-260;
hyperpyyaml 1.2.2, environment.yaml -- third-party, not installed here).  This module implements the subset those two files use
(SURVEY.md section 5 "config / flags"), with the same observable results:

  !ref <key>                      reference to another top-level (or dotted) key; the SAME object is returned every time
  !ref <a>/<b>.txt                string interpolation          !ref <a> * <b> + 1    arithmetic on numbers
  !new:pkg.Class  {kwargs}|[args] instantiate                   !name:pkg.func {kw}   functools.partial
  !apply:pkg.func [args]|{kw}     call at load time             !PLACEHOLDER          must be overridden
  "(398, 189)"                    plain scalars that look like tuples become tuples
  overrides                       dict or YAML string, applied to top-level keys before resolution

``speechbrain.*`` class paths are served by piano_a2s_amd.sb_compat when SpeechBrain itself is not importable.
"""
import ast
import functools
import importlib
import operator
import re

import yaml

_SB_ALIASES = {
    "speechbrain.utils.epoch_loop.EpochCounter": "piano_a2s_amd.sb_compat.EpochCounter",
    "speechbrain.processing.features.InputNormalization": "piano_a2s_amd.sb_compat.InputNormalization",
    "speechbrain.nnet.schedulers.NewBobScheduler": "piano_a2s_amd.sb_compat.NewBobScheduler",
    "speechbrain.utils.checkpoints.Checkpointer": "piano_a2s_amd.sb_compat.Checkpointer",
    "speechbrain.utils.train_logger.FileTrainLogger": "piano_a2s_amd.sb_compat.FileTrainLogger",
}


class _Tagged:
    def __init__(self, kind, target, value):
        self.kind, self.target, self.value = kind, target, value


class _Ref:
    def __init__(self, expr):
        self.expr = expr


class _Placeholder:
    pass


class _Loader(yaml.SafeLoader):
    pass


def _multi(kind):
    def construct(loader, suffix, node):
        if isinstance(node, yaml.MappingNode):
            value = loader.construct_mapping(node, deep=True)
        elif isinstance(node, yaml.SequenceNode):
            value = loader.construct_sequence(node, deep=True)
        else:
            s = loader.construct_scalar(node)
            value = None if s in ("", None) else s
        return _Tagged(kind, suffix, value)
    return construct


for _k in ("new", "name", "apply"):
    _Loader.add_multi_constructor(f"!{_k}:", _multi(_k))
_Loader.add_constructor("!ref", lambda loader, node: _Ref(loader.construct_scalar(node)))
_Loader.add_constructor("!PLACEHOLDER", lambda loader, node: _Placeholder())
_Loader.add_constructor("!tuple", lambda loader, node: tuple(loader.construct_sequence(node, deep=True)))

_TUPLE_RE = re.compile(r"^\(\s*-?[\d.]+(\s*,\s*-?[\d.]+)*\s*,?\s*\)$")
_REF_RE = re.compile(r"<([A-Za-z_][\w.\[\]]*)>")
_OPS = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv,
        ast.FloorDiv: operator.floordiv, ast.Pow: operator.pow, ast.Mod: operator.mod, ast.USub: operator.neg}


def _arith(expr):
    def ev(n):
        if isinstance(n, ast.Expression):
            return ev(n.body)
        if isinstance(n, ast.Constant) and isinstance(n.value, (int, float)):
            return n.value
        if isinstance(n, ast.BinOp) and type(n.op) in _OPS:
            return _OPS[type(n.op)](ev(n.left), ev(n.right))
        if isinstance(n, ast.UnaryOp) and type(n.op) in _OPS:
            return _OPS[type(n.op)](ev(n.operand))
        raise ValueError(expr)
    return ev(ast.parse(expr, mode="eval"))


def _import(path):
    try:
        import speechbrain  # noqa: F401
    except Exception:  # noqa: BLE001  (SpeechBrain absent: serve its classes from the compat layer)
        path = _SB_ALIASES.get(path, path)
    mod, _, attr = path.rpartition(".")
    obj = importlib.import_module(mod)
    return getattr(obj, attr)


class _Resolver:
    def __init__(self, raw):
        self.raw = raw
        self.done = {}
        self.busy = set()

    def key(self, dotted):
        """Value of a (possibly dotted / indexed) top-level key, resolved once and memoised."""
        head, *rest = re.split(r"\.", dotted)
        idx = None
        m = re.match(r"(\w+)\[(\d+)\]$", head)
        if m:
            head, idx = m.group(1), int(m.group(2))
        if head not in self.done:
            if head not in self.raw:
                raise KeyError(f"!ref <{dotted}>: unknown key")
            if head in self.busy:
                raise ValueError(f"circular reference through <{head}>")
            self.busy.add(head)
            self.done[head] = self.value(self.raw[head], where=head)
            self.busy.discard(head)
        v = self.done[head]
        if idx is not None:
            v = v[idx]
        for part in rest:
            v = v[part] if isinstance(v, dict) else getattr(v, part)
        return v

    def ref(self, expr):
        expr = expr.strip()
        whole = _REF_RE.fullmatch(expr)
        if whole:
            return self.key(whole.group(1))
        parts = _REF_RE.findall(expr)
        vals = {p: self.key(p) for p in parts}
        if parts and all(isinstance(v, (int, float)) and not isinstance(v, bool) for v in vals.values()):
            try:
                return _arith(_REF_RE.sub(lambda m: repr(vals[m.group(1)]), expr))
            except (ValueError, SyntaxError):
                pass
        return _REF_RE.sub(lambda m: str(vals[m.group(1)]), expr)

    def value(self, v, where=""):
        if isinstance(v, _Placeholder):
            raise ValueError(f"'{where}' is a !PLACEHOLDER and must be overridden (e.g. --{where}=...)")
        if isinstance(v, _Ref):
            return self.ref(v.expr)
        if isinstance(v, _Tagged):
            target = _import(v.target)
            val = self.value(v.value, where) if v.value is not None else None
            args, kwargs = (), {}
            if isinstance(val, dict):
                kwargs = val
            elif isinstance(val, (list, tuple)):
                args = tuple(val)
            elif val is not None:
                args = (val,)
            if v.kind == "name":
                return functools.partial(target, *args, **kwargs) if (args or kwargs) else target
            return target(*args, **kwargs)                     # !new / !apply
        if isinstance(v, dict):
            return {k: self.value(x, f"{where}.{k}") for k, x in v.items()}
        if isinstance(v, list):
            return [self.value(x, where) for x in v]
        if isinstance(v, str) and _TUPLE_RE.match(v.strip()):
            return tuple(ast.literal_eval(v.strip()))
        return v


def load_hyperpyyaml(stream, overrides=None):
    """Drop-in for hyperpyyaml.load_hyperpyyaml for the recipe's files.  Returns a dict of resolved top-level values;
    keys are resolved in file order (so the seeding ``!apply`` lines at the top run first, as in the reference)."""
    text = stream.read() if hasattr(stream, "read") else stream
    raw = yaml.load(text, Loader=_Loader) or {}
    if overrides:
        if isinstance(overrides, str):
            overrides = yaml.load(overrides, Loader=_Loader) or {}
        for k, v in overrides.items():
            raw[k] = v
    res = _Resolver(raw)
    return {k: res.key(k) for k in raw}
