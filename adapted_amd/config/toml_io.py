"""Minimal TOML reader/writer for the flat two-level config files of ``adapted detect``
(sections of scalars and short arrays, with ``inf``/``-inf``/``nan`` float literals as in
the reference presets).  Uses ``tomli``/``tomllib``/``toml`` when importable; otherwise a
small built-in parser for exactly this subset."""
from __future__ import annotations

import math
import re
from typing import Any, Dict


def _parse_scalar(tok: str) -> Any:
    tok = tok.strip()
    if tok.startswith('"') and tok.endswith('"'):
        return bytes(tok[1:-1], "utf-8").decode("unicode_escape")
    if tok.startswith("'") and tok.endswith("'"):
        return tok[1:-1]
    if tok == "true":
        return True
    if tok == "false":
        return False
    if tok in ("inf", "+inf"):
        return math.inf
    if tok == "-inf":
        return -math.inf
    if tok in ("nan", "+nan", "-nan"):
        return math.nan
    t = tok.replace("_", "")
    if re.fullmatch(r"[+-]?\d+", t):
        return int(t)
    return float(t)


def _strip_comment(line: str) -> str:
    out, q = [], None
    for ch in line:
        if q:
            out.append(ch)
            if ch == q:
                q = None
        elif ch in "\"'":
            q = ch
            out.append(ch)
        elif ch == "#":
            break
        else:
            out.append(ch)
    return "".join(out).strip()


def _parse_builtin(text: str) -> Dict[str, Any]:
    root: Dict[str, Any] = {}
    cur = root
    pending = ""
    for raw in text.splitlines():
        line = _strip_comment(raw)
        if not line:
            continue
        if pending:
            line = pending + " " + line
            pending = ""
        if line.startswith("["):
            name = line.strip("[] \t")
            cur = root.setdefault(name, {})
            continue
        key, _, val = line.partition("=")
        key, val = key.strip().strip('"'), val.strip()
        if val.startswith("[") and not val.endswith("]"):
            pending = line
            continue
        if val.startswith("["):
            inner = val[1:-1].strip()
            cur[key] = [_parse_scalar(t) for t in inner.split(",") if t.strip()] if inner else []
        else:
            cur[key] = _parse_scalar(val)
    return root


def loads(text: str) -> Dict[str, Any]:
    try:
        import tomllib as _t  # py >= 3.11

        return _t.loads(text)
    except ImportError:
        pass
    try:
        import tomli as _t

        return _t.loads(text)
    except ImportError:
        pass
    try:
        import toml as _t

        return _t.loads(text)
    except ImportError:
        return _parse_builtin(text)


def load(path) -> Dict[str, Any]:
    with open(path, "r", encoding="utf-8") as fh:
        return loads(fh.read())


def _fmt(v: Any) -> str:
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, int):
        return str(v)
    if isinstance(v, float):
        if math.isinf(v):
            return "inf" if v > 0 else "-inf"
        if math.isnan(v):
            return "nan"
        return repr(v)
    if isinstance(v, str):
        return '"' + v.replace("\\", "\\\\").replace('"', '\\"') + '"'
    if isinstance(v, (list, tuple)):
        return "[ " + ", ".join(_fmt(x) for x in v) + ",]" if len(v) else "[]"
    raise TypeError("cannot write %r to TOML" % (v,))


def dumps(d: Dict[str, Any]) -> str:
    lines = []
    for k, v in d.items():
        if not isinstance(v, dict) and v is not None:
            lines.append("%s = %s" % (k, _fmt(v)))
    for k, v in d.items():
        if isinstance(v, dict):
            lines.append("")
            lines.append("[%s]" % k)
            for kk, vv in v.items():
                if vv is not None:
                    lines.append("%s = %s" % (kk, _fmt(vv)))
    return "\n".join(lines).lstrip("\n") + "\n"


def dump(d: Dict[str, Any], path) -> None:
    with open(path, "w", encoding="utf-8") as fh:
        fh.write(dumps(d))
