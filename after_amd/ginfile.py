"""A reader for the subset of gin-config syntax AFTER's `config.gin` files use
(after/diffusion/configs/*.gin, after/autoencoder/configs/baseAE.gin, and the
`gin.operative_config_str()` dumps written next to the checkpoints, model.py:262-265).

gin-config itself is not a dependency of after_amd: the GPU box has no gin, and the hot path
only needs the constructor arguments.  Supported:

    NAME = value                         macro
    [scope/]selector.param = value       flat binding
    [scope/]selector:                    block binding (indented `param = value` lines)
        param = value
    %NAME                                macro reference (resolved lazily, macros can be re-bound)
    @[scope/]selector  /  @...()         configurable reference -> Ref(scope, selector, call)
    include 'file.gin'                   relative to the including file
    import x / from x import y           ignored (dynamic registration)
    # comments, multi-line bracketed values

Selectors match by dotted suffix, as in gin: `Encoder1D` matches
`diffusion.networks.Encoder1D` and `after.diffusion.networks.encoder.Encoder1D`."""
import ast
import os
import re
from typing import Any, Dict, Optional, Tuple


class GinError(ValueError):
    pass


class Macro:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return f"%{self.name}"


class Ref:
    """@[scope/]selector or @[scope/]selector()"""

    def __init__(self, target: str, call: bool):
        parts = target.split("/")
        self.scope = "/".join(parts[:-1])
        self.selector = parts[-1]
        self.call = call

    def __repr__(self):
        s = f"{self.scope}/" if self.scope else ""
        return f"@{s}{self.selector}{'()' if self.call else ''}"


_MACRO_RE = re.compile(r"%([A-Za-z_][A-Za-z0-9_.]*)")
_REF_RE = re.compile(r"@([A-Za-z_][A-Za-z0-9_./]*)(\(\))?")


def _strip_comment(line: str) -> str:
    out, quote = [], None
    for ch in line:
        if quote:
            out.append(ch)
            if ch == quote:
                quote = None
        elif ch in "'\"":
            quote = ch
            out.append(ch)
        elif ch == "#":
            break
        else:
            out.append(ch)
    return "".join(out).rstrip()


def _depth(text: str) -> int:
    d, quote = 0, None
    for ch in text:
        if quote:
            if ch == quote:
                quote = None
        elif ch in "'\"":
            quote = ch
        elif ch in "([{":
            d += 1
        elif ch in ")]}":
            d -= 1
    return d


def _eval_node(node):
    if isinstance(node, ast.Constant):
        return node.value
    if isinstance(node, ast.List):
        return [_eval_node(e) for e in node.elts]
    if isinstance(node, ast.Tuple):
        return tuple(_eval_node(e) for e in node.elts)
    if isinstance(node, ast.Dict):
        return {_eval_node(k): _eval_node(v) for k, v in zip(node.keys, node.values)}
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
        v = _eval_node(node.operand)
        return -v if isinstance(node.op, ast.USub) else v
    if isinstance(node, ast.Call) and isinstance(node.func, ast.Name):
        if node.func.id == "__gin_macro__":
            return Macro(node.args[0].value)
        if node.func.id == "__gin_ref__":
            return Ref(node.args[0].value, node.args[1].value)
    raise GinError(f"unsupported expression in gin value: {ast.dump(node)}")


def _sub_outside_quotes(text: str) -> str:
    """%NAME / @ref -> marker calls, leaving string literals untouched."""
    out, i, n = [], 0, len(text)
    while i < n:
        ch = text[i]
        if ch in "'\"":
            j = i + 1
            while j < n and text[j] != ch:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
            continue
        m = _REF_RE.match(text, i) if ch == "@" else _MACRO_RE.match(text, i) if ch == "%" else None
        if m and ch == "@":
            out.append(f'__gin_ref__("{m.group(1)}", {bool(m.group(2))})')
            i = m.end()
        elif m:
            out.append(f'__gin_macro__("{m.group(1)}")')
            i = m.end()
        else:
            out.append(ch)
            i += 1
    return "".join(out)


def parse_value(text: str):
    src = _sub_outside_quotes(text.strip())
    try:
        tree = ast.parse(src, mode="eval")
    except SyntaxError as e:
        raise GinError(f"cannot parse gin value {text!r}: {e}") from None
    return _eval_node(tree.body)


def _match(binding_selector: str, query: str) -> bool:
    return (binding_selector == query or binding_selector.endswith("." + query)
            or query.endswith("." + binding_selector))


class GinConfig:

    def __init__(self):
        self.macros: Dict[str, Any] = {}
        # (scope, selector) -> {param: raw value}
        self.bindings: Dict[Tuple[str, str], Dict[str, Any]] = {}

    # ------------------------------------------------------------ parsing
    @classmethod
    def parse_file(cls, path: str) -> "GinConfig":
        cfg = cls()
        cfg._parse_file(path)
        return cfg

    @classmethod
    def parse_files(cls, paths) -> "GinConfig":
        """Later files override earlier ones (train.py parses base.gin then e.g. cycle.gin)."""
        cfg = cls()
        for p in paths:
            cfg._parse_file(p)
        return cfg

    @classmethod
    def parse_string(cls, text: str, base_dir: Optional[str] = None) -> "GinConfig":
        cfg = cls()
        cfg._parse(text, base_dir)
        return cfg

    def _parse_file(self, path):
        with open(path) as f:
            self._parse(f.read(), os.path.dirname(os.path.abspath(path)))

    def _bind(self, target: str, value):
        target = target.strip()
        if "." not in target and "/" not in target:
            self.macros[target] = value
            return
        parts = target.split("/")
        scope, last = "/".join(parts[:-1]), parts[-1]
        if "." not in last:
            raise GinError(f"binding {target!r} has no parameter name")
        selector, param = last.rsplit(".", 1)
        self.bindings.setdefault((scope, selector), {})[param] = value

    def _parse(self, text, base_dir):
        lines = text.split("\n")
        i, block = 0, None  # block = "scope/selector" while inside `selector:` block
        while i < len(lines):
            raw = _strip_comment(lines[i])
            i += 1
            if not raw.strip():
                continue
            indented = raw[0] in " \t"
            stmt = raw.strip()
            while stmt.endswith("\\") and i < len(lines):  # operative_config_str line wrapping
                stmt = stmt[:-1].rstrip() + " " + _strip_comment(lines[i]).strip()
                i += 1
            while _depth(stmt) > 0 and i < len(lines):  # multi-line bracketed value
                stmt += " " + _strip_comment(lines[i]).strip()
                i += 1
            if not indented:
                block = None
                if stmt.startswith(("import ", "from ")):
                    continue
                if stmt.startswith("include "):
                    inc = ast.literal_eval(stmt[len("include "):].strip())
                    self._parse_file(inc if os.path.isabs(inc) or base_dir is None else os.path.join(base_dir, inc))
                    continue
                if stmt.endswith(":") and "=" not in stmt:
                    block = stmt[:-1].strip()
                    continue
            if "=" not in stmt:
                raise GinError(f"cannot parse gin line: {raw!r}")
            target, value = stmt.split("=", 1)
            value = parse_value(value)
            if indented and block is not None:
                self._bind(f"{block}.{target.strip()}", value)
            else:
                self._bind(target, value)

    # ------------------------------------------------------------ queries
    def bind(self, target: str, value):
        """gin.bind_parameter: '%NAME' re-binds a macro, 'scope/selector.param' a parameter."""
        if target.startswith("%"):
            self.macros[target[1:]] = value
        else:
            self._bind(target, value)

    def resolve(self, v):
        if isinstance(v, Macro):
            if v.name not in self.macros:
                raise GinError(f"unbound macro %{v.name}")
            return self.resolve(self.macros[v.name])
        if isinstance(v, list):
            return [self.resolve(e) for e in v]
        if isinstance(v, tuple):
            return tuple(self.resolve(e) for e in v)
        if isinstance(v, dict):
            return {k: self.resolve(e) for k, e in v.items()}
        return v

    def macro(self, name: str, default=...):
        """gin.query_parameter('%NAME')"""
        if name.startswith("%"):
            name = name[1:]
        if name not in self.macros:
            if default is ...:
                raise GinError(f"unbound macro %{name}")
            return default
        return self.resolve(self.macros[name])

    def kwargs(self, selector: str, scope: str = "") -> Dict[str, Any]:
        """All bindings that apply to `selector` when called in `scope` (unscoped bindings first,
        then each enclosing scope, innermost last -- gin's precedence)."""
        out: Dict[str, Any] = {}
        scopes = [""]
        if scope:
            parts = scope.split("/")
            # gin: a binding scoped `a` applies inside `a/b` too; longer matches win
            scopes += ["/".join(parts[j:]) for j in range(len(parts) - 1, -1, -1)]
        for sc in scopes:
            for (bscope, bsel), params in self.bindings.items():
                if bscope == sc and _match(bsel, selector):
                    out.update(params)
        return {k: self.resolve(v) for k, v in out.items()}

    def query(self, selector: str, param: str, scope: str = "", default=...):
        kw = self.kwargs(selector, scope)
        if param not in kw:
            if default is ...:
                raise GinError(f"no binding for {scope + '/' if scope else ''}{selector}.{param}")
            return default
        return kw[param]
