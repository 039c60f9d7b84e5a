"""ORACLE (test infrastructure only - never imported by the product path).

Tree-walking CEL evaluator: the CPU restatement of what the reference gets from
github.com/google/cel-go v0.30.0 (go.mod:44; NOT vendored under /root/reference) through
``conditions.StdEnv`` (``internal/conditions/cel.go:65-107``) plus the Cerbos function
library (``internal/conditions/cerbos_lib.go:55-245``).

Semantics follow the CEL language definition (cel-spec langdef) with the reference's
environment options: cross-type numeric comparisons (``cel.go:69``), two-variable
comprehensions, ext Strings/Lists/Math/Sets/Encoders/Bindings, frozen ``now``
(``cerbos_lib.go:274-343``). Pinned by the reference's ``internal/test/testdata/cel_eval``
KATs and the engine golden cases (see tests/test_oracle_golden.py); anything those do
not exercise (uint overflow corners, NaN ordering, RE2-vs-Python regex dialect
differences, optional types, hierarchy/SPIFFE types) is parity-unpinned.

Errors are Python exceptions (CelError); ``&&``/``||`` absorb them as CEL requires.
"""
from __future__ import annotations

import base64
import calendar
import datetime
import ipaddress
import math
import re

from .celparse import parse   # the oracle reads CEL text with its own parser (tests/test_oracle_parser.py holds it against the product's)

from . import crosspath

INT_MIN, INT_MAX = -(1 << 63), (1 << 63) - 1
UINT_MAX = (1 << 64) - 1


class CelError(Exception):
    pass


class UInt(int):
    """CEL uint (distinct from int)."""

    def __repr__(self):
        return "%du" % int(self)


class Timestamp:
    __slots__ = ("ns",)

    def __init__(self, ns):
        self.ns = int(ns)

    def __eq__(self, o):
        return isinstance(o, Timestamp) and self.ns == o.ns

    def __hash__(self):
        return hash(("ts", self.ns))

    def __repr__(self):
        return "Timestamp(%d)" % self.ns


class Duration:
    __slots__ = ("ns",)

    def __init__(self, ns):
        self.ns = int(ns)

    def __eq__(self, o):
        return isinstance(o, Duration) and self.ns == o.ns

    def __hash__(self):
        return hash(("dur", self.ns))

    def __repr__(self):
        return "Duration(%d)" % self.ns


class Message:
    """A proto message value (request / principal / resource / aux_data / runtime).
    ``fields`` maps proto field name -> value; ``json_names`` maps JSON name -> proto name
    (conditions/types/registry.go:58-125)."""

    def __init__(self, fields, json_names=None):
        self.fields = fields
        self.json_names = json_names or {}

    def resolve(self, name):
        if name in self.fields:
            return name
        return self.json_names.get(name)


class Variables:
    """types.VariablesMap (conditions/types/variables.go:22-38)."""

    def __init__(self, values):
        self.values = values


def no_such_overload():
    return CelError("no such overload")


# ---------------------------------------------------------------- type helpers

def is_int(v):
    return isinstance(v, int) and not isinstance(v, (bool, UInt))


def is_uint(v):
    return isinstance(v, UInt)


def is_double(v):
    return isinstance(v, float)


def is_num(v):
    return (isinstance(v, (int, float))) and not isinstance(v, bool)


def type_name(v):
    if v is None:
        return "null_type"
    if isinstance(v, bool):
        return "bool"
    if is_uint(v):
        return "uint"
    if isinstance(v, int):
        return "int"
    if isinstance(v, float):
        return "double"
    if isinstance(v, str):
        return "string"
    if isinstance(v, bytes):
        return "bytes"
    if isinstance(v, list):
        return "list"
    if isinstance(v, dict):
        return "map"
    if isinstance(v, Timestamp):
        return "google.protobuf.Timestamp"
    if isinstance(v, Duration):
        return "google.protobuf.Duration"
    if isinstance(v, Hierarchy):
        return "cerbos.lib.hierarchy"
    return type(v).__name__


def _num_cmp(a, b):
    """Mathematical comparison across int/uint/double; None if unordered (NaN)."""
    if isinstance(a, float) or isinstance(b, float):
        fa, fb = a, b
        if isinstance(a, float) and math.isnan(a):
            return None
        if isinstance(b, float) and math.isnan(b):
            return None
        # exact comparison between float and big ints
        if isinstance(a, float) and not isinstance(b, float):
            if math.isinf(a):
                return 1 if a > 0 else -1
            ia = int(a)
            if ia != b:
                return -1 if ia < b else 1
            frac = a - ia
            return 0 if frac == 0 else (1 if frac > 0 else -1)
        if isinstance(b, float) and not isinstance(a, float):
            r = _num_cmp(b, a)
            return None if r is None else -r
        return -1 if fa < fb else (1 if fa > fb else 0)
    return -1 if a < b else (1 if a > b else 0)


def cel_equal(a, b):
    """cel-go ``Equal`` with cross-type numeric equality; mismatched types -> False."""
    if is_num(a) and is_num(b):
        return _num_cmp(a, b) == 0
    if a is None or b is None:
        return a is None and b is None
    if isinstance(a, Hierarchy) or isinstance(b, Hierarchy):   # Hierarchy.Equal (hierarchy.go:222-238)
        return isinstance(a, Hierarchy) and isinstance(b, Hierarchy) and tuple(a) == tuple(b)
    if isinstance(a, bool) or isinstance(b, bool):
        return isinstance(a, bool) and isinstance(b, bool) and a == b
    if isinstance(a, str):
        return isinstance(b, str) and a == b
    if isinstance(a, bytes):
        return isinstance(b, bytes) and a == b
    if isinstance(a, list):
        if not isinstance(b, list) or len(a) != len(b):
            return False
        return all(cel_equal(x, y) for x, y in zip(a, b))
    if isinstance(a, dict):
        if not isinstance(b, dict) or len(a) != len(b):
            return False
        for k, v in a.items():
            found, bv = map_lookup(b, k)
            if not found or not cel_equal(v, bv):
                return False
        return True
    if isinstance(a, (Timestamp, Duration)):
        return a == b
    if isinstance(a, (Optional, NetIP, NetCIDR)):
        return a == b
    for x, y in ((a, b),):   # (cel-go asks the LEFT operand: lhs.Equal(rhs))
        if isinstance(x, SpiffeTrustDomain):   # spiffe.go SPIFFETrustDomain.Equal: another trust domain, or a string that parses to one
            if isinstance(y, SpiffeTrustDomain):
                return x.name == y.name
            if isinstance(y, str):
                try:
                    return _spiffe_td_from_string(y).name == x.name
                except CelError:
                    return False
            raise no_such_overload()
        if isinstance(x, SpiffeID):            # SPIFFEID.Equal: another id, or its string form
            if isinstance(y, SpiffeID):
                return x.text == y.text
            if isinstance(y, str):
                return x.text == y
            raise no_such_overload()
    if isinstance(a, SpiffeMatcher) or isinstance(b, SpiffeMatcher):
        return False
    return a is b


def map_lookup(m, k):
    """(found, value) with CEL's numeric key cross-type lookup."""
    if isinstance(k, str):
        return (True, m[k]) if k in m else (False, None)
    if isinstance(k, bool):
        # python: True == 1 - make sure bool keys only hit bool keys
        for mk, mv in m.items():
            if type(mk) == type(k) and mk == k:  # noqa: E721
                return True, mv
        return False, None
    if is_num(k):
        for mk, mv in m.items():
            if is_num(mk) and _num_cmp(mk, k) == 0:
                return True, mv
        return False, None
    return False, None


def cel_compare(a, b):
    if is_num(a) and is_num(b):
        r = _num_cmp(a, b)
        return r  # None for NaN -> all comparisons false
    if isinstance(a, bool) and isinstance(b, bool):
        return (a > b) - (a < b)
    if isinstance(a, str) and isinstance(b, str):
        ab, bb = a.encode("utf-8"), b.encode("utf-8")
        return (ab > bb) - (ab < bb)
    if isinstance(a, bytes) and isinstance(b, bytes):
        return (a > b) - (a < b)
    if isinstance(a, Timestamp) and isinstance(b, Timestamp):
        return (a.ns > b.ns) - (a.ns < b.ns)
    if isinstance(a, Duration) and isinstance(b, Duration):
        return (a.ns > b.ns) - (a.ns < b.ns)
    raise no_such_overload()


def _chk_int(v):
    if v < INT_MIN or v > INT_MAX:
        raise CelError("integer overflow")
    return v


def _chk_uint(v):
    if v < 0 or v > UINT_MAX:
        raise CelError("unsigned integer overflow")
    return UInt(v)


def _chk_dur(ns):
    if ns < INT_MIN or ns > INT_MAX:
        raise CelError("integer overflow")
    return Duration(ns)


_TS_MIN_S, _TS_MAX_S = -62135596800, 253402300799


def _chk_ts(ns):
    s = ns // 1_000_000_000
    if s < _TS_MIN_S or s > _TS_MAX_S:
        raise CelError("timestamp overflow")
    return Timestamp(ns)


# ---------------------------------------------------------------- time parsing

_RFC3339 = re.compile(
    r"^(\d{4})-(\d\d)-(\d\d)[Tt](\d\d):(\d\d):(\d\d)(\.\d+)?([Zz]|[+-]\d\d:\d\d)$")


def parse_timestamp(s: str) -> Timestamp:
    m = _RFC3339.match(s)
    if not m:
        raise CelError("invalid timestamp: %s" % s)
    y, mo, d, h, mi, sec = (int(m.group(i)) for i in range(1, 7))
    try:
        datetime.datetime(y, mo, d, h, mi, min(sec, 59))
    except ValueError:
        raise CelError("invalid timestamp: %s" % s)
    frac = m.group(7)
    ns = int((frac[1:] + "000000000")[:9]) if frac else 0
    epoch = calendar.timegm((y, mo, d, h, mi, sec, 0, 0, 0))
    tz = m.group(8)
    if tz not in ("Z", "z"):
        sign = 1 if tz[0] == "+" else -1
        off = sign * (int(tz[1:3]) * 3600 + int(tz[4:6]) * 60)
        epoch -= off
    return _chk_ts(epoch * 1_000_000_000 + ns)


_DUR_UNITS = {"ns": 1, "us": 1_000, "\u00b5s": 1_000, "\u03bcs": 1_000, "ms": 1_000_000,
              "s": 1_000_000_000, "m": 60_000_000_000, "h": 3_600_000_000_000}
_DUR_PART = re.compile(r"(\d+(?:\.\d*)?|\.\d+)(ns|us|\u00b5s|\u03bcs|ms|s|m|h)")


def parse_duration(s: str) -> Duration:
    """Go time.ParseDuration."""
    orig = s
    sign = 1
    if s[:1] in "+-":
        sign = -1 if s[0] == "-" else 1
        s = s[1:]
    if s == "0":
        return Duration(0)
    if not s:
        raise CelError("invalid duration: %s" % orig)
    pos, total = 0, 0
    while pos < len(s):
        m = _DUR_PART.match(s, pos)
        if not m:
            raise CelError("invalid duration: %s" % orig)
        num, unit = m.group(1), _DUR_UNITS[m.group(2)]
        if "." in num:
            ip, fp = num.split(".")
            total += int(ip or "0") * unit
            if fp:
                total += int(fp) * unit // (10 ** len(fp))
        else:
            total += int(num) * unit
        pos = m.end()
    return _chk_dur(sign * total)


def _tz_offset_seconds(ts: Timestamp, tz):
    if tz is None or tz == "" or tz == "UTC":
        return 0
    m = re.match(r"^([+-]?)(\d\d):(\d\d)$", tz)
    if m:
        sign = -1 if m.group(1) == "-" else 1
        return sign * (int(m.group(2)) * 3600 + int(m.group(3)) * 60)
    try:
        import zoneinfo
        z = zoneinfo.ZoneInfo(tz)
    except Exception:
        raise CelError("unknown time zone %s" % tz)
    dt = datetime.datetime.fromtimestamp(ts.ns // 1_000_000_000, tz=z)
    return int(dt.utcoffset().total_seconds())


def _civil(ts: Timestamp, tz):
    off = _tz_offset_seconds(ts, tz)
    secs = ts.ns // 1_000_000_000 + off
    nanos = ts.ns % 1_000_000_000
    st = datetime.datetime(1970, 1, 1) + datetime.timedelta(seconds=secs)
    return st, nanos


# ---------------------------------------------------------------- evaluator

class Env:
    """Activation for one evaluation (check.go:635-649)."""

    def __init__(self, idents: dict, now_ns: int):
        self.idents = idents
        self.now_ns = now_ns


def evaluate(src_or_ast, env: Env):
    ast = parse(src_or_ast) if isinstance(src_or_ast, str) else src_or_ast
    return _eval(ast, env, {})


def _eval(n, env, loc):
    k = n[0]
    if k == "lit":
        if n[1] == "uint":
            return UInt(n[2])
        return n[2]
    if k == "ident":
        name = n[1]
        if name in loc:
            return loc[name]
        if name in env.idents:
            v = env.idents[name]
            if callable(v):
                v = v()
            return v
        raise CelError("undeclared reference to '%s'" % name)
    if k == "select":
        # namespaced constants/functions do not reach here (handled in 'call')
        return _select(_eval(n[1], env, loc), n[2])
    if k == "has":
        return _has(_eval(n[1], env, loc), n[2])
    if k == "index":
        return _index(_eval(n[1], env, loc), _eval(n[2], env, loc))
    if k == "and":
        return _logic(n, env, loc, False)
    if k == "or":
        return _logic(n, env, loc, True)
    if k == "not":
        v = _eval(n[1], env, loc)
        if not isinstance(v, bool):
            raise no_such_overload()
        return not v
    if k == "neg":
        v = _eval(n[1], env, loc)
        if is_int(v):
            return _chk_int(-v)
        if is_double(v):
            return -v
        raise no_such_overload()
    if k == "tern":
        c = _eval(n[1], env, loc)
        if not isinstance(c, bool):
            raise no_such_overload()
        return _eval(n[2] if c else n[3], env, loc)
    if k == "bin":
        return _binop(n[1], _eval(n[2], env, loc), _eval(n[3], env, loc))
    if k == "list":
        return [_eval(e, env, loc) for e in n[1]]
    if k == "map":
        out = {}
        for ke, ve in n[1]:
            kv = _eval(ke, env, loc)
            if not (isinstance(kv, (bool, str)) or is_num(kv)):
                raise CelError("unsupported key type")
            out[kv] = _eval(ve, env, loc)
        return out
    if k == "comp":
        return _comprehension(n, env, loc)
    if k == "bind":
        inner = dict(loc)
        inner[n[1]] = _eval(n[2], env, loc)
        return _eval(n[3], env, inner)
    if k == "call":
        return _call(n, env, loc)
    raise CelError("unsupported node %s" % k)


def _logic(n, env, loc, is_or):
    """&& / || with CEL's commutative error absorption."""
    err = None
    for side in (n[1], n[2]):
        try:
            v = _eval(side, env, loc)
        except CelError as e:
            err = err or e
            continue
        if not isinstance(v, bool):
            err = err or no_such_overload()
            continue
        if v == is_or:
            return is_or
    if err is not None:
        raise err
    return not is_or


def _select(v, field):
    if isinstance(v, dict):
        found, out = map_lookup(v, field)
        if not found:
            raise CelError("no such key: %s" % field)
        return out
    if isinstance(v, Message):
        name = v.resolve(field)
        if name is None:
            raise CelError("no such field '%s'" % field)
        out = v.fields[name]
        return out() if callable(out) else out
    if isinstance(v, Variables):
        if field not in v.values:
            raise CelError("undefined field '%s'" % field)
        return v.values[field]
    raise CelError("no such overload")  # select on a non-container (null, scalar, list)


def _has(v, field):
    if isinstance(v, dict):
        return map_lookup(v, field)[0]
    if isinstance(v, Message):
        name = v.resolve(field)
        if name is None:
            raise CelError("no such field '%s'" % field)
        out = v.fields[name]
        if callable(out):
            out = out()
        # proto3 presence: scalars/repeated/maps are "set" when non-default
        if isinstance(out, Message):
            return True
        return out not in ("", None, [], {}, 0, False)
    if isinstance(v, Variables):
        return field in v.values
    raise CelError("no such overload")


def _index(v, i):
    if isinstance(v, Hierarchy):   # Hierarchy.Get (hierarchy.go:240-251)
        if isinstance(i, bool) or not isinstance(i, int) or is_uint(i):
            raise CelError("unsupported index type '%s'" % type_name(i))
        if i < 0 or i >= len(v):
            raise CelError("index out of range")
        return v[i]
    if isinstance(v, list):
        if isinstance(i, bool) or not is_num(i):
            raise no_such_overload()
        if isinstance(i, float):
            if i != int(i):
                raise CelError("unsupported index value")
            i = int(i)
        if i < 0 or i >= len(v):
            raise CelError("index out of bounds: %d" % i)
        return v[i]
    if isinstance(v, dict):
        found, out = map_lookup(v, i)
        if not found:
            raise CelError("no such key: %s" % (i if isinstance(i, str) else format_value(i)))
        return out
    if isinstance(v, (Message, Variables)) and isinstance(i, str):
        return _select(v, i)
    raise no_such_overload()


def _binop(op, a, b):
    if op == "==":
        return cel_equal(a, b)
    if op == "!=":
        return not cel_equal(a, b)
    if op in ("<", "<=", ">", ">="):
        r = cel_compare(a, b)
        if r is None:
            return False
        return {"<": r < 0, "<=": r <= 0, ">": r > 0, ">=": r >= 0}[op]
    if op == "in":
        if isinstance(b, list):
            return any(cel_equal(a, x) for x in b)
        if isinstance(b, dict):
            return map_lookup(b, a)[0]
        raise no_such_overload()
    if op == "+":
        if is_int(a) and is_int(b):
            return _chk_int(a + b)
        if is_uint(a) and is_uint(b):
            return _chk_uint(int(a) + int(b))
        if is_double(a) and is_double(b):
            return a + b
        if isinstance(a, str) and isinstance(b, str):
            return a + b
        if isinstance(a, bytes) and isinstance(b, bytes):
            return a + b
        if isinstance(a, list) and isinstance(b, list):
            return a + b
        if isinstance(a, Timestamp) and isinstance(b, Duration):
            return _chk_ts(a.ns + b.ns)
        if isinstance(a, Duration) and isinstance(b, Timestamp):
            return _chk_ts(a.ns + b.ns)
        if isinstance(a, Duration) and isinstance(b, Duration):
            return _chk_dur(a.ns + b.ns)
        raise no_such_overload()
    if op == "-":
        if is_int(a) and is_int(b):
            return _chk_int(a - b)
        if is_uint(a) and is_uint(b):
            return _chk_uint(int(a) - int(b))
        if is_double(a) and is_double(b):
            return a - b
        if isinstance(a, Timestamp) and isinstance(b, Timestamp):
            return _chk_dur(a.ns - b.ns)
        if isinstance(a, Timestamp) and isinstance(b, Duration):
            return _chk_ts(a.ns - b.ns)
        if isinstance(a, Duration) and isinstance(b, Duration):
            return _chk_dur(a.ns - b.ns)
        raise no_such_overload()
    if op == "*":
        if is_int(a) and is_int(b):
            return _chk_int(a * b)
        if is_uint(a) and is_uint(b):
            return _chk_uint(int(a) * int(b))
        if is_double(a) and is_double(b):
            return a * b
        raise no_such_overload()
    if op == "/":
        if is_int(a) and is_int(b):
            if b == 0:
                raise CelError("division by zero")
            if a == INT_MIN and b == -1:
                raise CelError("integer overflow")
            q = abs(a) // abs(b)
            return q if (a < 0) == (b < 0) else -q
        if is_uint(a) and is_uint(b):
            if b == 0:
                raise CelError("division by zero")
            return UInt(int(a) // int(b))
        if is_double(a) and is_double(b):
            if b == 0:
                if a == 0 or math.isnan(a):
                    return math.nan
                return math.copysign(math.inf, a) * math.copysign(1.0, b)
            return a / b
        raise no_such_overload()
    if op == "%":
        if is_int(a) and is_int(b):
            if b == 0:
                raise CelError("modulus by zero")
            if a == INT_MIN and b == -1:
                raise CelError("integer overflow")
            r = abs(a) % abs(b)
            return -r if a < 0 else r
        if is_uint(a) and is_uint(b):
            if b == 0:
                raise CelError("modulus by zero")
            return UInt(int(a) % int(b))
        raise no_such_overload()
    raise CelError("unknown operator %s" % op)


def _iter_range(target):
    """(key-ish, value) pairs for two-variable forms; element for single-variable."""
    if isinstance(target, list):
        return [(i, v) for i, v in enumerate(target)], "list"
    if isinstance(target, dict):
        return list(target.items()), "map"
    raise no_such_overload()


def _comprehension(n, env, loc):
    _, kind, target_ast, vars_, args = n
    target = _eval(target_ast, env, loc)
    pairs, shape = _iter_range(target)
    two = len(vars_) == 2

    def bind(pair):
        inner = dict(loc)
        if two:
            inner[vars_[0]], inner[vars_[1]] = pair
        else:
            inner[vars_[0]] = pair[1] if shape == "list" else pair[0]
        return inner

    def pred(ast, inner):
        v = _eval(ast, env, inner)
        if not isinstance(v, bool):
            raise no_such_overload()
        return v

    if kind in ("all", "exists"):
        want = kind == "exists"
        err = None
        for p in pairs:
            try:
                v = pred(args[0], bind(p))
            except CelError as e:
                err = err or e
                continue
            if v == want:
                return want
        if err is not None:
            raise err
        return not want
    if kind in ("exists_one", "existsOne"):
        cnt = 0
        for p in pairs:
            if pred(args[0], bind(p)):
                cnt += 1
        return cnt == 1
    if kind == "filter":
        out = []
        for p in pairs:
            inner = bind(p)
            if pred(args[0], inner):
                out.append(inner[vars_[0]])
        return out
    if kind == "map":
        out = []
        for p in pairs:
            inner = bind(p)
            if len(args) == 2 and not pred(args[0], inner):
                continue
            out.append(_eval(args[-1], env, inner))
        return out
    if kind == "transformList":
        out = []
        for p in pairs:
            inner = bind(p)
            if len(args) == 2 and not pred(args[0], inner):
                continue
            out.append(_eval(args[-1], env, inner))
        return out
    if kind in ("transformMap", "transformMapEntry"):
        out = {}
        for p in pairs:
            inner = bind(p)
            if len(args) == 2 and not pred(args[0], inner):
                continue
            v = _eval(args[-1], env, inner)
            if kind == "transformMap":
                out[p[0]] = v
            else:
                if not isinstance(v, dict):
                    raise no_such_overload()
                for kk, vv in v.items():
                    if map_lookup(out, kk)[0]:
                        raise CelError("insert failed: key %s already exists" % format_value(kk))
                    out[kk] = vv
        return out
    if kind == "sortBy":
        keyed = [(_eval(args[0], env, bind(p)), bind(p)[vars_[0]]) for p in pairs]
        return [v for _, v in _sorted(keyed, key=lambda kv: kv[0])]
    raise CelError("unsupported macro %s" % kind)


class _Key:
    def __init__(self, v):
        self.v = v

    def __lt__(self, o):
        r = cel_compare(self.v, o.v)
        return r is not None and r < 0


def _sorted(items, key=lambda x: x):
    return sorted(items, key=lambda x: _Key(key(x)))


# ---------------------------------------------------------------- functions

def format_value(v):
    if v is None:
        return "null"
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, float):
        if v == int(v) and abs(v) < 1e21:
            return str(int(v))
        return repr(v)
    if isinstance(v, (int, str)):
        return str(v)
    if isinstance(v, bytes):
        return v.decode("utf-8", "replace")
    if isinstance(v, list):
        return "[" + ", ".join(_format_nested(x) for x in v) + "]"
    if isinstance(v, dict):
        return "{" + ", ".join("%s: %s" % (_format_nested(k), _format_nested(x)) for k, x in sorted(
            v.items(), key=lambda kv: str(kv[0]))) + "}"
    if isinstance(v, Duration):
        return "%ss" % format_value(v.ns / 1e9)
    if isinstance(v, Timestamp):
        st, nanos = _civil(v, None)
        return st.strftime("%Y-%m-%dT%H:%M:%S") + (".%09d" % nanos).rstrip("0").rstrip(".") + "Z"
    return str(v)


def _format_nested(v):
    if isinstance(v, str):
        return '"%s"' % v
    return format_value(v)


def _format(fmt, args):
    """ext.Strings format (subset: %s %d %f %e %x %X %o %b %%)."""
    out = []
    i, ai = 0, 0
    while i < len(fmt):
        c = fmt[i]
        if c != "%":
            out.append(c)
            i += 1
            continue
        i += 1
        if i >= len(fmt):
            raise CelError("unexpected end of format string")
        prec = None
        if fmt[i] == ".":
            j = i + 1
            while j < len(fmt) and fmt[j].isdigit():
                j += 1
            prec = int(fmt[i + 1:j] or "0")
            i = j
        spec = fmt[i]
        i += 1
        if spec == "%":
            out.append("%")
            continue
        if ai >= len(args):
            raise CelError("index %d out of range" % ai)
        a = args[ai]
        ai += 1
        if spec == "s":
            out.append(format_value(a))
        elif spec == "d":
            if isinstance(a, bool) or not is_num(a) or isinstance(a, float) and a != int(a):
                raise CelError("error during formatting: decimal clause can only be used on integers")
            out.append(str(int(a)))
        elif spec == "f":
            out.append("%.*f" % (6 if prec is None else prec, float(a)))
        elif spec == "e":
            out.append("%.*e" % (6 if prec is None else prec, float(a)))
        elif spec in "xX":
            if isinstance(a, str):
                s = a.encode().hex()
            elif isinstance(a, bytes):
                s = a.hex()
            else:
                s = "%x" % int(a)
            out.append(s.upper() if spec == "X" else s)
        elif spec == "o":
            out.append("%o" % int(a))
        elif spec == "b":
            out.append(("1" if a else "0") if isinstance(a, bool) else bin(int(a))[2:])
        else:
            raise CelError("unrecognized formatting clause %s" % spec)
    return "".join(out)


def _re2_to_python(p):
    """The RE2 dialect (Go regexp, what cel-go's `matches` compiles) written for Python's `re` where the two read the
    same text differently: `$` is the end of the text only (Python also stops before a trailing newline), `\\z` is
    Python's `\\Z`, `\\s` is [\\t\\n\\f\\r ] (Python adds \\v), a `{` that does not open `{n}`, `{n,}` or `{n,m}` is a
    literal.  With re.ASCII for \\d \\w (RE2's are ASCII-only)."""
    out, i, in_class = [], 0, False
    while i < len(p):
        c = p[i]
        if c == "\\" and i + 1 < len(p):
            nx = p[i + 1]
            if nx == "z" and not in_class:
                out.append("\\Z")
            elif nx == "s":
                out.append("\\t\\n\\f\\r " if in_class else "[\\t\\n\\f\\r ]")
            elif nx == "S" and not in_class:
                out.append("[^\\t\\n\\f\\r ]")
            else:
                out.append(c + nx)
            i += 2
            continue
        if c == "[" and not in_class:
            in_class = True
            out.append(c)
            i += 1
            if i < len(p) and p[i] == "^":
                out.append("^"); i += 1
            if i < len(p) and p[i] == "]":
                out.append("\\]"); i += 1
            continue
        if c == "]" and in_class:
            in_class = False
        if c == "$" and not in_class:
            out.append("\\Z")
        elif c == "{" and not in_class and not re.match(r"\{\d+(,\d*)?\}", p[i:]):
            out.append("\\{")
        else:
            out.append(c)
        i += 1
    return "".join(out)


def _re(pattern):
    try:
        return re.compile(_re2_to_python(pattern), re.ASCII)
    except re.error as e:
        raise CelError("invalid regex: %s" % e)


def _to_set_list(v):
    if not isinstance(v, list):
        raise no_such_overload()
    return v


def _contains_all(a, b):
    return all(any(cel_equal(x, y) for y in a) for x in b)


def _ip_in_range(ip, cidr):
    """cerbos_lib.go:513-526: net.ParseIP, net.ParseCIDR, IPNet.Contains.  Where Python's ipaddress reads more than Go
    does it is narrowed to Go: a CIDR is "<address>/<decimal digits>" (no netmask form, no bare address), an address
    carries no zone, and an IPv4-mapped IPv6 address is the IPv4 address (IP.To4 in Contains)."""
    addr_s, sep, bits_s = cidr.partition("/")
    if not sep or not bits_s.isascii() or not bits_s.isdigit() or "%" in cidr or "%" in ip:
        raise CelError("invalid CIDR address: %s" % cidr)
    try:
        net_addr = ipaddress.ip_address(addr_s)
        addr = ipaddress.ip_address(ip)
    except ValueError as e:
        raise CelError(str(e))
    bits = int(bits_s)
    if bits > net_addr.max_prefixlen:
        raise CelError("invalid CIDR address: %s" % cidr)
    if addr.version == 6 and addr.ipv4_mapped is not None:
        addr = addr.ipv4_mapped
    width = net_addr.max_prefixlen
    net_int = (int(net_addr) >> (width - bits)) << (width - bits)        # ParseCIDR keeps ip.Mask(m)
    if width == 128 and (net_int >> 32) == 0xFFFF:
        # networkNumberAndMask (net/ip.go): an IPv4-mapped network number is its IPv4 form, the mask its last four bytes
        net_int, width, bits = net_int & 0xFFFFFFFF, 32, max(bits - 96, 0)
    if addr.max_prefixlen != width:
        return False
    return (int(addr) >> (width - bits)) == (net_int >> (width - bits))


def _codepoints(s):
    return s  # python str is already a sequence of code points


def _call(n, env, loc):
    _, name, target_ast, arg_asts = n
    # namespaced functions: sets.contains(...), lists.range(...), math.greatest(...), base64.encode(...)
    if target_ast is not None and target_ast[0] == "ident" and target_ast[1] not in loc \
            and target_ast[1] not in env.idents and target_ast[1] in _NAMESPACES:
        args = [_eval(a, env, loc) for a in arg_asts]
        fn = _NS_FUNCS.get((target_ast[1], name))
        if fn is None:
            raise CelError("unsupported function %s.%s" % (target_ast[1], name))
        return fn(env, *args)
    if target_ast is None:
        args = [_eval(a, env, loc) for a in arg_asts]
        fn = _GLOBAL_FUNCS.get(name)
        if fn is None:
            raise CelError("unsupported function %s" % name)
        return fn(env, *args)
    target = _eval(target_ast, env, loc)
    args = [_eval(a, env, loc) for a in arg_asts]
    fn = _METHODS.get(name)
    if fn is None:
        raise CelError("unsupported method %s" % name)
    return fn(env, target, *args)


def _f_size(env, v):
    if isinstance(v, (str, list, dict, bytes, Hierarchy)):
        return len(v)
    raise no_such_overload()


# ---- cerbos.lib.hierarchy (internal/conditions/types/hierarchy.go) --------------------------------------------
class Hierarchy(tuple):
    """hierarchy.go:160: the segments of a delimited path."""


def _f_hierarchy(env, v, delim=None):
    # unaryHierarchyFnImpl / binaryHierarchyFnImpl (hierarchy.go:130-158)
    if delim is not None:
        return Hierarchy(_need(v, str).split(_need(delim, str)))
    if isinstance(v, Hierarchy):
        return v
    if isinstance(v, str):
        return Hierarchy(v.split("."))
    if isinstance(v, list):
        if not all(isinstance(x, str) for x in v):
            raise CelError("failed to convert list to string slice")
        return Hierarchy(v)
    raise no_such_overload()


def _hier(v):   # toHierarchy (hierarchy.go:400-407)
    if not isinstance(v, Hierarchy):
        raise no_such_overload()
    return v


def _h_ancestor_of(h, child):          # hierarchy.go:259-276
    return len(child) > len(h) and child[:len(h)] == h


def _h_immediate_parent_of(h, child):  # hierarchy.go:326-343
    return len(child) == len(h) + 1 and child[:len(h)] == h


def _h_sibling_of(h, other):           # hierarchy.go:345-362
    return len(other) == len(h) and h[:len(h) - 1] == other[:len(h) - 1]


def _h_overlaps(h, other):             # hierarchy.go:364-385
    short, long_ = (other, h) if len(other) < len(h) else (h, other)
    return long_[:len(short)] == short


def _h_common_ancestors(h, other):     # hierarchy.go:278-306
    short, long_ = (other, h) if len(other) < len(h) else (h, other)
    if len(long_) == len(short):
        long_, short = long_[:-1], short[:-1]
    out = []
    for a, b in zip(short, long_):
        if a != b:
            break
        out.append(a)
    return Hierarchy(out)


def _f_int(env, v):
    if isinstance(v, bool):
        raise no_such_overload()
    if is_uint(v):
        return _chk_int(int(v))
    if isinstance(v, int):
        return v
    if isinstance(v, float):
        if math.isnan(v) or math.isinf(v) or v >= 9.223372036854775807e18 or v <= -9.223372036854775808e18:
            raise CelError("integer overflow")
        return int(v)
    if isinstance(v, str):
        try:
            return _chk_int(int(v, 10))
        except ValueError:
            raise CelError("cannot parse %r as int" % v)
    if isinstance(v, Timestamp):
        return v.ns // 1_000_000_000
    if isinstance(v, Duration):
        return v.ns
    raise no_such_overload()


def _f_uint(env, v):
    if isinstance(v, bool):
        raise no_such_overload()
    if isinstance(v, int):
        return _chk_uint(int(v))
    if isinstance(v, float):
        if math.isnan(v) or v < 0 or v >= 1.8446744073709552e19:
            raise CelError("unsigned integer overflow")
        return UInt(int(v))
    if isinstance(v, str):
        try:
            return _chk_uint(int(v, 10))
        except ValueError:
            raise CelError("cannot parse %r as uint" % v)
    raise no_such_overload()


def _f_double(env, v):
    if isinstance(v, bool):
        raise no_such_overload()
    if isinstance(v, (int, float)):
        return float(v)
    if isinstance(v, str):
        try:
            return float(v)
        except ValueError:
            raise CelError("cannot parse %r as double" % v)
    raise no_such_overload()


def _f_string(env, v):
    if isinstance(v, str):
        return v
    if isinstance(v, bytes):
        try:
            return v.decode("utf-8")
        except UnicodeDecodeError:
            raise CelError("invalid UTF-8")
    if isinstance(v, bool) or is_num(v) or isinstance(v, (Timestamp, Duration)):
        if isinstance(v, float) and v != int(v):
            return repr(v)
        return format_value(v)
    raise no_such_overload()


def _f_bool(env, v):
    if isinstance(v, bool):
        return v
    if isinstance(v, str):
        if v in ("1", "t", "true", "TRUE", "True"):
            return True
        if v in ("0", "f", "false", "FALSE", "False"):
            return False
        raise CelError("cannot parse %r as bool" % v)
    raise no_such_overload()


def _f_bytes(env, v):
    if isinstance(v, bytes):
        return v
    if isinstance(v, str):
        return v.encode("utf-8")
    raise no_such_overload()


def _f_timestamp(env, v):
    if isinstance(v, Timestamp):
        return v
    if isinstance(v, str):
        return parse_timestamp(v)
    if is_int(v):
        return _chk_ts(v * 1_000_000_000)
    raise no_such_overload()


def _f_duration(env, v):
    if isinstance(v, Duration):
        return v
    if isinstance(v, str):
        return parse_duration(v)
    if is_int(v):
        return Duration(v)
    raise no_such_overload()


def _f_now(env):
    return Timestamp(env.now_ns)


def _f_time_since(env, ts):
    if not isinstance(ts, Timestamp):
        raise no_such_overload()
    return _chk_dur(env.now_ns - ts.ns)


def _need(v, *types):
    if isinstance(v, bool) and bool not in types:
        raise no_such_overload()
    if not isinstance(v, types):
        raise no_such_overload()
    return v


def _m_matches(env, s, p):
    _need(s, str)
    _need(p, str)
    return _re(p).search(s) is not None


def _ts_getter(fn):
    def g(env, v, tz=None):
        if isinstance(v, Timestamp):
            st, nanos = _civil(v, tz)
            return fn(st, nanos)
        raise no_such_overload()
    return g


def _dur_or_ts(dur_fn, ts_fn):
    def g(env, v, tz=None):
        if isinstance(v, Duration):
            return dur_fn(v.ns)
        if isinstance(v, Timestamp):
            st, nanos = _civil(v, tz)
            return ts_fn(st, nanos)
        raise no_such_overload()
    return g


def _trunc_div(a, b):
    q = abs(a) // b
    return q if a >= 0 else -q


def _m_substring(env, s, a, b=None):
    _need(s, str)
    cps = _codepoints(s)
    if b is None:
        b = len(cps)
    if not is_int(a) or not is_int(b):
        raise no_such_overload()
    if a < 0 or a > len(cps) or b < 0 or b > len(cps):
        raise CelError("index out of range: %d" % (a if (a < 0 or a > len(cps)) else b))
    if a > b:
        raise CelError("invalid substring range. start: %d, end: %d" % (a, b))
    return cps[a:b]


def _m_char_at(env, s, i):
    _need(s, str)
    if not is_int(i) or i < 0 or i > len(s):
        raise CelError("index out of range: %s" % i)
    return s[i] if i < len(s) else ""


def _m_index_of(env, s, sub, start=0):
    _need(s, str)
    _need(sub, str)
    if not is_int(start):
        raise no_such_overload()
    if start < 0 or start > len(s):
        raise CelError("index out of range: %d" % start)
    return s.find(sub, start)


def _m_last_index_of(env, s, sub, start=None):
    _need(s, str)
    _need(sub, str)
    if start is None:
        return s.rfind(sub)
    if not is_int(start):
        raise no_such_overload()
    if start < 0 or start > len(s):
        raise CelError("index out of range: %d" % start)
    return s.rfind(sub, 0, start + len(sub))


def _m_replace(env, s, old, new, limit=-1):
    _need(s, str), _need(old, str), _need(new, str)
    if not is_int(limit):
        raise no_such_overload()
    return s.replace(old, new) if limit < 0 else s.replace(old, new, limit)


def _m_split(env, s, sep, limit=-1):
    _need(s, str), _need(sep, str)
    if not is_int(limit):
        raise no_such_overload()
    if limit == 0:
        return []
    if limit == 1:
        return [s]
    if sep == "":
        parts = list(s)
        if limit > 0 and len(parts) > limit:
            parts = parts[:limit - 1] + ["".join(parts[limit - 1:])]
        return parts
    return s.split(sep) if limit < 0 else s.split(sep, limit - 1)


def _m_join(env, lst, sep=""):
    _need(lst, list), _need(sep, str)
    if not all(isinstance(x, str) for x in lst):
        raise no_such_overload()
    return sep.join(lst)


def _m_sort(env, lst):
    _need(lst, list)
    return _sorted(lst)


def _m_distinct(env, lst):
    _need(lst, list)
    out = []
    for x in lst:
        if not any(cel_equal(x, y) for y in out):
            out.append(x)
    return out


def _m_flatten(env, lst, depth=1):
    _need(lst, list)
    if depth < 0:
        raise CelError("level must be non-negative")
    out = []
    for x in lst:
        if isinstance(x, list) and depth > 0:
            out.extend(_m_flatten(env, x, depth - 1))
        else:
            out.append(x)
    return out


def _m_slice(env, lst, a, b):
    _need(lst, list)
    if a < 0 or b < 0:
        raise CelError("cannot slice(%d, %d), negative indexes not supported" % (a, b))
    if a > b:
        raise CelError("cannot slice(%d, %d), start index must be less than or equal to end index" % (a, b))
    if b > len(lst):
        raise CelError("cannot slice(%d, %d), list is length %d" % (a, b, len(lst)))
    return lst[a:b]


def _intersect(env, a, b):
    a, b = _to_set_list(a), _to_set_list(b)
    if len(a) > len(b):     # cerbos_lib.go:434-437: the shorter list is the one iterated
        a, b = b, a
    return [x for x in a if any(cel_equal(x, y) for y in b)]


def _except(env, a, b):
    a, b = _to_set_list(a), _to_set_list(b)
    return [x for x in a if not any(cel_equal(x, y) for y in b)]


def _has_intersection(env, a, b):
    a, b = _to_set_list(a), _to_set_list(b)
    return any(any(cel_equal(x, y) for y in b) for x in a)


def _is_subset(env, a, b):
    """a.isSubset(b): every element of a is in b (cerbos_lib.go setsContains)."""
    a, b = _to_set_list(a), _to_set_list(b)
    return _contains_all(b, a)


def _m_in_ip_range(env, ip, cidr):
    _need(ip, str)
    _need(cidr, str)
    return _ip_in_range(ip, cidr)


# cel-go ext.Math() beyond greatest / least (ext/math.go; cerbos enables the library at its latest version, conditions/cel.go:78;
# the functions and one example each: docs/modules/policies/pages/conditions.adoc:456-472).  Bit operations take two ints or two
# uints; shifts take the value and an int offset: a negative offset is an error, an offset of 64 or more gives 0, a right shift
# of an int fills with zeros (the value is shifted as its 64-bit pattern).
def _wrap_int(x):
    x &= 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >> 63 else x


def _math_bit(op):
    def fn(env, a, b):
        if is_uint(a) and is_uint(b):
            return UInt(op(int(a), int(b)) & 0xFFFFFFFFFFFFFFFF)
        if is_int(a) and is_int(b):
            return _wrap_int(op(int(a), int(b)))
        raise no_such_overload()
    return fn


def _math_bit_not(env, a):
    if is_uint(a):
        return UInt(~int(a) & 0xFFFFFFFFFFFFFFFF)
    if is_int(a):
        return ~int(a)
    raise no_such_overload()


def _math_shift(left):
    def fn(env, a, n):
        if not (is_int(a) or is_uint(a)) or not is_int(n):
            raise no_such_overload()
        if n < 0:
            raise CelError("math.bitShift%s() negative offset: %d" % ("Left" if left else "Right", n))
        if n >= 64:
            return UInt(0) if is_uint(a) else 0
        bits = int(a) & 0xFFFFFFFFFFFFFFFF
        out = (bits << n) & 0xFFFFFFFFFFFFFFFF if left else bits >> n
        return UInt(out) if is_uint(a) else _wrap_int(out)
    return fn


def _math_sign(env, v):
    if is_uint(v):
        return UInt(1 if v > 0 else 0)
    if is_int(v):
        return (v > 0) - (v < 0)
    if is_double(v):
        return v if math.isnan(v) else float((v > 0) - (v < 0))
    raise no_such_overload()


def _math_sqrt(env, v):
    if not is_num(v):
        raise no_such_overload()
    f = float(v)
    return math.nan if math.isnan(f) or f < 0 else math.sqrt(f)


def _go_round(d):
    """Go's math.Round: half away from zero, exact (floor(|d| + 0.5) is not: 0.49999999999999994 + 0.5 rounds up to 1.0,
    and above 2^52 adding 0.5 moves to the next even integer); the sign of zero is kept."""
    t = float(math.trunc(d))
    if abs(d - t) >= 0.5:
        t += math.copysign(1.0, d)
    return math.copysign(t, d)


def _dbl_round(fn, v):
    """Go's math.Ceil / Floor / Trunc / Round: NaN and the infinities come back unchanged."""
    _need(v, float)
    return v if math.isnan(v) or math.isinf(v) else float(fn(v))


def _math_extreme(pick):
    def f(env, *args):
        if len(args) == 1 and isinstance(args[0], list):
            args = args[0]
        if not args or not all(is_num(a) for a in args):
            raise no_such_overload()
        best = args[0]
        for a in args[1:]:
            r = _num_cmp(a, best)
            if r is not None and ((r > 0) if pick > 0 else (r < 0)):
                best = a
        return best
    return f


# ---- cel-go ext.Network (v0.30.0 ext/network.go, not vendored: restated from its documentation - the k8s IP / CIDR library
# it ports): addresses are netip.Addr values parsed with netip.ParseAddr, zones and IPv4-mapped IPv6 forms refused;
# networks are netip.Prefix values parsed with netip.ParsePrefix.  Pinned by cel_eval/network.yaml.
class NetIP:
    def __init__(self, addr):
        self.addr = addr

    def __eq__(self, o):
        return isinstance(o, NetIP) and self.addr == o.addr

    def __hash__(self):
        return hash(self.addr)


class NetCIDR:
    def __init__(self, addr, bits):
        self.addr, self.bits = addr, bits

    def __eq__(self, o):
        return isinstance(o, NetCIDR) and (self.addr, self.bits) == (o.addr, o.bits)

    def __hash__(self):
        return hash((self.addr, self.bits))

    def network(self):
        return ipaddress.ip_network("%s/%d" % (self.addr, self.bits), strict=False)


def _netip_parse(text):
    """netip.ParseAddr, then the two refusals of the library."""
    if not isinstance(text, str):
        raise no_such_overload()
    if "%" in text:
        raise CelError("IP address with zone value is not allowed")
    if ":" not in text:
        parts = text.split(".")
        if len(parts) != 4 or not all(p.isascii() and p.isdigit() and (p == "0" or p[0] != "0") and int(p) < 256 and len(p) <= 3 for p in parts):
            raise CelError("IP Address %r parse error during conversion from string" % text)
        return ipaddress.IPv4Address(text)
    try:
        a = ipaddress.IPv6Address(text)
    except ValueError:
        raise CelError("IP Address %r parse error during conversion from string" % text)
    if a.ipv4_mapped is not None:
        raise CelError("IPv4-mapped IPv6 address is not allowed")
    return a


def _netip_prefix(text):
    if not isinstance(text, str):
        raise no_such_overload()
    addr, sep, bits = text.rpartition("/")
    if not sep or not bits.isascii() or not bits.isdigit() or (len(bits) > 1 and bits[0] == "0"):
        raise CelError("network address parse error during conversion from string")
    try:
        a = _netip_parse(addr)
    except CelError:
        raise CelError("network address parse error during conversion from string")
    if int(bits) > a.max_prefixlen:
        raise CelError("network address parse error during conversion from string")
    return NetCIDR(a, int(bits))


def _try(fn, *a):
    try:
        fn(*a)
        return True
    except CelError as e:
        if "no such overload" in str(e):
            raise
        return False


def _n_is_ip(env, s, version=None):
    _need(s, str)
    if version is not None and not is_int(version):
        raise no_such_overload()
    try:
        a = _netip_parse(s)
    except CelError:
        return False
    return version is None or a.version == version


def _n_ip(env, v):
    if isinstance(v, NetCIDR):
        return NetIP(v.addr)
    return NetIP(_netip_parse(v))


def _n_global_unicast(a):
    # netip.Addr.IsGlobalUnicast: not the zero / unspecified / IPv4 broadcast / loopback / multicast / link-local unicast address
    if a.version == 4 and a == ipaddress.IPv4Address("255.255.255.255"):
        return False
    return not (a.is_unspecified or a.is_loopback or a.is_multicast or a.is_link_local)


def _n_ll_multicast(a):
    if a.version == 4:
        return a in ipaddress.ip_network("224.0.0.0/24")
    return a.packed[0] == 0xFF and (a.packed[1] & 0x0F) == 0x02


def _nip(v):
    if not isinstance(v, NetIP):
        raise no_such_overload()
    return v.addr


def _ncidr(v):
    if not isinstance(v, NetCIDR):
        raise no_such_overload()
    return v


def _n_contains_ip(env, c, ip):
    a = ip.addr if isinstance(ip, NetIP) else _netip_parse(ip)
    return a.version == _ncidr(c).addr.version and a in c.network()


def _n_contains_cidr(env, c, other):
    o = other if isinstance(other, NetCIDR) else _netip_prefix(other)
    c = _ncidr(c)
    return o.addr.version == c.addr.version and c.bits <= o.bits and o.addr in c.network()


# ---- cel-go ext.Regex / optional values (what the reference's string_funcs.yaml exercises)
class Optional:
    def __init__(self, *value):
        self.has = bool(value)
        self.value = value[0] if value else None

    def __eq__(self, o):
        return isinstance(o, Optional) and self.has == o.has and (not self.has or cel_equal(self.value, o.value))

    def __hash__(self):
        return hash(self.has)


def go_matches(rx, s):
    """The matches Go's regexp yields for its *All functions over `s` (regexp.go allMatches): Python's, minus an EMPTY match
    that begins where the previous match ended ("if 'All' is present ... empty matches abutting a preceding match are ignored")
    - `x*` over "abxd" matches at 0, 1, 2..3 and 4, not at 3."""
    prev_end = -1
    for m in rx.finditer(s):
        if m.start() == m.end() == prev_end:
            continue
        prev_end = m.end()
        yield m


def go_replace_all(rx, s, template, limit):
    """regexp.ReplaceAllString with Go's match list; `template` in Python's expand syntax; limit < 0 = all."""
    out, at, n = [], 0, 0
    for m in go_matches(rx, s):
        if 0 <= limit <= n:
            break
        out.append(s[at:m.start()])
        out.append(m.expand(template))
        at = m.end()
        n += 1
    out.append(s[at:])
    return "".join(out)


def _rx_replace(env, s, pattern, repl, limit=-1):
    _need(s, str); _need(repl, str)
    if not is_int(limit):
        raise no_such_overload()
    rx = _re(_need(pattern, str))
    # the replacement names groups \0 .. \9; any other backslash sequence is an error (ext/regex.go)
    out, i = [], 0
    while i < len(repl):
        if repl[i] == "\\":
            if i + 1 >= len(repl) or not repl[i + 1].isdigit():
                raise CelError("invalid replacement string")
            if int(repl[i + 1]) > rx.groups:
                raise CelError("replacement string references a group that does not exist")
            out.append("\\g<%s>" % repl[i + 1]); i += 2
        else:
            out.append(repl[i].replace("\\", "\\\\")); i += 1
    if limit == 0:
        return s
    return go_replace_all(rx, s, "".join(out), limit)


def _rx_extract(env, s, pattern):
    rx = _re(_need(pattern, str))
    if rx.groups > 1:
        raise CelError("regular expression has more than one capturing group")
    m = rx.search(_need(s, str))
    if m is None:
        return Optional()
    got = m.group(rx.groups)
    return Optional(got) if got is not None else Optional()


def _rx_extract_all(env, s, pattern):
    rx = _re(_need(pattern, str))
    if rx.groups > 1:
        raise CelError("regular expression has more than one capturing group")
    out = []
    for m in go_matches(rx, _need(s, str)):
        if rx.groups == 0:
            out.append(m.group(0))
        elif m.group(1):
            out.append(m.group(1))
    return out


def _opt(v):
    if not isinstance(v, Optional):
        raise no_such_overload()
    return v


def _o_value(env, o):
    if not _opt(o).has:
        raise CelError("optional.none() dereference")
    return o.value


# ---- SPIFFE (internal/conditions/types/spiffe.go over github.com/spiffe/go-spiffe/v2 v2.8.1 spiffeid, go.mod:81 - third party,
# not vendored: the published grammar of spiffeid.FromString / TrustDomainFromString / ValidatePath is restated; pinned by the 18
# TestCerbosLib KATs that use it) ----------------------------------------------------------------------------------------------
_TD_CHARS = set("abcdefghijklmnopqrstuvwxyz0123456789-._")
_SEG_CHARS = _TD_CHARS | set("ABCDEFGHIJKLMNOPQRSTUVWXYZ")


class SpiffeID:
    def __init__(self, text, pathidx):
        self.text, self.pathidx = text, pathidx

    def __eq__(self, o):
        return isinstance(o, SpiffeID) and o.text == self.text

    def __hash__(self):
        return hash(self.text)


class SpiffeTrustDomain:
    def __init__(self, name):
        self.name = name

    def __eq__(self, o):
        return isinstance(o, SpiffeTrustDomain) and o.name == self.name

    def __hash__(self):
        return hash(self.name)


class SpiffeMatcher:
    def __init__(self, kind, arg=None):
        self.kind, self.arg = kind, arg   # "any" | "exact" (id text) | "oneof" (set of id texts) | "td" (name)

    def matches(self, sid):
        if self.kind == "any":
            return True
        if self.kind == "exact":
            return sid.text == self.arg
        if self.kind == "oneof":
            return sid.text in self.arg
        return sid.text[9:sid.pathidx] == self.arg


def _spiffe_validate_path(path):
    if path == "":
        return
    if path[0] != "/":
        raise CelError("path must have a leading slash")
    start = 0
    for end, c in enumerate(path):
        if c == "/":
            seg = path[start:end]
            if seg == "/":
                raise CelError("path cannot contain empty segments")
            if seg in ("/.", "/.."):
                raise CelError("path cannot contain dot segments")
            start = end
            continue
        if c not in _SEG_CHARS:
            raise CelError("path segment characters are limited to letters, numbers, dots, dashes, and underscores")
    seg = path[start:]
    if seg == "/":
        raise CelError("path cannot have a trailing slash")
    if seg in ("/.", "/.."):
        raise CelError("path cannot contain dot segments")


def _spiffe_id_from_string(text, what="failed to parse SPIFFE ID"):
    def bad(m):
        return CelError("%s: %s" % (what, m))
    if text == "":
        raise bad("cannot be empty")
    if not text.startswith("spiffe://"):
        raise bad("scheme is missing or invalid")
    i = 9
    while i < len(text) and text[i] != "/":
        if text[i] not in _TD_CHARS:
            raise bad("trust domain characters are limited to lowercase letters, numbers, dots, dashes, and underscores")
        i += 1
    if i == 9:
        raise bad("trust domain is missing")
    try:
        _spiffe_validate_path(text[i:])
    except CelError as e:
        raise bad(str(e))
    return SpiffeID(text, i)


def _spiffe_td_from_string(text, what="failed to parse SPIFFE trust domain"):
    if text == "":
        raise CelError("%s: trust domain is missing" % what)
    if ":/" in text:
        sid = _spiffe_id_from_string(text, what)
        return SpiffeTrustDomain(sid.text[9:sid.pathidx])
    if any(c not in _TD_CHARS for c in text):
        raise CelError("%s: trust domain characters are limited to lowercase letters, numbers, dots, dashes, and underscores" % what)
    return SpiffeTrustDomain(text)


def _f_spiffe_id(env, v):                       # spiffe.go unarySPIFFEIDFnImpl
    if isinstance(v, SpiffeID):
        return v
    return _spiffe_id_from_string(_need(v, str))


def _f_spiffe_td(env, v):                       # unarySPIFFETrustDomainFnImpl
    if isinstance(v, SpiffeTrustDomain):
        return v
    if isinstance(v, SpiffeID):
        return SpiffeTrustDomain(v.text[9:v.pathidx])
    return _spiffe_td_from_string(_need(v, str))


def _f_spiffe_match_exact(env, v):
    sid = v if isinstance(v, SpiffeID) else _spiffe_id_from_string(_need(v, str))
    return SpiffeMatcher("exact", sid.text)


def _f_spiffe_match_one_of(env, lst):           # unarySPIFFEMatchOneOfFnImpl: a list of ids, else a list of strings
    lst = _need(lst, list)
    if all(isinstance(x, SpiffeID) for x in lst):
        return SpiffeMatcher("oneof", {x.text for x in lst})
    if all(isinstance(x, str) for x in lst):
        try:
            return SpiffeMatcher("oneof", {_spiffe_id_from_string(x).text for x in lst})
        except CelError:
            raise no_such_overload()
    raise no_such_overload()


def _f_spiffe_match_td(env, v):
    td = v if isinstance(v, SpiffeTrustDomain) else _spiffe_td_from_string(_need(v, str))
    return SpiffeMatcher("td", td.name)


def _m_spiffe_is_member_of(env, sid, td):
    if not isinstance(sid, SpiffeID) or not isinstance(td, SpiffeTrustDomain):
        raise no_such_overload()
    return sid.text[9:sid.pathidx] == td.name


def _m_spiffe_matches_id(env, m, v):
    if not isinstance(m, SpiffeMatcher):
        raise no_such_overload()
    sid = v if isinstance(v, SpiffeID) else _spiffe_id_from_string(_need(v, str), "invalid SPIFFE ID")
    return m.matches(sid)


def _m_spiffe_td_or_id(field):
    def g(env, v):
        if isinstance(v, SpiffeID):
            if field == "path":
                return v.text[v.pathidx:]
            if field == "trustDomain":
                return SpiffeTrustDomain(v.text[9:v.pathidx])
        if isinstance(v, SpiffeTrustDomain):
            if field == "name":
                return v.name
            if field == "id":
                return "spiffe://" + v.name
        raise no_such_overload()
    return g


# ---- file-path helpers (cerbos_lib.go:138-236 over internal/conditions/crosspath; restated in oracle/crosspath.py) ----------
def _path_fn(fn, *kinds):
    """callInString...Err (cerbos_lib.go:555-700): string / list-of-strings arguments, a Go error becomes a CEL error."""
    def g(env, *args):
        if len(args) != len(kinds):
            raise no_such_overload()
        vals = []
        for a, k in zip(args, kinds):
            if k == "s":
                vals.append(_need(a, str))
            else:
                lst = _need(a, list)
                if not all(isinstance(x, str) for x in lst):
                    raise CelError("failed to convert list to string slice")
                vals.append(list(lst))
        try:
            return fn(*vals)
        except crosspath.PathError as x:
            raise CelError(str(x))
    return g


_PATH_FUNCS = {
    "basePath": _path_fn(crosspath.base, "s"), "dirPath": _path_fn(crosspath.dir_, "s"), "extPath": _path_fn(crosspath.ext, "s"),
    "joinPath": _path_fn(crosspath.join, "l"), "relPath": _path_fn(crosspath.rel, "s", "s"),
    "volumeName": _path_fn(crosspath.volume_name, "s"),
}
# these three also have member overloads (cerbos_lib.go:174-218)
_PATH_MEMBER_FUNCS = {
    "pathHasPrefix": _path_fn(crosspath.has_prefix, "s", "s"), "pathMatch": _path_fn(crosspath.match, "s", "s"),
    "pathMatchAnyOf": _path_fn(crosspath.match_any_of, "s", "l"),
}

_GLOBAL_FUNCS = {
    "size": _f_size, "int": _f_int, "uint": _f_uint, "double": _f_double, "string": _f_string,
    "bool": _f_bool, "bytes": _f_bytes, "timestamp": _f_timestamp, "duration": _f_duration,
    "dyn": lambda env, v: v, "type": lambda env, v: type_name(v),
    "now": _f_now, "timeSince": _f_time_since,
    "intersect": _intersect, "except": _except,
    "hasIntersection": _has_intersection, "has_intersection": _has_intersection,
    "isSubset": _is_subset, "is_subset": _is_subset,
    "startsWith": lambda env, s, p: _need(s, str).startswith(_need(p, str)),
    "endsWith": lambda env, s, p: _need(s, str).endswith(_need(p, str)),
    "contains": lambda env, s, p: _need(p, str) in _need(s, str),
    "matches": _m_matches,
    "inIPAddrRange": _m_in_ip_range,
    "hierarchy": _f_hierarchy,
    "isIP": _n_is_ip, "ip": _n_ip,
    "isCIDR": lambda env, s: _try(_netip_prefix, _need(s, str)), "cidr": lambda env, s: _netip_prefix(s),
    "spiffeID": _f_spiffe_id, "spiffeTrustDomain": _f_spiffe_td, "spiffeMatchAny": lambda env: SpiffeMatcher("any"),
    "spiffeMatchExact": _f_spiffe_match_exact, "spiffeMatchOneOf": _f_spiffe_match_one_of, "spiffeMatchTrustDomain": _f_spiffe_match_td,
    **_PATH_FUNCS, **_PATH_MEMBER_FUNCS,
}

_METHODS = {
    **_PATH_MEMBER_FUNCS,
    "ancestorOf": lambda env, h, o: _h_ancestor_of(_hier(h), _hier(o)),
    "descendentOf": lambda env, h, o: _h_ancestor_of(_hier(o), _hier(h)),
    "immediateParentOf": lambda env, h, o: _h_immediate_parent_of(_hier(h), _hier(o)),
    "immediateChildOf": lambda env, h, o: _h_immediate_parent_of(_hier(o), _hier(h)),
    "siblingOf": lambda env, h, o: _h_sibling_of(_hier(h), _hier(o)),
    "overlaps": lambda env, h, o: _h_overlaps(_hier(h), _hier(o)),
    "commonAncestors": lambda env, h, o: _h_common_ancestors(_hier(h), _hier(o)),
    "size": _f_size,
    "startsWith": lambda env, s, p: _need(s, str).startswith(_need(p, str)),
    "endsWith": lambda env, s, p: _need(s, str).endswith(_need(p, str)),
    "contains": lambda env, s, p: _need(p, str) in _need(s, str),
    "matches": _m_matches,
    "charAt": _m_char_at, "indexOf": _m_index_of, "lastIndexOf": _m_last_index_of,
    "lowerAscii": lambda env, s: "".join(chr(ord(c) + 32) if "A" <= c <= "Z" else c for c in _need(s, str)),
    "upperAscii": lambda env, s: "".join(chr(ord(c) - 32) if "a" <= c <= "z" else c for c in _need(s, str)),
    "replace": _m_replace, "split": _m_split, "substring": _m_substring,
    "trim": lambda env, s: _need(s, str).strip(),
    "join": _m_join,
    "format": lambda env, s, args: _format(_need(s, str), _need(args, list)),
    "reverse": lambda env, v: v[::-1] if isinstance(v, (str, list)) else (_ for _ in ()).throw(no_such_overload()),
    "sort": _m_sort, "distinct": _m_distinct, "flatten": _m_flatten, "slice": _m_slice,
    "first": lambda env, l: _need(l, list)[0] if l else (_ for _ in ()).throw(CelError("empty list")),
    "last": lambda env, l: _need(l, list)[-1] if l else (_ for _ in ()).throw(CelError("empty list")),
    "timeSince": _f_time_since,
    "inIPAddrRange": _m_in_ip_range,
    "intersect": _intersect, "except": _except,
    "hasIntersection": _has_intersection, "has_intersection": _has_intersection,
    "isSubset": _is_subset, "is_subset": _is_subset,
    "getFullYear": _ts_getter(lambda st, ns: st.year),
    "getMonth": _ts_getter(lambda st, ns: st.month - 1),
    "getDate": _ts_getter(lambda st, ns: st.day),
    "getDayOfMonth": _ts_getter(lambda st, ns: st.day - 1),
    "getDayOfWeek": _ts_getter(lambda st, ns: (st.weekday() + 1) % 7),
    "getDayOfYear": _ts_getter(lambda st, ns: st.timetuple().tm_yday - 1),
    "getHours": _dur_or_ts(lambda ns: _trunc_div(ns, 3_600_000_000_000), lambda st, ns: st.hour),
    "getMinutes": _dur_or_ts(lambda ns: _trunc_div(ns, 60_000_000_000), lambda st, ns: st.minute),
    "getSeconds": _dur_or_ts(lambda ns: _trunc_div(ns, 1_000_000_000), lambda st, ns: st.second),
    "getMilliseconds": _dur_or_ts(lambda ns: _trunc_div(ns, 1_000_000), lambda st, ns: ns // 1_000_000),
    "family": lambda env, a: _nip(a).version,
    "isUnspecified": lambda env, a: _nip(a).is_unspecified,
    "isLoopback": lambda env, a: _nip(a).is_loopback,
    "isLinkLocalUnicast": lambda env, a: _nip(a).is_link_local,
    "isLinkLocalMulticast": lambda env, a: _n_ll_multicast(_nip(a)),
    "isGlobalUnicast": lambda env, a: _n_global_unicast(_nip(a)),
    "containsIP": _n_contains_ip, "containsCIDR": _n_contains_cidr,
    "prefixLength": lambda env, c: _ncidr(c).bits,
    "isMask": lambda env, c: _ncidr(c).network().network_address == c.addr,
    "masked": lambda env, c: NetCIDR(_ncidr(c).network().network_address, c.bits),
    "ip": lambda env, c: NetIP(_ncidr(c).addr),
    "isMemberOf": _m_spiffe_is_member_of, "matchesID": _m_spiffe_matches_id,
    "path": _m_spiffe_td_or_id("path"), "trustDomain": _m_spiffe_td_or_id("trustDomain"), "name": _m_spiffe_td_or_id("name"), "id": _m_spiffe_td_or_id("id"),
    "hasValue": lambda env, o: _opt(o).has, "value": _o_value,
    "orValue": lambda env, o, d: o.value if _opt(o).has else d,
}

_NAMESPACES = {"sets", "lists", "math", "base64", "strings", "regex", "optional", "ip"}


def _lists_range(env, n):
    if not is_int(n):
        raise no_such_overload()
    return list(range(n))


_NS_FUNCS = {
    ("sets", "contains"): lambda env, a, b: _contains_all(_to_set_list(a), _to_set_list(b)),
    ("sets", "equivalent"): lambda env, a, b: _contains_all(_to_set_list(a), _to_set_list(b)) and _contains_all(b, a),
    ("sets", "intersects"): _has_intersection,
    ("lists", "range"): _lists_range,
    ("math", "greatest"): _math_extreme(+1),
    ("math", "least"): _math_extreme(-1),
    ("math", "abs"): lambda env, v: (v if is_uint(v) else _chk_int(abs(v)) if is_int(v) else abs(v)) if is_num(v) else (_ for _ in ()).throw(no_such_overload()),
    ("math", "ceil"): lambda env, v: _dbl_round(math.ceil, v),
    ("math", "floor"): lambda env, v: _dbl_round(math.floor, v),
    ("math", "round"): lambda env, v: _dbl_round(_go_round, v),
    ("math", "trunc"): lambda env, v: _dbl_round(math.trunc, v),
    ("math", "isNaN"): lambda env, v: math.isnan(_need(v, float)),
    ("math", "isInf"): lambda env, v: math.isinf(_need(v, float)),
    ("math", "isFinite"): lambda env, v: math.isfinite(_need(v, float)),
    ("math", "sign"): _math_sign, ("math", "sqrt"): _math_sqrt,
    ("math", "bitAnd"): _math_bit(lambda a, b: a & b), ("math", "bitOr"): _math_bit(lambda a, b: a | b),
    ("math", "bitXor"): _math_bit(lambda a, b: a ^ b), ("math", "bitNot"): _math_bit_not,
    ("math", "bitShiftLeft"): _math_shift(True), ("math", "bitShiftRight"): _math_shift(False),
    ("base64", "encode"): lambda env, b: base64.b64encode(_need(b, bytes)).decode("ascii"),
    ("base64", "decode"): lambda env, s: base64.b64decode(_need(s, str) + "=" * (-len(s) % 4)),
    ("regex", "replace"): _rx_replace, ("regex", "extract"): _rx_extract, ("regex", "extractAll"): _rx_extract_all,
    ("optional", "of"): lambda env, v: Optional(v), ("optional", "none"): lambda env: Optional(),
    ("ip", "isCanonical"): lambda env, s: str(_netip_parse(s)) == s,
    ("strings", "quote"): lambda env, s: '"%s"' % _need(s, str).replace("\\", "\\\\").replace('"', '\\"'),
}
