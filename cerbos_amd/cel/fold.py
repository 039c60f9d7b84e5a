"""Constant folding of CEL expressions at lowering time.

A sub-expression that reads nothing of the request - ``cidr("10.0.0.0/8")``, ``"a,b,c".split(",")``,
``hierarchy("a.b.c").size()``, ``[3, 2, 1].sort()`` - has ONE value whatever the input, so the lowering computes it and
the device program carries the value as a constant of the table (cel-go offers the same as an optimizer,
``cel.NewConstantFoldingOptimizer``; the reference evaluates such sub-expressions on every request).  This is also what
puts the parts of the Cerbos / cel-go extension libraries that BUILD values (strings, lists, maps, hierarchies, IP
addresses) within reach of a device that only compares and tests: where their operands are constants of the policy, the
device never sees them.

What is folded: literals and container literals; arithmetic, comparison and logic on them; indexing and selection; the
comprehension macros (one- and two-variable forms, ``transformList`` / ``transformMap`` / ``transformMapEntry`` /
``sortBy``) over constant ranges; cel-go ``ext.Strings`` / ``ext.Lists`` / ``ext.Sets`` / ``ext.Encoders`` / ``ext.Regex``
(simple patterns) / the network helpers (``isIP``, ``ip``, ``isCIDR``, ``cidr``); the Cerbos library's list, hierarchy
and address-range functions (internal/conditions/cerbos_lib.go:96-244, types/hierarchy.go).  Everything else - and any
constant expression whose evaluation is a CEL ERROR - is left exactly as written: the device evaluates (or flags) it.

Semantics follow cel-go v0.x as pinned by the reference's own known-answer tests
(internal/test/testdata/cel_eval/*.yaml, internal/conditions/cerbos_lib_test.go); tests/test_cel_fold.py holds this
module against oracle/celeval.py on those and on generated expressions.  When in doubt the folder declines.
"""
from __future__ import annotations

import base64 as _b64
import ipaddress
import math
import re

from . import crosspath

_PATH_FUNCS = {"basePath": (crosspath.base, "s"), "dirPath": (crosspath.dir_, "s"), "extPath": (crosspath.ext, "s"),
               "joinPath": (crosspath.join, "l"), "relPath": (crosspath.rel, "ss"), "volumeName": (crosspath.volume_name, "s"),
               "pathHasPrefix": (crosspath.has_prefix, "ss"), "pathMatch": (crosspath.match, "ss"), "pathMatchAnyOf": (crosspath.match_any_of, "sl")}
_PATH_MEMBERS = ("pathHasPrefix", "pathMatch", "pathMatchAnyOf")

INT_MIN, INT_MAX, UINT_MAX = -(1 << 63), (1 << 63) - 1, (1 << 64) - 1


class NotConst(Exception):
    """The expression is not a constant the folder can compute."""


class FoldError(Exception):
    """Evaluating the constant expression is a CEL error: it is left to the device."""


class Unknown(NotConst):
    """The expression reads a part of the request the caller has not given (the query planner's unknown resource attributes)."""


class PartialMap(dict):
    """A map of which only some keys are known (cerbos_amd/plan): a key it does not hold is unknown, not absent."""


class UInt(int):
    pass


class Hier(tuple):
    """cerbos.lib.hierarchy (types/hierarchy.go): the segments."""


class Opt:
    def __init__(self, present, value=None):
        self.present, self.value = present, value


class IPAddr:
    def __init__(self, a):
        self.a = a


class CIDR:
    def __init__(self, addr, bits):
        self.addr, self.bits = addr, bits   # the address AS WRITTEN (not masked) and the prefix length


class SpiffeId:
    """cerbos.lib.spiffeID (types/spiffe.go over go-spiffe v2 spiffeid.ID): the id's text and where its path begins."""

    def __init__(self, text, pathidx):
        self.text, self.pathidx = text, pathidx

    @property
    def domain(self):
        return self.text[9:self.pathidx]


class SpiffeDomain:
    """cerbos.lib.spiffeTrustDomain"""

    def __init__(self, name):
        self.name = name


class SpiffeMatch:
    """cerbos.lib.spiffeMatcher: any | one id | one of several | every id of a trust domain"""

    def __init__(self, kind, arg=None):
        self.kind, self.arg = kind, arg


def _is_int(v):
    return isinstance(v, int) and not isinstance(v, (bool, UInt))


def _is_num(v):
    return isinstance(v, (int, float)) and not isinstance(v, bool)


def _num_cmp(a, b):
    """Mathematical order across int / uint / double; None when a NaN is involved."""
    if isinstance(a, float) and math.isnan(a) or isinstance(b, float) and math.isnan(b):
        return None
    if isinstance(a, float) != isinstance(b, float):
        f, i, sign = (a, b, 1) if isinstance(a, float) else (b, a, -1)
        if math.isinf(f):
            return sign * (1 if f > 0 else -1)
        whole = int(f)
        r = (whole > i) - (whole < i)
        if r == 0:
            frac = f - whole
            r = (frac > 0) - (frac < 0)
        return sign * r
    return (a > b) - (a < b)


def equal(a, b):
    if _is_num(a) and _is_num(b):
        return _num_cmp(a, b) == 0
    if a is None or b is None:
        return a is None and b is None
    if isinstance(a, Hier) or isinstance(b, Hier):
        return isinstance(a, Hier) and isinstance(b, Hier) and tuple(a) == tuple(b)
    if isinstance(a, bool) or isinstance(b, bool):
        return isinstance(a, bool) and isinstance(b, bool) and a == b
    if isinstance(a, (str, bytes)):
        return type(a) is type(b) and a == b
    if isinstance(a, list):
        return isinstance(b, list) and len(a) == len(b) and all(equal(x, y) for x, y in zip(a, b))
    if isinstance(a, dict):
        if not isinstance(b, dict) or len(a) != len(b):
            return False
        for k, v in a.items():
            found, bv = _map_get(b, k)
            if not found or not equal(v, bv):
                return False
        return True
    if isinstance(a, Opt):
        return isinstance(b, Opt) and a.present == b.present and (not a.present or equal(a.value, b.value))
    if isinstance(a, IPAddr):
        return isinstance(b, IPAddr) and a.a == b.a
    if isinstance(a, CIDR):
        return isinstance(b, CIDR) and a.addr == b.addr and a.bits == b.bits
    if isinstance(a, SpiffeDomain):   # spiffe.go SPIFFETrustDomain.Equal (cel-go asks the left operand): a trust domain, or a string that parses to one
        if isinstance(b, SpiffeDomain):
            return a.name == b.name
        if isinstance(b, str):
            try:
                return _spiffe_domain(b).name == a.name
            except FoldError:
                return False
        raise FoldError("no such overload")
    if isinstance(a, SpiffeId):       # SPIFFEID.Equal: an id, or its string form
        if isinstance(b, SpiffeId):
            return a.text == b.text
        if isinstance(b, str):
            return a.text == b
        raise FoldError("no such overload")
    if isinstance(a, SpiffeMatch):
        return False
    raise NotConst("equality of %s" % type(a).__name__)


def _map_get(m, k):
    if isinstance(k, bool) or isinstance(k, str):
        for mk, mv in m.items():
            if type(mk) is type(k) and mk == k:
                return True, mv
        if isinstance(m, PartialMap):
            raise Unknown(str(k))
        return False, None
    if _is_num(k):
        for mk, mv in m.items():
            if _is_num(mk) and _num_cmp(mk, k) == 0:
                return True, mv
    if isinstance(m, PartialMap):
        raise Unknown(str(k))
    return False, None


def compare(a, b):
    if _is_num(a) and _is_num(b):
        return _num_cmp(a, b)
    if isinstance(a, bool) and isinstance(b, bool):
        return (a > b) - (a < b)
    if isinstance(a, str) and isinstance(b, str):
        x, y = a.encode("utf-8"), b.encode("utf-8")
        return (x > y) - (x < y)
    if isinstance(a, bytes) and isinstance(b, bytes):
        return (a > b) - (a < b)
    raise FoldError("no such overload")


def _int(v):
    if v < INT_MIN or v > INT_MAX:
        raise FoldError("integer overflow")
    return int(v)


def _uint(v):
    if v < 0 or v > UINT_MAX:
        raise FoldError("unsigned integer overflow")
    return UInt(v)


_M64 = (1 << 64) - 1


def _as_int64(bits):
    bits &= _M64
    return bits - (1 << 64) if bits >> 63 else bits


def _math(name, vals):
    """cel-go ext.Math() (cerbos enables it unversioned = latest, conditions/cel.go:78) on constants - greatest / least are
    handled by the caller.  Rounding is half away from zero; bit operations want two ints or two uints; a shift takes an int
    offset (negative: an error; 64 and more: zero) and moves the 64-bit pattern, so a right shift of a negative int fills
    with zeros; sign keeps the operand's type."""
    n = len(vals)
    if n == 1:
        v = vals[0]
        if name in ("ceil", "floor", "round", "trunc", "isNaN", "isInf", "isFinite"):
            d = _need(v, float)
            if name == "isNaN":
                return math.isnan(d)
            if name == "isInf":
                return math.isinf(d)
            if name == "isFinite":
                return not (math.isnan(d) or math.isinf(d))
            if math.isnan(d) or math.isinf(d):
                return d
            if name == "ceil":
                return float(math.ceil(d))
            if name == "floor":
                return float(math.floor(d))
            if name == "trunc":
                return float(math.trunc(d))
            t = float(math.trunc(d))       # Go's math.Round: half away from zero, exact where floor(|d| + 0.5) is not
            return math.copysign(t + math.copysign(1.0, d) if abs(d - t) >= 0.5 else t, d)
        if name == "abs":
            if isinstance(v, UInt):
                return v
            if _is_int(v):
                return _int(-v if v < 0 else v)
            return abs(_need(v, float))
        if name == "sign":
            if isinstance(v, UInt):
                return UInt(1 if v else 0)
            if _is_int(v):
                return 1 if v > 0 else -1 if v < 0 else 0
            d = _need(v, float)
            return d if math.isnan(d) else 1.0 if d > 0 else -1.0 if d < 0 else 0.0
        if name == "sqrt":
            if not _is_num(v):
                raise FoldError("no such overload")
            d = float(v)
            return math.nan if math.isnan(d) or d < 0 else math.sqrt(d)
        if name == "bitNot":
            if isinstance(v, UInt):
                return UInt(~int(v) & _M64)
            return ~_need(v, int)
    if n == 2:
        a, b = vals
        if name in ("bitAnd", "bitOr", "bitXor"):
            both_u = isinstance(a, UInt) and isinstance(b, UInt)
            if not both_u and not (_is_int(a) and _is_int(b)):
                raise FoldError("no such overload")
            r = (int(a) & int(b)) if name == "bitAnd" else (int(a) | int(b)) if name == "bitOr" else (int(a) ^ int(b))
            return UInt(r & _M64) if both_u else _as_int64(r)
        if name in ("bitShiftLeft", "bitShiftRight"):
            if not (isinstance(a, UInt) or _is_int(a)) or not _is_int(b):
                raise FoldError("no such overload")
            if b < 0:
                raise FoldError("negative offset")
            if b >= 64:
                return UInt(0) if isinstance(a, UInt) else 0
            bits = int(a) & _M64
            r = (bits << b) & _M64 if name == "bitShiftLeft" else bits >> b
            return UInt(r) if isinstance(a, UInt) else _as_int64(r)
    raise NotConst("math.%s" % name)


def _cps(s):
    return list(s)   # Python strings index by code point, as cel-go's string extensions do


def _need(v, *types):
    for t in types:
        if t is int:
            if _is_int(v):
                return v
        elif isinstance(v, t) and not (t is not bool and isinstance(v, bool)):
            return v
    raise FoldError("no such overload")


# ---- hierarchy (types/hierarchy.go)
def _hierarchy(v, delim="."):
    if isinstance(v, Hier):
        return v
    if isinstance(v, str):
        _need(delim, str)
        return Hier(v.split(delim)) if delim else Hier(list(v))
    if isinstance(v, list) and all(isinstance(x, str) for x in v):
        return Hier(v)
    raise FoldError("no such overload")


def _h_ancestor(h, c):           # hierarchy.go:259-276
    return len(h) < len(c) and tuple(c[:len(h)]) == tuple(h)


def _h_common(h, o):             # hierarchy.go:278-306
    n = 0
    lim = min(len(h), len(o))
    while n < lim and h[n] == o[n]:
        n += 1
    if n == len(h) == len(o):    # identical: the common ancestors stop at the parent
        n -= 1
    return Hier(h[:max(n, 0)])


def _h_imm_parent(h, c):         # hierarchy.go:326-343
    return len(h) + 1 == len(c) and tuple(c[:len(h)]) == tuple(h)


def _h_sibling(h, o):            # hierarchy.go:345-362
    return len(h) == len(o) and len(h) > 0 and tuple(h[:-1]) == tuple(o[:-1])


def _h_overlaps(h, o):           # hierarchy.go:364-385
    n = min(len(h), len(o))
    return tuple(h[:n]) == tuple(o[:n])


# ---- network (k8s-style helpers as cel-go / the reference expose them)
def _parse_ip(s):
    if not isinstance(s, str) or "%" in s:
        return None
    try:
        a = ipaddress.ip_address(s)
    except ValueError:
        return None
    if isinstance(a, ipaddress.IPv6Address) and a.ipv4_mapped is not None:
        return None   # IPv4-mapped IPv6 addresses are rejected
    if isinstance(a, ipaddress.IPv4Address) and any(len(p) > 1 and p[0] == "0" for p in s.split(".")):
        return None   # netip.ParseAddr: no leading zeros in a dotted quad
    return a


def _parse_cidr(s):
    if not isinstance(s, str) or s.count("/") != 1:
        return None
    addr, bits = s.split("/")
    a = _parse_ip(addr)
    if a is None or not bits.isdigit() or (len(bits) > 1 and bits[0] == "0"):
        return None
    n = int(bits)
    if n > (32 if a.version == 4 else 128):
        return None
    return CIDR(a, n)


def _cidr_contains_ip(c, a):
    if a.version != c.addr.version:
        return False
    width = 32 if a.version == 4 else 128
    shift = width - c.bits
    return (int(a) >> shift) == (int(c.addr) >> shift)


def _ip_method(name, a):
    v, n = a.version, int(a)
    if name == "family":
        return 4 if v == 4 else 6
    if name == "isUnspecified":
        return n == 0
    if name == "isLoopback":
        return (n >> 24) == 127 if v == 4 else n == 1
    multicast = (n >> 28) == 0xE if v == 4 else (n >> 120) == 0xFF
    ll_unicast = (n >> 16) == 0xA9FE if v == 4 else (n >> 118) == (0xFE80 >> 6)
    if name == "isLinkLocalUnicast":
        return ll_unicast
    if name == "isLinkLocalMulticast":
        return (n >> 8) == 0xE00000 if v == 4 else (n >> 112) & 0xFF0F == 0xFF02
    if name == "isGlobalUnicast":
        loop = (n >> 24) == 127 if v == 4 else n == 1
        return n != 0 and not (v == 4 and n == 0xFFFFFFFF) and not loop and not multicast and not ll_unicast
    raise NotConst(name)


# ---- regex (cel-go ext.Regex): only patterns whose RE2 and Python readings coincide
_SAFE_PATTERN = re.compile(r"^(?:[A-Za-z0-9 _\-,:;@#%&=<>!~'\"/]|\\[dDwWsS.\\]|[.*+?()|]|\[\^?[A-Za-z0-9_\-]+\])*$")


def _regex(p):
    """RE2 (Go regexp) reads \\d \\w \\s as ASCII classes and \\s without \\v; Python's `re` needs re.ASCII and its own
    spelling of \\s for the same reading (the whitelist admits those escapes outside brackets only)."""
    if not isinstance(p, str) or not _SAFE_PATTERN.match(p) or "(?" in p:
        raise NotConst("regular expression outside the folder's subset")
    py = re.sub(r"\\([sS\\])", lambda m: {"s": "[\t\n\f\r ]", "S": "[^\t\n\f\r ]", "\\": "\\\\"}[m.group(1)], p)
    try:
        return re.compile(py, re.ASCII)
    except re.error:
        raise NotConst("regular expression")


def _go_matches(rx, s):
    """Go's regexp drops an EMPTY match that begins where the previous match ended (regexp.go allMatches); Python keeps it."""
    prev_end = -1
    for m in rx.finditer(s):
        if m.start() == m.end() == prev_end:
            continue
        prev_end = m.end()
        yield m


def _regex_replace(s, p, repl, limit=-1):
    rx = _regex(p)
    if not all(ch.isalnum() or ch in " _-\\" for ch in repl) or re.search(r"\\(?![0-9])", repl):
        raise NotConst("replacement outside the folder's subset")
    if limit == 0:
        return s
    if re.search(r"\\([0-9])", repl) and any(int(d) > rx.groups for d in re.findall(r"\\([0-9])", repl)):
        raise FoldError("invalid replacement")
    template = re.sub(r"\\([0-9])", r"\\g<\1>", repl)
    out, at, n = [], 0, 0
    for m in _go_matches(rx, s):
        if 0 <= limit <= n:
            break
        out += [s[at:m.start()], m.expand(template)]
        at = m.end()
        n += 1
    out.append(s[at:])
    return "".join(out)


# ---- SPIFFE ids (types/spiffe.go; the grammar of github.com/spiffe/go-spiffe/v2 v2.8.1 spiffeid - go.mod:81, not vendored:
# FromString, TrustDomainFromString, ValidatePath as published)
_SPIFFE_TD = frozenset("abcdefghijklmnopqrstuvwxyz0123456789-._")
_SPIFFE_SEG = _SPIFFE_TD | frozenset("ABCDEFGHIJKLMNOPQRSTUVWXYZ")


def _spiffe_id(text):
    if not text.startswith("spiffe://"):
        raise FoldError("failed to parse SPIFFE ID")
    i = 9
    while i < len(text) and text[i] != "/":
        if text[i] not in _SPIFFE_TD:
            raise FoldError("failed to parse SPIFFE ID")
        i += 1
    if i == 9:
        raise FoldError("failed to parse SPIFFE ID")
    path = text[i:]
    if path:
        segs = path.split("/")[1:]   # path starts with '/': the first piece is empty
        if any(seg in ("", ".", "..") or any(c not in _SPIFFE_SEG for c in seg) for seg in segs):
            raise FoldError("failed to parse SPIFFE ID")
    return SpiffeId(text, i)


def _spiffe_domain(text):
    if text == "":
        raise FoldError("failed to parse SPIFFE trust domain")
    if ":/" in text:
        try:
            return SpiffeDomain(_spiffe_id(text).domain)
        except FoldError:
            raise FoldError("failed to parse SPIFFE trust domain")
    if any(c not in _SPIFFE_TD for c in text):
        raise FoldError("failed to parse SPIFFE trust domain")
    return SpiffeDomain(text)


class _Eval:
    def __init__(self):
        self.steps = 0

    def ev(self, n, env):   # noqa: C901
        self.steps += 1
        if self.steps > 200_000:
            raise NotConst("too large")
        k = n[0]
        if k == "lit":
            if n[1] == "folded":
                return n[2]
            if n[1] == "uint":
                return UInt(n[2])
            if n[1] == "bytes":
                return bytes(n[2]) if not isinstance(n[2], bytes) else n[2]
            return n[2]
        if k == "ident":
            if n[1] in env:
                return env[n[1]]
            raise NotConst(n[1])
        if k == "list":
            return [self.ev(e, env) for e in n[1]]
        if k == "map":
            pairs = []
            for ke, ve in n[1]:
                kk = self.ev(ke, env)
                if not isinstance(kk, (bool, str)) and not (_is_num(kk) and not isinstance(kk, float)):
                    raise FoldError("unsupported key type")
                pairs.append((kk, self.ev(ve, env)))
            return _mkmap(pairs)
        if k == "not":
            v = self.ev(n[1], env)
            return not _need(v, bool)
        if k == "neg":
            v = self.ev(n[1], env)
            if _is_int(v):
                return _int(-v)
            if isinstance(v, float):
                return -v
            raise FoldError("no such overload")
        if k in ("and", "or"):
            # both operands must be constants for the node to fold (an error on one side is left to the device)
            a, b = _need(self.ev(n[1], env), bool), _need(self.ev(n[2], env), bool)
            return (a and b) if k == "and" else (a or b)
        if k == "tern":
            c = _need(self.ev(n[1], env), bool)
            return self.ev(n[2] if c else n[3], env)
        if k == "bin":
            return self.binop(n[1], self.ev(n[2], env), self.ev(n[3], env))
        if k == "index":
            return self.index(self.ev(n[1], env), self.ev(n[2], env))
        if k == "select":
            v = self.ev(n[1], env)
            if isinstance(v, dict):
                found, out = _map_get(v, n[2])
                if not found:
                    raise FoldError("no such key")
                return out
            raise NotConst("select")
        if k == "has":
            v = self.ev(n[1], env)
            if isinstance(v, dict):
                return _map_get(v, n[2])[0]
            raise NotConst("has")
        if k == "bind":
            return self.ev(n[3], dict(env, **{n[1]: self.ev(n[2], env)}))
        if k == "comp":
            return self.comp(n, env)
        if k == "call":
            return self.call(n, env)
        raise NotConst(k)

    # ---- operators
    def binop(self, op, a, b):   # noqa: C901
        if op == "==":
            return equal(a, b)
        if op == "!=":
            return not equal(a, b)
        if op in ("<", "<=", ">", ">="):
            r = compare(a, b)
            if r is None:
                return False
            return {"<": r < 0, "<=": r <= 0, ">": r > 0, ">=": r >= 0}[op]
        if op == "in":
            if isinstance(b, list):
                return any(equal(a, x) for x in b)
            if isinstance(b, dict):
                return _map_get(b, a)[0]
            raise FoldError("no such overload")
        if op == "+":
            if isinstance(a, str) and isinstance(b, str):
                return a + b
            if isinstance(a, bytes) and isinstance(b, bytes):
                return a + b
            if isinstance(a, list) and isinstance(b, list):
                return a + b
        if isinstance(a, bool) or isinstance(b, bool) or not (_is_num(a) and _is_num(b)):
            raise FoldError("no such overload")
        if isinstance(a, float) != isinstance(b, float) or isinstance(a, UInt) != isinstance(b, UInt):
            raise FoldError("no such overload")   # arithmetic does not mix numeric types
        if isinstance(a, float):
            if op == "+":
                return a + b
            if op == "-":
                return a - b
            if op == "*":
                return a * b
            if op == "/":
                if b == 0:
                    return math.copysign(math.inf, a) * math.copysign(1.0, b) if a != 0 and not math.isnan(a) else math.nan
                return a / b
            raise FoldError("no such overload")
        chk = _uint if isinstance(a, UInt) else _int
        if op == "+":
            return chk(a + b)
        if op == "-":
            return chk(a - b)
        if op == "*":
            return chk(a * b)
        if op in ("/", "%"):
            if b == 0:
                raise FoldError("division by zero" if op == "/" else "modulus by zero")
            q = abs(a) // abs(b) * (1 if (a < 0) == (b < 0) else -1)   # Go truncates toward zero
            return chk(q) if op == "/" else chk(a - b * q)
        raise NotConst(op)

    def index(self, v, i):
        if isinstance(v, Hier):
            v = list(v)
        if isinstance(v, list):
            if isinstance(i, float) and i == int(i):
                i = int(i)
            if not _is_num(i) or isinstance(i, float):
                raise FoldError("no such overload")
            if i < 0 or i >= len(v):
                raise FoldError("index out of range")
            return v[i]
        if isinstance(v, dict):
            found, out = _map_get(v, i)
            if not found:
                raise FoldError("no such key")
            return out
        raise FoldError("no such overload")

    # ---- comprehension macros
    def comp(self, n, env):   # noqa: C901
        _, kind, target, vars_, args = n
        rng = self.ev(target, env)
        if isinstance(rng, list):
            pairs = [(i, x) for i, x in enumerate(rng)]
            one = [x for x in rng]
        elif isinstance(rng, dict):
            pairs = list(rng.items())
            one = list(rng.keys())
        else:
            raise FoldError("no such overload")
        two = len(vars_) == 2

        def scope(item):
            e = dict(env)
            if two:
                e[vars_[0]], e[vars_[1]] = item
            else:
                e[vars_[0]] = item
            return e
        items = pairs if two else one
        if kind in ("all", "exists", "exists_one", "existsOne"):
            # (errors inside are left to the device: evaluating every element is only right when none of them fails)
            vals = [_need(self.ev(args[0], scope(it)), bool) for it in items]
            if kind == "all":
                return all(vals)
            if kind == "exists":
                return any(vals)
            return sum(vals) == 1
        if kind == "filter":
            return [it for it in one if _need(self.ev(args[0], scope(it)), bool)]
        if kind in ("map", "transformList"):
            out = []
            for it in items:
                e = scope(it)
                if len(args) == 2 and not _need(self.ev(args[0], e), bool):
                    continue
                out.append(self.ev(args[-1], e))
            return out
        if kind == "transformMap":
            out = []
            for it in pairs:
                e = scope(it)
                if len(args) == 2 and not _need(self.ev(args[0], e), bool):
                    continue
                out.append((it[0], self.ev(args[-1], e)))
            return _mkmap(out)
        if kind == "transformMapEntry":
            out = []
            for it in pairs:
                e = scope(it)
                if len(args) == 2 and not _need(self.ev(args[0], e), bool):
                    continue
                ent = self.ev(args[-1], e)
                if not isinstance(ent, dict):
                    raise FoldError("no such overload")
                out.extend(ent.items())
            return _mkmap(out)
        if kind == "sortBy":
            keyed = [(self.ev(args[0], scope(it)), it) for it in one]
            return [it for _, it in _sorted(keyed, key=lambda t: t[0])]
        raise NotConst(kind)

    # ---- functions
    def call(self, n, env):   # noqa: C901
        _, name, target, args = n
        ns = None
        if target is not None and target[0] == "ident" and target[1] not in env and \
                target[1] in ("sets", "math", "lists", "base64", "strings", "regex", "optional", "ip", "cidr"):
            ns = target[1]
            vals = [self.ev(a, env) for a in args]
        else:
            vals = ([self.ev(target, env)] if target is not None else []) + [self.ev(a, env) for a in args]
        nv = len(vals)
        method = target is not None and ns is None
        if ns == "base64":
            if name == "encode" and nv == 1:
                return _b64.b64encode(_need(vals[0], bytes)).decode("ascii")
            if name == "decode" and nv == 1:
                s = _need(vals[0], str)
                try:
                    return _b64.b64decode(s + "=" * (-len(s) % 4), validate=True)
                except Exception:
                    raise FoldError("illegal base64 data")
        if ns == "optional":
            if name == "of" and nv == 1:
                return Opt(True, vals[0])
            if name == "none" and nv == 0:
                return Opt(False)
        if ns == "lists" and name == "range" and nv == 1:
            return [i for i in range(max(0, _need(vals[0], int)))]
        if ns == "sets" and nv == 2 and all(isinstance(v, list) for v in vals):
            a, b = vals
            if name == "contains":
                return all(any(equal(x, y) for y in a) for x in b)
            if name == "intersects":
                return any(equal(x, y) for x in a for y in b)
            if name == "equivalent":
                return all(any(equal(x, y) for y in a) for x in b) and all(any(equal(x, y) for y in b) for x in a)
        if ns == "math" and name in ("greatest", "least") and nv >= 1:
            xs = vals[0] if nv == 1 and isinstance(vals[0], list) else vals
            if not xs or not all(_is_num(x) for x in xs):
                raise FoldError("no such overload")
            best = xs[0]
            for x in xs[1:]:
                r = _num_cmp(x, best)
                if r is None:
                    raise NotConst("NaN")
                if (r > 0) == (name == "greatest") and r != 0:
                    best = x
            return best
        if ns == "math":
            return _math(name, vals)
        if ns == "regex":
            if name == "replace" and nv in (3, 4):
                return _regex_replace(_need(vals[0], str), vals[1], _need(vals[2], str), _need(vals[3], int) if nv == 4 else -1)
            if name == "extract" and nv == 2:
                rx = _regex(vals[1])
                if rx.groups > 1:
                    raise FoldError("regular expression has more than one capturing group")
                m = rx.search(_need(vals[0], str))
                if m is None:
                    return Opt(False)
                got = m.group(1) if rx.groups == 1 else m.group(0)
                if rx.groups == 1 and not got:
                    raise NotConst("empty capture")
                return Opt(True, got)
            if name == "extractAll" and nv == 2:
                rx = _regex(vals[1])
                if rx.groups > 1:
                    raise FoldError("regular expression has more than one capturing group")
                out = []
                for m in _go_matches(rx, _need(vals[0], str)):
                    got = m.group(1) if rx.groups == 1 else m.group(0)
                    if got is not None and (rx.groups == 0 or got != ""):
                        out.append(got)
                return out
        if ns == "ip" and name == "isCanonical" and nv == 1:
            a = _parse_ip(_need(vals[0], str))
            if a is None:
                raise FoldError("IP Address %r parse error during conversion from string" % vals[0])
            return str(a) == vals[0]
        if ns is not None:
            raise NotConst("%s.%s" % (ns, name))

        # ---- global functions and methods
        if name == "size" and nv == 1:
            v = vals[0]
            if isinstance(v, str):
                return len(v)
            if isinstance(v, (bytes, list, dict, Hier)):
                return len(v)
            raise FoldError("no such overload")
        if name in ("startsWith", "endsWith", "contains") and nv == 2 and method:
            s, t = _need(vals[0], str), _need(vals[1], str)
            return s.startswith(t) if name == "startsWith" else s.endswith(t) if name == "endsWith" else t in s
        if name == "dyn" and nv == 1 and not method:
            return vals[0]
        if name == "int" and nv == 1 and not method:
            v = vals[0]
            if _is_int(v):
                return v
            if isinstance(v, UInt):
                return _int(int(v))
            if isinstance(v, float):
                # cel-go doubleToInt64Checked: both ends excluded (-2^63 itself, exact as a double, is an overflow)
                if math.isnan(v) or math.isinf(v) or v <= -9223372036854775808.0 or v >= 9223372036854775808.0:
                    raise FoldError("integer overflow")
                return _int(int(v))
            if isinstance(v, str):
                if not re.fullmatch(r"[+-]?[0-9]+", v):
                    raise FoldError("cannot convert string to int")
                return _int(int(v))
            raise NotConst("int()")
        if name == "uint" and nv == 1 and not method:
            v = vals[0]
            if isinstance(v, UInt):
                return v
            if _is_int(v):
                return _uint(v)
            if isinstance(v, float):
                if math.isnan(v) or math.isinf(v) or v < 0 or v >= 1.8446744073709552e19:
                    raise FoldError("unsigned integer overflow")
                return _uint(int(v))
            if isinstance(v, str):
                if not re.fullmatch(r"[0-9]+", v):
                    raise FoldError("cannot convert string to uint")
                return _uint(int(v))
            raise NotConst("uint()")
        if name == "double" and nv == 1 and not method:
            v = vals[0]
            if isinstance(v, float):
                return v
            if _is_num(v):
                return float(v)
            raise NotConst("double()")
        if name == "string" and nv == 1 and not method:
            v = vals[0]
            if isinstance(v, str):
                return v
            if isinstance(v, bool):
                return "true" if v else "false"
            if _is_num(v) and not isinstance(v, float):
                return str(int(v))
            if isinstance(v, bytes):
                try:
                    return v.decode("utf-8")
                except UnicodeDecodeError:
                    raise FoldError("invalid UTF-8")
            if isinstance(v, IPAddr):
                return str(v.a)
            raise NotConst("string()")
        if name == "bytes" and nv == 1 and not method:
            v = vals[0]
            if isinstance(v, bytes):
                return v
            if isinstance(v, str):
                return v.encode("utf-8")
            raise FoldError("no such overload")
        if name == "bool" and nv == 1 and not method:
            v = vals[0]
            if isinstance(v, bool):
                return v
            if isinstance(v, str):
                if v in ("1", "t", "true", "TRUE", "True"):
                    return True
                if v in ("0", "f", "false", "FALSE", "False"):
                    return False
                raise FoldError("cannot convert string to bool")
            raise NotConst("bool()")

        # strings (cel-go ext/strings.go)
        if method and isinstance(vals[0], str):
            s = vals[0]
            if name == "charAt" and nv == 2:
                i, cp = _need(vals[1], int), _cps(s)
                if i < 0 or i > len(cp):
                    raise FoldError("index out of range")
                return "" if i == len(cp) else cp[i]
            if name == "indexOf" and nv in (2, 3):
                sub, cp = _need(vals[1], str), _cps(s)
                off = _need(vals[2], int) if nv == 3 else 0
                if off < 0 or off > len(cp):
                    raise FoldError("index out of range")
                return s.find(sub, off) if sub else off
            if name == "lastIndexOf" and nv == 2:
                sub = _need(vals[1], str)
                return s.rfind(sub) if sub else len(_cps(s))
            if name in ("lowerAscii", "upperAscii") and nv == 1:
                f = (lambda c: c.lower()) if name == "lowerAscii" else (lambda c: c.upper())
                return "".join(f(c) if c.isascii() else c for c in s)
            if name == "replace" and nv in (3, 4):
                old, new = _need(vals[1], str), _need(vals[2], str)
                lim = _need(vals[3], int) if nv == 4 else -1
                return s.replace(old, new) if lim < 0 else s.replace(old, new, lim)
            if name == "split" and nv in (2, 3):
                sep, lim = _need(vals[1], str), (_need(vals[2], int) if nv == 3 else -1)
                if lim == 0:
                    return []
                if lim == 1:
                    return [s]
                if sep == "":
                    cp = _cps(s)
                    if lim > 0 and len(cp) > lim:
                        return cp[:lim - 1] + ["".join(cp[lim - 1:])]
                    return cp
                return s.split(sep) if lim < 0 else s.split(sep, lim - 1)
            if name == "substring" and nv in (2, 3):
                cp = _cps(s)
                a = _need(vals[1], int)
                b = _need(vals[2], int) if nv == 3 else len(cp)
                if a < 0 or a > len(cp) or b < 0 or b > len(cp):
                    raise FoldError("index out of range")
                if a > b:
                    raise FoldError("invalid substring range")
                return "".join(cp[a:b])
            if name == "trim" and nv == 1:
                return s.strip(_GO_SPACE)
            if name == "reverse" and nv == 1:
                return "".join(reversed(_cps(s)))
            if name == "inIPAddrRange" and nv == 2:
                a, c = _parse_ip_loose(s), _parse_cidr_loose(_need(vals[1], str))
                if a is None or c is None:
                    raise FoldError("invalid address")
                return _cidr_contains_ip(c, a)
        if name == "join" and method and isinstance(vals[0], list) and nv in (1, 2):
            sep = _need(vals[1], str) if nv == 2 else ""
            return sep.join(_need(x, str) for x in vals[0])

        # lists (cel-go ext/lists.go, cerbos_lib.go)
        if method and isinstance(vals[0], list):
            lst = vals[0]
            if name == "sort" and nv == 1:
                return _sorted(lst)
            if name == "distinct" and nv == 1:
                out = []
                for x in lst:
                    if not any(equal(x, y) for y in out):
                        out.append(x)
                return out
            if name == "flatten" and nv in (1, 2):
                depth = _need(vals[1], int) if nv == 2 else 1
                if depth < 0:
                    raise FoldError("level must be non-negative")
                return _flatten(lst, depth)
            if name == "reverse" and nv == 1:
                return list(reversed(lst))
            if name == "slice" and nv == 3:
                a, b = _need(vals[1], int), _need(vals[2], int)
                if a < 0 or b < 0 or a > b or b > len(lst):
                    raise FoldError("index out of range")
                return lst[a:b]
            if name in ("first", "last") and nv == 1:
                return Opt(False) if not lst else Opt(True, lst[0 if name == "first" else -1])
        if name in ("intersect", "except", "hasIntersection", "has_intersection", "isSubset", "is_subset") and nv == 2:
            a, b = vals
            if not isinstance(a, list) or not isinstance(b, list):
                raise FoldError("no such overload")
            if name == "intersect":   # cerbos_lib.go:416-452: the shorter list is the one iterated
                x, y = (a, b) if len(a) <= len(b) else (b, a)
                return [e for e in x if any(equal(e, f) for f in y)]
            if name == "except":
                return [e for e in a if not any(equal(e, f) for f in b)]
            if name in ("hasIntersection", "has_intersection"):
                return any(equal(e, f) for e in a for f in b)
            return all(any(equal(e, f) for f in b) for e in a)

        # file paths (cerbos_lib.go:138-236 over crosspath): the three with member overloads take their first argument as the target
        if ns is None and name in _PATH_FUNCS and (not method or name in _PATH_MEMBERS):
            fn, kinds = _PATH_FUNCS[name]
            if nv != len(kinds):
                raise FoldError("no such overload")
            args = []
            for v, kind in zip(vals, kinds):
                if kind == "s":
                    args.append(_need(v, str))
                elif isinstance(v, list) and all(isinstance(x, str) for x in v):
                    args.append(v)
                else:
                    raise FoldError("not a list of strings")
            try:
                return fn(*args)
            except crosspath.PathError as err:
                raise FoldError(str(err))

        # SPIFFE (types/spiffe.go)
        if not method and ns is None and name.startswith("spiffe"):
            if name == "spiffeID" and nv == 1:
                return vals[0] if isinstance(vals[0], SpiffeId) else _spiffe_id(_need(vals[0], str))
            if name == "spiffeTrustDomain" and nv == 1:
                v0 = vals[0]
                return v0 if isinstance(v0, SpiffeDomain) else SpiffeDomain(v0.domain) if isinstance(v0, SpiffeId) else _spiffe_domain(_need(v0, str))
            if name == "spiffeMatchAny" and nv == 0:
                return SpiffeMatch("any")
            if name == "spiffeMatchExact" and nv == 1:
                return SpiffeMatch("exact", (vals[0] if isinstance(vals[0], SpiffeId) else _spiffe_id(_need(vals[0], str))).text)
            if name == "spiffeMatchOneOf" and nv == 1:
                lst = _need(vals[0], list)
                if all(isinstance(x, SpiffeId) for x in lst):
                    return SpiffeMatch("oneof", frozenset(x.text for x in lst))
                if all(isinstance(x, str) for x in lst):
                    try:
                        return SpiffeMatch("oneof", frozenset(_spiffe_id(x).text for x in lst))
                    except FoldError:
                        raise FoldError("no such overload")
                raise FoldError("no such overload")
            if name == "spiffeMatchTrustDomain" and nv == 1:
                return SpiffeMatch("td", (vals[0] if isinstance(vals[0], SpiffeDomain) else _spiffe_domain(_need(vals[0], str))).name)
        if method and isinstance(vals[0], SpiffeId):
            if name == "isMemberOf" and nv == 2:
                if not isinstance(vals[1], SpiffeDomain):
                    raise FoldError("no such overload")
                return vals[0].domain == vals[1].name
            if name == "path" and nv == 1:
                return vals[0].text[vals[0].pathidx:]
            if name == "trustDomain" and nv == 1:
                return SpiffeDomain(vals[0].domain)
            raise FoldError("no such overload")
        if method and isinstance(vals[0], SpiffeDomain):
            if name == "name" and nv == 1:
                return vals[0].name
            if name == "id" and nv == 1:
                return "spiffe://" + vals[0].name
            raise FoldError("no such overload")
        if method and isinstance(vals[0], SpiffeMatch):
            if name != "matchesID" or nv != 2:
                raise FoldError("no such overload")
            sid = vals[1] if isinstance(vals[1], SpiffeId) else _spiffe_id(_need(vals[1], str))
            mt = vals[0]
            return mt.kind == "any" or (mt.kind == "exact" and sid.text == mt.arg) or (mt.kind == "oneof" and sid.text in mt.arg) or \
                (mt.kind == "td" and sid.domain == mt.arg)

        # hierarchy (types/hierarchy.go)
        if name == "hierarchy" and not method and nv in (1, 2):
            return _hierarchy(vals[0], vals[1] if nv == 2 else ".")
        if method and isinstance(vals[0], Hier) and nv == 2 and name in _HIER:
            if not isinstance(vals[1], Hier):
                raise FoldError("no such overload")
            return _HIER[name](vals[0], vals[1])

        # network
        if name == "isIP" and not method and nv in (1, 2):
            a = _parse_ip(_need(vals[0], str))
            return a is not None and (nv == 1 or a.version == _need(vals[1], int))
        if name == "ip" and not method and nv == 1:
            a = _parse_ip(_need(vals[0], str))
            if a is None:
                raise FoldError("IP Address parse error")
            return IPAddr(a)
        if name == "isCIDR" and not method and nv == 1:
            return _parse_cidr(_need(vals[0], str)) is not None
        if name == "cidr" and not method and nv == 1:
            c = _parse_cidr(_need(vals[0], str))
            if c is None:
                raise FoldError("network address parse error")
            return c
        if method and isinstance(vals[0], IPAddr) and nv == 1:
            return _ip_method(name, vals[0].a)
        if method and isinstance(vals[0], CIDR):
            c = vals[0]
            width = 32 if c.addr.version == 4 else 128
            if name == "containsIP" and nv == 2:
                a = vals[1].a if isinstance(vals[1], IPAddr) else _parse_ip(_need(vals[1], str))
                if a is None:
                    raise FoldError("IP Address parse error")
                return _cidr_contains_ip(c, a)
            if name == "containsCIDR" and nv == 2:
                o = vals[1] if isinstance(vals[1], CIDR) else _parse_cidr(_need(vals[1], str))
                if o is None:
                    raise FoldError("network address parse error")
                return o.bits >= c.bits and _cidr_contains_ip(c, o.addr)
            if name == "prefixLength" and nv == 1:
                return c.bits
            if name == "isMask" and nv == 1 or name == "masked" and nv == 1:
                shift = width - c.bits
                masked = (int(c.addr) >> shift) << shift
                if name == "isMask":
                    return masked == int(c.addr)
                return CIDR(type(c.addr)(masked), c.bits)
            if name == "ip" and nv == 1:
                return IPAddr(c.addr)
        if method and isinstance(vals[0], Opt):
            o = vals[0]
            if name == "hasValue" and nv == 1:
                return o.present
            if name == "value" and nv == 1:
                if not o.present:
                    raise FoldError("optional.none() dereference")
                return o.value
            if name == "orValue" and nv == 2:
                return o.value if o.present else vals[1]
        raise NotConst("function %s/%d" % (name, nv))


_HIER = {"ancestorOf": _h_ancestor, "descendentOf": lambda h, o: _h_ancestor(o, h), "immediateParentOf": _h_imm_parent,
         "immediateChildOf": lambda h, o: _h_imm_parent(o, h), "siblingOf": _h_sibling, "overlaps": _h_overlaps,
         "commonAncestors": _h_common}
_GO_SPACE = "\t\n\v\f\r \x85\xa0\u1680\u2000\u2001\u2002\u2003\u2004\u2005\u2006\u2007\u2008\u2009\u200a\u2028\u2029\u202f\u205f\u3000"


def _parse_ip_loose(s):   # net.ParseIP (inIPAddrRange, cerbos_lib.go:513-525): IPv4-mapped forms are accepted
    if "%" in s:
        return None
    try:
        a = ipaddress.ip_address(s)
    except ValueError:
        return None
    if isinstance(a, ipaddress.IPv4Address) and any(len(p) > 1 and p[0] == "0" for p in s.split(".")):
        return None
    if isinstance(a, ipaddress.IPv6Address) and a.ipv4_mapped is not None:
        return a.ipv4_mapped
    return a


def _parse_cidr_loose(s):
    if s.count("/") != 1:
        return None
    addr, bits = s.split("/")
    a = _parse_ip_loose(addr)
    if a is None or not bits.isdigit() or (len(bits) > 1 and bits[0] == "0") or int(bits) > (32 if a.version == 4 else 128):
        return None
    return CIDR(a, int(bits))


def _mkmap(pairs):
    """A CEL map from (key, value) pairs.  CEL keys keep their type (1, 1u, true and "1" are four keys) while a Python
    dict would merge 1 / True / 1.0: where such keys would meet the folder declines; a repeated key is a CEL error."""
    out = {}
    for k, v in pairs:
        for o in out:
            if type(o) is type(k) and o == k:
                raise FoldError("duplicate map key")
            if o == k:
                raise NotConst("map keys of mixed types")
        out[k] = v
    return out


def _sorted(items, key=lambda x: x):
    ks = [key(x) for x in items]
    if not ks:
        return []
    kinds = {("num" if _is_num(k) else type(k).__name__) for k in ks}
    if len(kinds) != 1 or not (kinds <= {"num", "bool", "str", "bytes"}):
        raise FoldError("list elements must be comparable")
    if any(isinstance(k, float) and math.isnan(k) for k in ks):
        raise NotConst("NaN in sort")
    import functools
    order = sorted(range(len(items)), key=functools.cmp_to_key(lambda i, j: compare(ks[i], ks[j]) or (i - j)))
    return [items[i] for i in order]


def _flatten(lst, depth):
    out = []
    for x in lst:
        if isinstance(x, list) and depth > 0:
            out.extend(_flatten(x, depth - 1))
        else:
            out.append(x)
    return out


# ---- AST in, AST out
def to_ast(v):
    """A folder value as a literal AST, or None when the device's constant pool has no form for it."""
    if v is None:
        return ("lit", "null", None)
    if isinstance(v, bool):
        return ("lit", "bool", v)
    if isinstance(v, UInt):
        return ("lit", "uint", int(v))
    if isinstance(v, int):
        return ("lit", "int", v)
    if isinstance(v, float):
        return ("lit", "double", v)
    if isinstance(v, str):
        return ("lit", "string", v)
    if isinstance(v, list):
        elems = [to_ast(x) for x in v]
        return None if any(e is None for e in elems) else ("list", tuple(elems))
    if isinstance(v, dict):
        ents = []
        for k, x in v.items():
            if not isinstance(k, str):
                return None
            e = to_ast(x)
            if e is None:
                return None
            ents.append((("lit", "string", k), e))
        return ("map", tuple(ents))
    return None


_FOLDABLE = ("list", "map", "not", "neg", "and", "or", "tern", "bin", "index", "select", "has", "bind", "comp", "call")


def _const_node(n):
    k = n[0]
    if k == "lit":
        return True
    if k == "list":
        return all(_const_node(e) for e in n[1])
    if k == "map":
        return all(_const_node(a) and _const_node(b) for a, b in n[1])
    return False


def _children(n):
    k = n[0]
    if k in ("select", "has", "not", "neg"):
        return [n[1]]
    if k == "index":
        return [n[1], n[2]]
    if k == "call":
        return ([n[2]] if n[2] is not None else []) + list(n[3])
    if k == "list":
        return list(n[1])
    if k == "map":
        return [x for pair in n[1] for x in pair]
    if k == "bin":
        return [n[2], n[3]]
    if k in ("and", "or"):
        return [n[1], n[2]]
    if k == "tern":
        return [n[1], n[2], n[3]]
    if k == "bind":
        return [n[2], n[3]]
    if k == "comp":
        return [n[2]] + list(n[4])
    return []


def _rebuild(n, ch):
    k = n[0]
    it = iter(ch)
    if k in ("select", "has"):
        return (k, next(it), n[2])
    if k in ("not", "neg"):
        return (k, next(it))
    if k == "index":
        return (k, next(it), next(it))
    if k == "call":
        tgt = next(it) if n[2] is not None else None
        return (k, n[1], tgt, tuple(it))
    if k == "list":
        return (k, tuple(it))
    if k == "map":
        flat = list(it)
        return (k, tuple((flat[i], flat[i + 1]) for i in range(0, len(flat), 2)))
    if k == "bin":
        return (k, n[1], next(it), next(it))
    if k in ("and", "or"):
        return (k, next(it), next(it))
    if k == "tern":
        return (k, next(it), next(it), next(it))
    if k == "bind":
        return (k, n[1], next(it), next(it))
    if k == "comp":
        return (k, n[1], next(it), n[3], tuple(it))
    return n


def _namespace_ident(n, parent_is_call_target):
    return parent_is_call_target and n[0] == "ident"


def fold(ast):
    """`ast` with every constant sub-expression the folder can compute replaced by its value."""
    def walk(n, bound):
        k = n[0]
        if k in ("lit", "ident") or k not in _FOLDABLE:
            return n
        kids = _children(n)
        if k == "comp":
            inner = bound | set(n[3])
            new = [walk(kids[0], bound)] + [walk(c, inner) for c in kids[1:]]
        elif k == "bind":
            new = [walk(kids[0], bound), walk(kids[1], bound | {n[1]})]
        else:
            new = [walk(c, bound) for c in kids]
        n2 = _rebuild(n, new)
        if k in ("list", "map") and _const_node(n2) and not _has_folded(n2):
            return n2   # already a literal
        try:
            v = _Eval().ev(n2, {})
        except (NotConst, FoldError, RecursionError, OverflowError, ValueError):
            return n2
        return ("lit", "folded", v, n2)

    def settle(n):
        """Folded values the constant pool can hold become literals; the others give way to the expression they came from."""
        if n[0] == "lit":
            if n[1] != "folded":
                return n
            lit = to_ast(n[2])
            return lit if lit is not None else settle_children(n[3])
        return settle_children(n)

    def settle_children(n):
        kids = _children(n)
        return _rebuild(n, [settle(c) for c in kids]) if kids else n

    return settle(walk(ast, frozenset()))


def _has_folded(n):
    if n[0] == "lit":
        return n[1] == "folded"
    return any(_has_folded(c) for c in _children(n))
