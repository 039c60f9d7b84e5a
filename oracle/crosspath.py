"""TEST INFRASTRUCTURE (oracle): the reference's file-path helpers restated - internal/conditions/crosspath/crosspath.go, which the
CEL functions basePath / dirPath / extPath / joinPath / pathHasPrefix / pathMatch / pathMatchAnyOf / relPath / volumeName
(cerbos_lib.go:138-236, 527-553) are bound to.  crosspath turns a UNIX, UNC or Win32 path into one slash-separated form, runs
Go's path/filepath (the UNIX flavour: the server is built for linux) on it and turns the answer back.  path/filepath is the Go
standard library (not in /root/reference): its lexical algorithms - Clean, Base, Dir, Ext, Join, Rel, Match - are restated
below from their documented behaviour.

Pinned by tests/golden/crosspath_vectors.json (the 104 vectors of crosspath_test.go) and the 35 path expressions of
TestCerbosLib (tests/golden/cerbos_lib_kats.json).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module."""

UNKNOWN, DRIVE, UNC = 0, 1, 2


class PathError(Exception):
    pass


# ---- Go path/filepath, GOOS=linux ------------------------------------------------------------------------------------------
def fp_clean(path):
    """filepath.Clean: the shortest equivalent path by lexical processing (rules 1-4 of its documentation)."""
    if path == "":
        return "."
    rooted = path[0] == "/"
    out = []
    for part in path.split("/"):
        if part in ("", "."):
            continue
        if part == "..":
            if out and out[-1] != "..":
                out.pop()            # rule 3: an inner .. removes the element before it
            elif not rooted:
                out.append("..")     # nothing to remove: it stays, unless the path is rooted (rule 4)
        else:
            out.append(part)
    text = "/".join(out)
    if rooted:
        return "/" + text
    return text or "."


def fp_base(path):
    if path == "":
        return "."
    path = path.rstrip("/")
    path = path[path.rfind("/") + 1:]
    return path or "/"


def fp_dir(path):
    return fp_clean(path[:path.rfind("/") + 1])


def fp_ext(path):
    i = len(path) - 1
    while i >= 0 and path[i] != "/":
        if path[i] == ".":
            return path[i:]
        i -= 1
    return ""


def fp_join(*elems):
    for i, e in enumerate(elems):
        if e != "":
            return fp_clean("/".join(elems[i:]))
    return ""


def fp_rel(base_path, targ_path):
    base, targ = fp_clean(base_path), fp_clean(targ_path)
    if targ == base:
        return "."
    if base == ".":
        base = ""
    if base.startswith("/") != targ.startswith("/"):
        raise PathError("Rel: can't make %s relative to %s" % (targ_path, base_path))
    bl, tl = len(base), len(targ)
    b0 = bi = t0 = ti = 0
    while True:                         # the first elements that differ (the paths are not equal: there are such)
        while bi < bl and base[bi] != "/":
            bi += 1
        while ti < tl and targ[ti] != "/":
            ti += 1
        if targ[t0:ti] != base[b0:bi]:
            break
        if bi < bl:
            bi += 1
        if ti < tl:
            ti += 1
        b0, t0 = bi, ti
    if base[b0:bi] == "..":
        raise PathError("Rel: can't make %s relative to %s" % (targ_path, base_path))
    if b0 != bl:                        # base elements left: up first, then down
        ups = [".."] * (1 + base[b0:bl].count("/"))
        if t0 != tl:
            ups.append(targ[t0:])
        return "/".join(ups)
    return targ[t0:]


class BadPattern(PathError):
    def __init__(self):
        PathError.__init__(self, "syntax error in pattern")


def _scan_chunk(pattern):
    star = False
    while pattern and pattern[0] == "*":
        pattern, star = pattern[1:], True
    in_range, i = False, 0
    while i < len(pattern):
        c = pattern[i]
        if c == "\\":
            if i + 1 < len(pattern):
                i += 1
        elif c == "[":
            in_range = True
        elif c == "]":
            in_range = False
        elif c == "*" and not in_range:
            break
        i += 1
    return star, pattern[:i], pattern[i:]


def _get_esc(chunk):
    if chunk == "" or chunk[0] in "-]":
        raise BadPattern()
    if chunk[0] == "\\":
        chunk = chunk[1:]
        if chunk == "":
            raise BadPattern()
    r, chunk = chunk[0], chunk[1:]
    if chunk == "":
        raise BadPattern()
    return r, chunk


def _match_chunk(chunk, s):
    """(rest of s, matched).  After the match fails the chunk is still read to its end, so that a malformed one is an error."""
    failed = False
    while chunk:
        if not failed and s == "":
            failed = True
        c = chunk[0]
        if c == "[":
            r = "\0"
            if not failed:
                r, s = s[0], s[1:]
            chunk = chunk[1:]
            negated = chunk[:1] == "^"
            if negated:
                chunk = chunk[1:]
            match, nrange = False, 0
            while True:
                if chunk[:1] == "]" and nrange > 0:
                    chunk = chunk[1:]
                    break
                lo, chunk = _get_esc(chunk)
                hi = lo
                if chunk[0] == "-":
                    hi, chunk = _get_esc(chunk[1:])
                if lo <= r <= hi:
                    match = True
                nrange += 1
            if match == negated:
                failed = True
        elif c == "?":
            if not failed:
                if s[0] == "/":
                    failed = True
                s = s[1:]
            chunk = chunk[1:]
        else:
            if c == "\\":
                chunk = chunk[1:]
                if chunk == "":
                    raise BadPattern()
            if not failed:
                if chunk[0] != s[0]:
                    failed = True
                s = s[1:]
            chunk = chunk[1:]
    return (None, False) if failed else (s, True)


def fp_match(pattern, name):
    """filepath.Match: * = any run of non-separators, ? = one non-separator, [..] a class, \\c = c."""
    while pattern:
        star, chunk, pattern = _scan_chunk(pattern)
        if star and chunk == "":
            return "/" not in name      # a trailing * takes the rest, if it has no separator
        t, ok = _match_chunk(chunk, name)
        if ok and (t == "" or pattern):
            name = t
            continue
        if star:
            found = False
            i = 0
            while i < len(name) and name[i] != "/":
                t, ok = _match_chunk(chunk, name[i + 1:])
                if ok and not (pattern == "" and t):
                    name, found = t, True
                    break
                i += 1
            if found:
                continue
        while pattern:                  # no match: the rest of the pattern must still be well formed
            _, chunk, pattern = _scan_chunk(pattern)
            _match_chunk(chunk, "")
        return False
    return name == ""


# ---- crosspath.go ----------------------------------------------------------------------------------------------------------
def _is_unc(path):                       # crosspath.go:247-249
    return path.startswith("\\\\")


def _is_drive(path):                     # crosspath.go:251-263
    return len(path) > 1 and path[0].isascii() and path[0].isalpha() and path[1] == ":"


class Encoded:
    __slots__ = ("value", "kind", "win32", "root")


def encode(path):
    """crosspath.go:31-73: `\\\\host\\share\\dir` -> /host/share/dir, `C:\\dir` -> /C:/dir, `a\\b` -> a/b, UNIX paths as they are; then Clean."""
    e = Encoded()
    e.win32, e.root = "\\" in path, False
    if _is_unc(path):
        e.kind, e.value = UNC, path.replace("\\", "/")
        if len(path[2:].split("\\", 2)) == 2:        # \\host\share
            e.root = True
    elif _is_drive(path):
        if not e.win32 and len(path) > 2:            # D:. or D:foo
            raise PathError("unsupported Win32 path")
        e.kind, e.value = DRIVE, ("\\" + path).replace("\\", "/")
        if len(path) in (2, 3):                      # D: or D:\
            e.root = True
    else:
        e.kind = UNKNOWN
        if e.win32:
            e.value = path.replace("\\", "/")
        else:
            e.value = path
            e.root = path == "/"
    e.value = fp_clean(e.value)
    return e


def decode(e):                           # crosspath.go:76-89
    if e.kind == UNC:
        return "\\" + e.value.replace("/", "\\")
    if e.kind == DRIVE:
        v = e.value.replace("/", "\\")
        return v[1:] if v.startswith("\\") else v
    if e.win32:
        return e.value.replace("/", "\\")
    return e.value


def _enc(path, what="path"):
    try:
        return encode(path)
    except PathError as x:
        raise PathError("failed to encode %s %s: %s" % (what, path, x))


def base(path):                          # crosspath.go:92-99
    return fp_base(_enc(path).value)


def _cut_last(e):
    i = e.value.rfind("/")
    if i < 0:
        raise PathError("slice bounds out of range")   # the reference slices value[:-1] here: a Go panic
    e.value = e.value[:i]
    return decode(e)


def dir_(path):                          # crosspath.go:102-135
    e = _enc(path)
    if e.kind == UNC:
        return decode(e) if e.root else _cut_last(e)
    if e.kind == DRIVE:
        return decode(e) + "\\" if e.root else _cut_last(e)
    if e.win32:
        return _cut_last(e)
    e.value = fp_dir(e.value)
    return decode(e)


def ext(path):                           # crosspath.go:138-145
    return fp_ext(_enc(path).value)


def join(paths):                         # crosspath.go:148-172
    if not paths:
        return ""
    if len(paths) == 1:
        return paths[0]
    result = _enc(paths[0], "first path")
    for p in paths[1:]:
        result.value = fp_join(result.value, _enc(p).value)
    return decode(result)


def match(path, pattern):                # crosspath.go:175-192
    p, q = _enc(path), _enc(pattern, "pattern")
    try:
        return fp_match(q.value, p.value)
    except PathError as x:
        raise PathError("failed to match pattern %r on path %r: %s" % (pattern, path, x))


def rel(base_path, target_path):         # crosspath.go:197-218
    b, t = _enc(base_path, "base path"), _enc(target_path, "target path")
    try:
        t.value = fp_rel(b.value, t.value)
    except PathError as x:
        raise PathError("failed to determine relative path of %s: %s" % (target_path, x))
    if t.value in (".", ".."):
        return t.value
    if t.kind == UNC and not t.value.startswith("\\"):
        return t.value.replace("/", "\\")
    return decode(t)


def volume_name(path):                   # crosspath.go:226-238
    if _is_unc(path):
        subs = path[2:].split("\\", 2)
        if len(subs) > 1 and subs[0] != "" and subs[1] != "":
            return "\\\\" + subs[0] + "\\" + subs[1]
    elif _is_drive(path):
        return path[:2]
    return ""


def has_prefix(path, prefix):            # cerbos_lib.go:527-538
    if prefix == path:
        return True
    r = rel(prefix, path)
    return len(r) > 0 and r[0] not in "./"


def match_any_of(path, patterns):        # cerbos_lib.go:540-553
    for p in patterns:
        if match(path, p):
            return True
    return False
