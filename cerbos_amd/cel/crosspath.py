"""The file-path functions of the Cerbos CEL library - basePath, dirPath, extPath, joinPath, pathHasPrefix, pathMatch,
pathMatchAnyOf, relPath, volumeName (internal/conditions/cerbos_lib.go:138-236, 527-553) - for the lowering's constant
folder (cel/fold.py).  The reference computes them with internal/conditions/crosspath/crosspath.go: a UNIX, UNC
(`\\\\host\\share\\x`) or Win32 (`C:\\x`, `a\\b`) path becomes a slash-separated one, Go's path/filepath (linux build) works
on that, and the answer goes back to the notation it came in.

Here a path is (flavour, segments-or-text); path/filepath's lexical rules are written over segment lists, its Match over a
pattern compiled once into chunks.  Held against the reference's own tables (tests/golden/crosspath_vectors.json) and against
the checker's separate restatement (oracle/crosspath.py) on generated paths: tests/test_crosspath.py."""
from collections import namedtuple

SLASH, BACK = "/", "\\"


class PathError(Exception):
    """The reference returns a Go error here (a CEL error once it reaches the expression)."""


# ---- path/filepath on slash-separated text -----------------------------------------------------------------------------------
def fp_clean(text):
    rooted = text.startswith(SLASH)
    kept = []
    for seg in text.split(SLASH):
        if seg == "" or seg == ".":
            continue
        if seg != "..":
            kept.append(seg)
        elif kept and kept[-1] != "..":
            del kept[-1]
        elif not rooted:
            kept.append(seg)
    if rooted:
        return SLASH + SLASH.join(kept)
    return SLASH.join(kept) if kept else "."


def fp_base(text):
    if not text:
        return "."
    stripped = text.rstrip(SLASH)
    if not stripped:
        return SLASH
    return stripped.rsplit(SLASH, 1)[-1]


def fp_dir(text):
    head, sep, _ = text.rpartition(SLASH)
    return fp_clean(head + sep)


def fp_ext(text):
    last = text.rsplit(SLASH, 1)[-1]
    dot = last.rfind(".")
    return last[dot:] if dot >= 0 else ""


def fp_join(*elems):
    elems = list(elems)
    while elems and elems[0] == "":
        del elems[0]
    return fp_clean(SLASH.join(elems)) if elems else ""


def _segments(clean, is_base):
    """A cleaned path as (rooted, segments): "/" has none, and neither has "." as the base (as the target it stays a segment -
    filepath.Rel("a", ".") is "../.")."""
    rooted = clean.startswith(SLASH)
    body = clean[1:] if rooted else clean
    return rooted, ([] if body == "" or (is_base and body == ".") else body.split(SLASH))


def fp_rel(base, target):
    cb, ct = fp_clean(base), fp_clean(target)
    if cb == ct:
        return "."
    (broot, bsegs), (troot, tsegs) = _segments(cb, True), _segments(ct, False)
    if broot != troot:
        raise PathError("Rel: can't make %s relative to %s" % (target, base))
    same = 0
    while same < len(bsegs) and same < len(tsegs) and bsegs[same] == tsegs[same]:
        same += 1
    if same < len(bsegs) and bsegs[same] == "..":
        raise PathError("Rel: can't make %s relative to %s" % (target, base))
    return SLASH.join([".."] * (len(bsegs) - same) + tsegs[same:])


# Match: the pattern is cut at every * outside brackets; each chunk is a list of single-character tests
_LIT, _ONE, _SET = 0, 1, 2
_BAD = "syntax error in pattern"


def _cut(pattern):
    """[(after a star?, chunk text)] - brackets are tracked the way filepath's scanner does: `[` opens, `]` closes, \\x is skipped."""
    out, i, n = [], 0, len(pattern)
    while i < n:
        star = False
        while i < n and pattern[i] == "*":
            star, i = True, i + 1
        j, bracket = i, False
        while j < n and (bracket or pattern[j] != "*"):
            c = pattern[j]
            if c == BACK and j + 1 < n:
                j += 1
            elif c == "[":
                bracket = True
            elif c == "]":
                bracket = False
            j += 1
        out.append((star, pattern[i:j]))
        i = j
    return out


def _class_char(chunk, k):
    """One end of a range inside brackets -> (character, next index); something must follow it."""
    if k >= len(chunk) or chunk[k] in "-]":
        raise PathError(_BAD)
    if chunk[k] == BACK:
        k += 1
        if k >= len(chunk):
            raise PathError(_BAD)
    if k + 1 >= len(chunk):
        raise PathError(_BAD)
    return chunk[k], k + 1


def _tests(chunk):
    out, k, n = [], 0, len(chunk)
    while k < n:
        c = chunk[k]
        if c == "?":
            out.append((_ONE,))
            k += 1
        elif c == "[":
            k += 1
            negated = k < n and chunk[k] == "^"
            if negated:
                k += 1
            ranges = []
            while not (ranges and k < n and chunk[k] == "]"):
                lo, k = _class_char(chunk, k)
                hi = lo
                if chunk[k] == "-":
                    hi, k = _class_char(chunk, k + 1)
                ranges.append((lo, hi))
            k += 1
            out.append((_SET, negated, ranges))
        else:
            if c == BACK:
                k += 1
                if k >= n:
                    raise PathError(_BAD)
            out.append((_LIT, chunk[k]))
            k += 1
    return out


def _chunk_at(tests, name, at):
    """The chunk's tests against name[at:] -> the index after them, or -1."""
    if at + len(tests) > len(name):
        return -1
    for t in tests:
        ch = name[at]
        if t[0] == _LIT:
            if ch != t[1]:
                return -1
        elif t[0] == _ONE:
            if ch == SLASH:
                return -1
        elif any(lo <= ch <= hi for lo, hi in t[2]) == t[1]:
            return -1
        at += 1
    return at


def fp_match(pattern, name):
    chunks = [(star, _tests(text)) for star, text in _cut(pattern)]    # a malformed pattern is an error whatever the name
    at = 0
    for k, (star, tests) in enumerate(chunks):
        last = k == len(chunks) - 1
        if star and not tests:
            return SLASH not in name[at:]
        start = at
        while True:
            end = _chunk_at(tests, name, start)
            if end >= 0 and (not last or end == len(name)):
                break
            if not star or start >= len(name) or name[start] == SLASH:   # a star never takes a separator
                return False
            start += 1
        at = end
    return at == len(name)


# ---- crosspath ---------------------------------------------------------------------------------------------------------------
Coded = namedtuple("Coded", "text flavour backslashes root")      # flavour: "unc" | "drive" | ""


def _drive(path):
    return len(path) >= 2 and path[1] == ":" and ("a" <= path[0] <= "z" or "A" <= path[0] <= "Z")


def encode(path):
    backslashes = BACK in path
    slashed = path.replace(BACK, SLASH)
    if path.startswith(BACK * 2):
        # \\host\share[\...]: the root is exactly host and share
        return Coded(fp_clean(slashed), "unc", backslashes, path[2:].count(BACK) == 1)
    if _drive(path):
        if not backslashes and len(path) > 2:
            raise PathError("unsupported Win32 path")     # C:. / C:dir: relative to the drive's current directory
        return Coded(fp_clean(SLASH + slashed), "drive", backslashes, len(path) <= 3)
    return Coded(fp_clean(slashed), "", backslashes, path == SLASH)


def decode(coded):
    if coded.flavour == "unc":
        return BACK + coded.text.replace(SLASH, BACK)
    if coded.flavour == "drive":
        text = coded.text.replace(SLASH, BACK)
        return text[1:] if text[:1] == BACK else text
    return coded.text.replace(SLASH, BACK) if coded.backslashes else coded.text


def _encoded(path, what="path"):
    try:
        return encode(path)
    except PathError as err:
        raise PathError("failed to encode %s %s: %s" % (what, path, err))


def base(path):
    return fp_base(_encoded(path).text)


def ext(path):
    return fp_ext(_encoded(path).text)


def dir_(path):
    coded = _encoded(path)
    if coded.flavour or coded.backslashes:
        if coded.root and coded.flavour:
            return decode(coded) + (BACK if coded.flavour == "drive" else "")
        head, sep, _ = coded.text.rpartition(SLASH)
        if not sep:
            raise PathError("no separator in %s" % path)    # the reference indexes out of range here (a Go panic)
        return decode(coded._replace(text=head))
    return decode(coded._replace(text=fp_dir(coded.text)))


def join(paths):
    if len(paths) < 2:
        return paths[0] if paths else ""
    first = _encoded(paths[0], "first path")
    text = first.text
    for p in paths[1:]:
        text = fp_join(text, _encoded(p).text)
    return decode(first._replace(text=text))


def match(path, pattern):
    name, pat = _encoded(path), _encoded(pattern, "pattern")
    try:
        return fp_match(pat.text, name.text)
    except PathError as err:
        raise PathError("failed to match pattern %r on path %r: %s" % (pattern, path, err))


def rel(base_path, target_path):
    b, t = _encoded(base_path, "base path"), _encoded(target_path, "target path")
    try:
        text = fp_rel(b.text, t.text)
    except PathError as err:
        raise PathError("failed to determine relative path of %s: %s" % (target_path, err))
    if text in (".", ".."):
        return text
    if t.flavour == "unc":
        return text.replace(SLASH, BACK)
    return decode(t._replace(text=text))


def volume_name(path):
    if path.startswith(BACK * 2):
        parts = path[2:].split(BACK, 2)
        return BACK * 2 + parts[0] + BACK + parts[1] if len(parts) > 1 and parts[0] and parts[1] else ""
    return path[:2] if _drive(path) else ""


def has_prefix(path, prefix):
    if path == prefix:
        return True
    r = rel(prefix, path)
    return r != "" and r[0] != "." and r[0] != SLASH


def match_any_of(path, patterns):
    return any(match(path, p) for p in patterns)
