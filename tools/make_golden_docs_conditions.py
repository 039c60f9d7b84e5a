"""Mines the reference's published condition examples: docs/modules/policies/pages/conditions.adoc holds, per function family
(durations, hierarchies, IP addresses, lists / maps / sets, math, paths, SPIFFE, strings, timestamps), a "Test data" request
fragment and a table whose third column is an example expression written against that fragment - most of them statements the
documentation presents as true.  -> tests/golden/docs_condition_examples.json: [{section, line, function, expr, request}].

    python tools/make_golden_docs_conditions.py        (needs /root/reference; the GPU box never runs this)"""
import json
import os
import re

REF = "/root/reference/docs/modules/policies/pages/conditions.adoc"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "docs_condition_examples.json")


def _unescape(cell):
    cell = cell.replace("\\|", "|").replace("\\<", "<").replace("+*+", "*")
    cell = re.sub(r"\s+\+\s*\n\s*", " ", cell)     # a hard line break inside a cell
    return " ".join(cell.split())


def _split_row(row):
    """Cells of one table row: separated by ' | ' where the bar is not escaped."""
    cells, cur, i = [], "", 0
    row = row[1:]   # the leading bar
    while i < len(row):
        if row[i] == "\\" and i + 1 < len(row) and row[i + 1] == "|":
            cur += "\\|"
            i += 2
        elif row[i] == "|":
            cells.append(cur)
            cur = ""
            i += 1
        else:
            cur += row[i]
            i += 1
    cells.append(cur)
    return [c.strip() for c in cells]


def main():
    lines = open(REF, encoding="utf-8").read().split("\n")
    section, request, out = None, None, []
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("== "):
            section, request = ln[3:].strip(), None
        elif ln.startswith("[source,json") and i > 0 and lines[i - 1].strip() == ".Test data":
            j = i + 2
            body = []
            while lines[j].strip() != "----":
                if lines[j].strip() != "...":
                    body.append(lines[j])
                j += 1
            text = re.sub(r",(\s*[}\]])", r"\1", "{" + "\n".join(body) + "}")   # the fragments carry trailing commas
            request = json.loads(text)
            i = j
        elif ln.strip() == "|===" and request is not None:
            j = i + 1
            rows, cur, start = [], None, None
            while lines[j].strip() != "|===":
                if lines[j].startswith("|"):
                    if cur is not None:
                        rows.append((start, cur))
                    cur, start = lines[j], j + 1
                elif cur is not None:
                    cur += "\n" + lines[j]
                j += 1
            if cur is not None:
                rows.append((start, cur))
            header = _split_row(rows[0][1]) if rows else []
            if len(header) == 3 and header[2] == "Example":
                for at, row in rows[1:]:
                    cells = _split_row(row)
                    if len(cells) != 3:
                        raise SystemExit("conditions.adoc:%d: %d cells" % (at, len(cells)))
                    out.append({"section": section, "line": at, "function": _unescape(cells[0]), "expr": _unescape(cells[2]), "request": request})
            i = j
        i += 1
    # sections without a request fragment (Math, Paths): constant expressions
    section, i = None, 0
    seen = {(e["section"], e["line"]) for e in out}
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("== "):
            section = ln[3:].strip()
        elif ln.strip() == "|===" and section in ("Math", "Paths"):
            j = i + 1
            while lines[j].strip() != "|===":
                if lines[j].startswith("|") and not lines[j].startswith("| Function"):
                    cells = _split_row(lines[j])
                    if len(cells) == 3 and (section, j + 1) not in seen:
                        out.append({"section": section, "line": j + 1, "function": _unescape(cells[0]), "expr": _unescape(cells[2]), "request": {}})
                j += 1
            i = j
        i += 1
    out.sort(key=lambda e: e["line"])
    doc = {"source": "docs/modules/policies/pages/conditions.adoc: the example column of the function tables with the section's test data", "examples": out}
    json.dump(doc, open(OUT, "w"), indent=1, ensure_ascii=False)
    print(len(out), "examples;", sorted({e["section"] for e in out}))


if __name__ == "__main__":
    main()
