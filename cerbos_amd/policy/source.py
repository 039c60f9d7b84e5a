"""Where the values of a policy document sit in its YAML text, for the compiler's error reports.

The reference's parser (internal/parser, built on goccy/go-yaml) answers "position of the value at this path" for the compile
errors (internal/compile/context.go:65-85); the conventions the compiler's fixtures show (internal/test/testdata/compile/*.yaml):

* a field of a message - the position of its KEY (`expr: ...` -> the `e`);
* an entry of a map field (variables.local, definitions ...) - the position of its VALUE; an error about the key itself - the KEY;
* an element of a list - the element; when it is a mapping, the `:` after its first key (go-yaml's mapping token).

Lines and columns are 1-based.  Build-time only."""
from __future__ import annotations

import yaml

from .loader import _Loader


class Source:
    def __init__(self, file: str, marks: dict | None = None):
        self.file = file
        self.marks = marks or {}   # path tuple -> {"key": (l, c), "value": (l, c), "colon": (l, c)}

    def position(self, path: tuple, at: str):
        return (self.marks.get(tuple(path)) or {}).get(at)


def _pos(mark):
    return (mark.line + 1, mark.column + 1)


def _walk(node, path, marks):
    if isinstance(node, yaml.MappingNode):
        for key_node, value_node in node.value:
            if not isinstance(key_node, yaml.ScalarNode):
                continue
            p = path + (key_node.value,)
            m = marks.setdefault(p, {})
            m["key"], m["value"] = _pos(key_node.start_mark), _pos(value_node.start_mark)
            _walk(value_node, p, marks)
    elif isinstance(node, yaml.SequenceNode):
        for i, item in enumerate(node.value):
            p = path + (i,)
            m = marks.setdefault(p, {})
            m["value"] = _pos(item.start_mark)
            if isinstance(item, yaml.MappingNode) and item.value and isinstance(item.value[0][0], yaml.ScalarNode):
                end = item.value[0][0].end_mark
                m["colon"] = (end.line + 1, end.column + 1)
            _walk(item, p, marks)


def load_yaml_with_source(text: str, file: str):
    """The first non-empty document of `text` and the positions of its values -> (dict, Source)."""
    loader = _Loader(text)
    try:
        while loader.check_node():
            node = loader.get_node()
            doc = loader.construct_document(node)
            if doc is not None:
                marks = {}
                _walk(node, (), marks)
                return doc, Source(file, marks)
    finally:
        loader.dispose()
    return None, Source(file)


def json_path(path) -> str:
    """`$.resourcePolicy.rules[0].condition.match.expr`; a key that holds a dot is quoted."""
    out = ["$"]
    for p in path:
        if isinstance(p, int):
            out.append("[%d]" % p)
        elif "." in p:
            out.append(".'%s'" % p)
        else:
            out.append("." + p)
    return "".join(out)
