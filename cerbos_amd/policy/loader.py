"""Policy documents (YAML/JSON) -> plain dicts keyed by FQN.

Build-time feeder only (the reference's ``internal/parser`` / ``internal/policy`` /
``internal/storage/index`` equivalents are control plane and out of scope; this is the
minimum needed to get from a policy file to rule-table rows).

Store walking rules follow ``internal/util/filesystem.go:21-66,139-150`` (skip hidden
files/dirs, ``testdata`` dirs, ``*_test.yaml``) and ``internal/storage/index/builder.go:142-143``
(skip ``disabled: true``). Schema directories (``_schemas``) are ignored.
"""
from __future__ import annotations

import os

import yaml

from .. import namer

_EXTS = (".yaml", ".yml", ".json")


class _Loader(yaml.SafeLoader):
    """SafeLoader that keeps timestamps as strings (policy text never contains dates)."""


_Loader.yaml_implicit_resolvers = {
    k: [(tag, rx) for tag, rx in v if tag != "tag:yaml.org,2002:timestamp"]
    for k, v in yaml.SafeLoader.yaml_implicit_resolvers.items()
}


def load_yaml_documents(text: str):
    return [d for d in yaml.load_all(text, Loader=_Loader) if d is not None]


def load_yaml(text: str):
    docs = load_yaml_documents(text)
    return docs[0] if docs else None


def policy_kind(p: dict) -> str:
    for k in ("resourcePolicy", "principalPolicy", "rolePolicy", "derivedRoles",
              "exportVariables", "exportConstants"):
        if k in p:
            return k
    raise ValueError("unknown policy type: keys=%s" % sorted(p))


def policy_fqn(p: dict) -> str:
    """namer_non_embedded.go FQN(): the FQN of a policy document."""
    k = policy_kind(p)
    b = p[k]
    if k == "resourcePolicy":
        return namer.resource_policy_fqn(b["resource"], str(b.get("version", "")), b.get("scope", "") or "")
    if k == "principalPolicy":
        return namer.principal_policy_fqn(b["principal"], str(b.get("version", "")), b.get("scope", "") or "")
    if k == "rolePolicy":
        return namer.role_policy_fqn(b["role"], str(b.get("version", "") or ""), b.get("scope", "") or "")
    if k == "derivedRoles":
        return namer.derived_roles_fqn(b["name"])
    if k == "exportVariables":
        return namer.export_variables_fqn(b["name"])
    return namer.export_constants_fqn(b["name"])


def _is_hidden(name: str) -> bool:
    return name.startswith(".")


def load_policy_dir(root: str) -> dict:
    """Walk ``root`` and return {fqn: policy dict} in deterministic (path-sorted) order."""
    return load_policy_dir_with_sources(root)[0]


def load_policy_dir_with_sources(root: str):
    """({fqn: policy dict}, {fqn: policy.source.Source}): the policies of a directory and where their values sit in the files
    (paths relative to ``root``), for the compiler's error reports."""
    from .source import load_yaml_with_source
    out, sources = {}, {}
    for dirpath, dirnames, filenames in os.walk(root):
        dirnames[:] = sorted(
            d for d in dirnames if not _is_hidden(d) and d != "testdata" and d != "_schemas"
        )
        for fn in sorted(filenames):
            if _is_hidden(fn) or not fn.endswith(_EXTS):
                continue
            stem = os.path.splitext(fn)[0]
            if stem.endswith("_test"):
                continue
            with open(os.path.join(dirpath, fn), "r", encoding="utf-8") as f:
                doc, source = load_yaml_with_source(f.read(), os.path.relpath(os.path.join(dirpath, fn), root))
            if not isinstance(doc, dict) or "apiVersion" not in doc:
                continue
            if doc.get("disabled"):
                continue
            out[policy_fqn(doc)] = doc
            sources[policy_fqn(doc)] = source
    return out, sources


def policies_from_docs(docs) -> dict:
    """{fqn: policy} from an iterable of already-parsed policy dicts."""
    out = {}
    for d in docs:
        if d.get("disabled"):
            continue
        out[policy_fqn(d)] = d
    return out
