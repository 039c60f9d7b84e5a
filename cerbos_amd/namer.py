"""Policy naming helpers: FQNs, policy keys, scope parents, sanitisation.

Follows the naming rules of the reference's ``internal/namer/namer.go``
(file:line cited per function) so that the ``policy`` strings in a
``CheckOutput`` are byte-identical.
"""
from __future__ import annotations

import re

FQN_PREFIX = "cerbos."
DERIVED_ROLES_PREFIX = FQN_PREFIX + "derived_roles"
EXPORT_CONSTANTS_PREFIX = FQN_PREFIX + "export_constants"
EXPORT_VARIABLES_PREFIX = FQN_PREFIX + "export_variables"
PRINCIPAL_POLICIES_PREFIX = FQN_PREFIX + "principal"
RESOURCE_POLICIES_PREFIX = FQN_PREFIX + "resource"
ROLE_POLICIES_PREFIX = FQN_PREFIX + "role"
DEFAULT_VERSION = "default"

# namer.go:18-20 (RE2 classes are ASCII-only)
_INVALID_IDENT_CHARS = re.compile(r"[^\w.]+", re.ASCII)
_OLD_NAME_PATTERN = re.compile(
    r"^[A-Za-z][0-9A-Za-z_@.\-/]*(:[A-Za-z][0-9A-Za-z_@.\-/]*)*\Z"   # \Z: Go's `$` is end of text, never "before a final newline"
)


def sanitize(v: str) -> str:
    """namer.go:213-218 - only names matching the pre-0.30 pattern are rewritten."""
    if _OLD_NAME_PATTERN.match(v):
        return _INVALID_IDENT_CHARS.sub("_", v)
    return v


def with_scope(fqn: str, scope: str) -> str:
    return fqn if scope == "" else fqn + "/" + scope


def resource_policy_fqn(resource: str, version: str, scope: str) -> str:
    """namer.go:109-112"""
    return with_scope(
        f"{RESOURCE_POLICIES_PREFIX}.{sanitize(resource)}.v{sanitize(version)}", scope
    )


def principal_policy_fqn(principal: str, version: str, scope: str) -> str:
    """namer.go:131-134"""
    return with_scope(
        f"{PRINCIPAL_POLICIES_PREFIX}.{sanitize(principal)}.v{sanitize(version)}", scope
    )


def role_policy_fqn(role: str, version: str, scope: str) -> str:
    """namer.go:143-149 (empty version means "default")"""
    if version == "":
        version = DEFAULT_VERSION
    return with_scope(f"{ROLE_POLICIES_PREFIX}.{sanitize(role)}.v{sanitize(version)}", scope)


def derived_roles_fqn(name: str) -> str:
    return f"{DERIVED_ROLES_PREFIX}.{sanitize(name)}"


def export_constants_fqn(name: str) -> str:
    return f"{EXPORT_CONSTANTS_PREFIX}.{sanitize(name)}"


def export_variables_fqn(name: str) -> str:
    return f"{EXPORT_VARIABLES_PREFIX}.{sanitize(name)}"


def policy_key_from_fqn(fqn: str) -> str:
    """namer.go:95-97"""
    return fqn[len(FQN_PREFIX):] if fqn.startswith(FQN_PREFIX) else fqn


def scope_parents(scope: str):
    """namer.go:77-87 - "a.b.c" -> "a.b", "a", "" ; "" -> nothing."""
    for i in range(len(scope) - 1, -1, -1):
        if scope[i] == "." or i == 0:
            yield scope[:i]


def scope_value(scope: str) -> str:
    """namer.go:276-278 - strip one leading dot."""
    return scope[1:] if scope.startswith(".") else scope


def resource_rule_name(name: str, idx: int) -> str:
    """namer_non_embedded.go:106-112 (idx is 1-based)."""
    return name if name else f"rule-{idx:03d}"


def principal_resource_action_rule_name(name: str, resource: str, idx: int) -> str:
    """namer_non_embedded.go:115-121"""
    return name if name else f"{resource}_rule-{idx:03d}"


def rule_fqn(kind: str, name: str, version: str, scope: str, rule_name: str) -> str:
    """namer.go:221-245; ``kind`` is one of "resource" | "principal" | "role"."""
    if kind == "resource":
        fqn = resource_policy_fqn(name, version, scope)
    elif kind == "principal":
        fqn = principal_policy_fqn(name, version, scope)
    else:
        fqn = role_policy_fqn(name, version, scope)
    return f"{policy_key_from_fqn(fqn)}#{rule_name}"
