#!/usr/bin/env python3
"""Partial preprocessor: fixes a set of macros in a C/C++/HIP source and removes the conditional code they no longer select.

    python tools/resolve_ifs.py FILE --set TW_XA=0 --set TW_FLAGS=0 --undef TW_EXP_2D --undef TW_ABL_NOVM [--in-place]

* `#ifndef X / #define X v / #endif` default blocks of a fixed macro become the bare `#define X v` (the value stays usable in code);
* every #if / #ifdef / #ifndef / #elif whose condition is decided by the fixed (--set) and known-undefined (--undef) names is
  resolved: false branches go, a true branch ends the chain and loses its directive when nothing undecided stands in front of it;
* conditions that mention anything else (UVA_INSTRUMENT, ...) stay as they are.

Used in round 6 to take the rejected A/B variants and the wrong-result ceiling experiments out of csrc/uva_wino.hip.h
(VERDICT r5 item 5); the removed variants are kept as tools/experiments/*.patch."""
import argparse
import re
import sys

IDENT = re.compile(r"\b[A-Za-z_][A-Za-z0-9_]*\b")


def evaluate(expr, fixed, undef):
    """-> True / False / None (undecided)"""
    e = expr.split("//")[0].strip()
    e = re.sub(r"/\*.*?\*/", " ", e)

    def defined(m):
        n = m.group(1) or m.group(2)
        if n in fixed:
            return " 1 "
        if n in undef:
            return " 0 "
        return " __UNKNOWN__ "
    e = re.sub(r"defined\s*\(\s*([A-Za-z_]\w*)\s*\)|defined\s+([A-Za-z_]\w*)", defined, e)

    def ident(m):
        n = m.group(0)
        if n in fixed:
            return str(fixed[n])
        if n in undef:
            return "0"
        return n
    e = IDENT.sub(ident, e)
    if IDENT.search(e):
        return None
    e = e.replace("&&", " and ").replace("||", " or ")
    e = re.sub(r"!(?!=)", " not ", e)
    try:
        return bool(eval(e, {"__builtins__": {}}, {}))
    except Exception:  # noqa: BLE001
        return None


def directive(line):
    m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", line)
    return (m.group(1), m.group(2).strip()) if m else (None, None)


def parse(lines, i, fixed, undef):
    """-> (output lines, next index) for lines[i:] up to an unmatched #elif/#else/#endif (not consumed)"""
    out = []
    while i < len(lines):
        d, rest = directive(lines[i])
        if d in ("elif", "else", "endif"):
            return out, i
        if d in ("if", "ifdef", "ifndef"):
            branches = []          # (directive text or None for else, value, body)
            cond_line = lines[i]
            if d == "if":
                val = evaluate(rest, fixed, undef)
            else:
                name = rest.split()[0]
                known = True if name in fixed else (False if name in undef else None)
                val = None if known is None else (known if d == "ifdef" else not known)
            body, i = parse(lines, i + 1, fixed, undef)
            branches.append((cond_line, val, body, d))
            while True:
                d2, rest2 = directive(lines[i])
                if d2 == "elif":
                    cl = lines[i]
                    body, i = parse(lines, i + 1, fixed, undef)
                    branches.append((cl, evaluate(rest2, fixed, undef), body, "elif"))
                elif d2 == "else":
                    cl = lines[i]
                    body, i = parse(lines, i + 1, fixed, undef)
                    branches.append((cl, True, body, "else"))
                else:
                    assert d2 == "endif", (i, lines[i])
                    endif_line = lines[i]
                    i += 1
                    break
            # a fixed macro's default block: #ifndef X / #define X v / #endif -> the define alone
            kept = []
            for cl, val, body, kind in branches:
                if val is False:
                    continue
                kept.append((cl, val, body, kind))
                if val is True:
                    break
            if not kept:
                continue
            if kept[0][1] is True:
                out.extend(kept[0][2])
                continue
            first = True
            for cl, val, body, kind in kept:
                if val is True:
                    out.append(re.sub(r"#\s*(elif|else).*", "#else", cl.rstrip("\n")) + "\n" if kind != "else" else cl)
                elif first:
                    out.append(cl if kind != "elif" else re.sub(r"#(\s*)elif", r"#\1if", cl))
                else:
                    out.append(cl)
                first = False
                out.extend(body)
            out.append(endif_line)
            continue
        out.append(lines[i])
        i += 1
    return out, i


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE")
    ap.add_argument("--undef", action="append", default=[])
    ap.add_argument("--in-place", action="store_true")
    a = ap.parse_args()
    fixed = {}
    for s in a.set:
        k, v = s.split("=", 1)
        fixed[k] = int(v, 0)
    undef = set(a.undef)
    lines = open(a.file).read().splitlines(keepends=True)
    # pass 1: `#ifndef X` default blocks of fixed macros become bare defines (the parse below would otherwise drop them: X counts
    # as defined).  Trailing comment lines that continue on the #endif line are kept.
    res, i = [], 0
    while i < len(lines):
        d, rest = directive(lines[i])
        if d == "ifndef" and rest.split()[0] in fixed and i + 2 < len(lines) and re.match(r"\s*#\s*define\s+%s\b" % re.escape(rest.split()[0]), lines[i + 1]) \
                and directive(lines[i + 2])[0] == "endif":
            name = rest.split()[0]
            tail = re.sub(r"^\s*#\s*ifndef\s+\w+", "", lines[i]).rstrip("\n")
            define = re.sub(r"(#\s*define\s+%s\s+)(\S+)" % re.escape(name), lambda m: m.group(1) + str(fixed[name]), lines[i + 1].rstrip("\n"), count=1)
            end_tail = re.sub(r"^\s*#\s*endif", "", lines[i + 2]).rstrip("\n")
            res.append(define + "\n")
            for t in (tail, end_tail):
                if t.strip():
                    res.append(" " * 30 + t.strip() + "\n")
            i += 3
            continue
        res.append(lines[i])
        i += 1
    out, j = parse(res, 0, fixed, undef)
    assert j == len(res), "unbalanced conditionals at line %d" % j
    text = "".join(out)
    if a.in_place:
        open(a.file, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
