"""Resolve compile-time A/B knobs to their shipped values in the kernel headers (a small unifdef): `#if KNOB ... #else ... #endif` keeps the
live arm, the `#ifndef KNOB / #define KNOB v / #endif` default block goes, remaining uses become the literal.  Used once in round 6 to
retire the measured losers (scripts/ab/round6_pruned_knobs.patch is the reverse diff: `git apply` it to get the variants back)."""
import re
import sys

KNOBS = dict(a.split("=") for a in sys.argv[1].split(","))
FILES = sys.argv[2:]


def evaluate(expr):
    """(value, rewritten) -- value None when the expression still depends on something that is not a knob"""
    e = expr
    for k, v in KNOBS.items():
        e = re.sub(rf"defined\({k}\)", "1", e)
        e = re.sub(rf"\b{k}\b", v, e)
    if e == expr:
        return None, expr
    if re.fullmatch(r"[\d\s()&|=!<>+\-*]+", e):
        py = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
        return bool(eval(py)), e
    m = re.fullmatch(r"(defined\(\w+\))\s*&&\s*(.+)", e)            # defined(__HIP_DEVICE_COMPILE__) && <knob expression>
    if m and re.fullmatch(r"[\d\s()&|=!<>]+", m.group(2)):
        v = bool(eval(m.group(2).replace("&&", " and ").replace("||", " or ")))
        return (None, m.group(1)) if v else (False, e)
    raise SystemExit(f"cannot resolve: {expr!r} -> {e!r}")


for path in FILES:
    out = []
    stack = []          # per open #if: [resolved (None = keep the directives), live_now, taken_already]
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        st = ln.strip()
        live = all(f[1] for f in stack if f[0] is not None)
        m = re.match(r"#\s*ifndef\s+(\w+)", st)
        if m and m.group(1) in KNOBS and live:        # the default block: #ifndef K / #define K v [comment lines] / #endif
            j = i + 1
            while not lines[j].strip().startswith("#endif"):
                j += 1
            i = j + 1
            continue
        if re.match(r"#\s*if(n?def)?\b", st):
            if st.startswith("#if ") and live:
                val, rew = evaluate(st[4:].split("//")[0].strip())
                if val is None and rew != st[4:].split("//")[0].strip():
                    out.append(ln.replace(st[4:].split("//")[0].strip(), rew))
                    stack.append([None, True, True])
                elif val is None:
                    out.append(ln); stack.append([None, True, True])
                else:
                    stack.append([True, val, val])
            else:
                if live: out.append(ln)
                stack.append([None, True, True])
        elif re.match(r"#\s*elif\b", st):
            f = stack[-1]
            if f[0] is None:
                if live: out.append(ln)
            else:
                val, _ = evaluate(st.split(None, 1)[1].split("//")[0].strip())
                assert val is not None
                f[1] = (not f[2]) and val
                f[2] = f[2] or val
        elif re.match(r"#\s*else\b", st):
            f = stack[-1]
            if f[0] is None:
                if live: out.append(ln)
            else:
                f[1] = not f[2]; f[2] = True
        elif re.match(r"#\s*endif\b", st):
            f = stack.pop()
            if f[0] is None and all(g[1] for g in stack if g[0] is not None): out.append(ln)
        else:
            if live:
                for k, v in KNOBS.items():
                    ln = re.sub(rf"\(\s*{k}\s*!=\s*0\s*\)", "true" if int(v) else "false", ln) if "//" not in ln.split(k)[0] else ln
                    if not ln.lstrip().startswith("//"):
                        code, sep, com = ln.partition("//")
                        code = re.sub(rf"\b{k}\b", v, code)
                        ln = code + sep + com
                out.append(ln)
        i += 1
    assert not stack, path
    open(path, "w").write("\n".join(out))
